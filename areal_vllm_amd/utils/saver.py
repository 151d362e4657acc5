"""Frequency-controlled HF checkpoints with the reference's directory layout
(``areal/utils/saver.py:24-120``): ``{fileroot}/checkpoints/{user}/{exp}/{trial}/{name}/
epoch{e}epochstep{s}globalstep{g}`` and ``.../recover_checkpoint``."""
import getpass
import os
from typing import Optional

from ..api.cli_args import SaverConfig
from ..api.io_struct import FinetuneSpec, SaveLoadMeta
from . import timeutil


class Saver:
    def __init__(self, config: SaverConfig, ft_spec: FinetuneSpec):
        self.config = config
        self.ft_spec = ft_spec
        self.freq_ctl = timeutil.EpochStepTimeFreqCtl(freq_epoch=config.freq_epochs, freq_step=config.freq_steps,
                                                      freq_sec=config.freq_secs)

    @staticmethod
    def get_save_root(experiment_name: str, trial_name: str, fileroot: str) -> str:
        path = os.path.join(f"{fileroot}/checkpoints/{getpass.getuser()}/{experiment_name}/{trial_name}")
        os.makedirs(path, exist_ok=True)
        return path

    @staticmethod
    def get_model_save_root(experiment_name: str, trial_name: str, fileroot: str, name: str = "default") -> str:
        path = os.path.join(Saver.get_save_root(experiment_name, trial_name, fileroot), name)
        os.makedirs(path, exist_ok=True)
        return path

    @staticmethod
    def get_model_save_path(experiment_name: str, trial_name: str, fileroot: str, epoch: int, step: int,
                            globalstep: int, name: str = "default") -> str:
        path = os.path.join(Saver.get_model_save_root(experiment_name, trial_name, fileroot, name),
                            f"epoch{epoch}epochstep{step}globalstep{globalstep}")
        os.makedirs(path, exist_ok=True)
        return path

    @staticmethod
    def get_recover_checkpoint_path(experiment_name: str, trial_name: str, fileroot: str, name: str = "default") -> str:
        path = os.path.join(Saver.get_model_save_root(experiment_name, trial_name, fileroot, name),
                            "recover_checkpoint")
        os.makedirs(path, exist_ok=True)
        return path

    def state_dict(self):
        return self.freq_ctl.state_dict()

    def load_state_dict(self, state_dict):
        self.freq_ctl.load_state_dict(state_dict)

    def save(self, engine, epoch: int, step: int, global_step: int, name: str = "default", tokenizer=None,
             processor=None, base_model_path: Optional[str] = None):
        if not self.freq_ctl.check(epochs=int(step == self.ft_spec.steps_per_epoch - 1), steps=1):
            return None
        path = Saver.get_model_save_path(self.config.experiment_name, self.config.trial_name, self.config.fileroot,
                                         epoch, step, global_step, name)
        engine.save(SaveLoadMeta(path=path, weight_format="hf", with_optim=False, tokenizer=tokenizer,
                                 processor=processor, base_model_path=base_model_path))
        return path
