"""Periodic HF checkpoints and where they live.

Directory scheme (part of the user-facing contract, same as ``areal/utils/saver.py:24-78`` of the reference):

    {fileroot}/checkpoints/{user}/{experiment}/{trial}/{model name}/epoch{e}epochstep{s}globalstep{g}
    {fileroot}/checkpoints/{user}/{experiment}/{trial}/{model name}/recover_checkpoint
"""
import getpass
from pathlib import Path
from typing import Optional

from ..api.io_struct import SaveLoadMeta
from .periodic import PeriodicTrigger


class CheckpointLayout:
    """Directories of one trial's checkpoints; every accessor creates the directory it names (callers write
    into it right away)."""

    def __init__(self, fileroot: str, experiment_name: str, trial_name: str):
        self.trial = Path(fileroot) / "checkpoints" / getpass.getuser() / experiment_name / trial_name

    @staticmethod
    def _made(p: Path) -> str:
        p.mkdir(parents=True, exist_ok=True)
        return str(p)

    def root(self) -> str:
        return self._made(self.trial)

    def model(self, name: str) -> str:
        return self._made(self.trial / name)

    def step(self, name: str, epoch: int, step: int, global_step: int) -> str:
        return self._made(self.trial / name / f"epoch{epoch}epochstep{step}globalstep{global_step}")

    def recover(self, name: str) -> str:
        return self._made(self.trial / name / "recover_checkpoint")


class Saver(PeriodicTrigger):
    # path getters kept as static methods: launchers and recover code call them without a Saver instance
    @staticmethod
    def get_save_root(experiment_name: str, trial_name: str, fileroot: str) -> str:
        return CheckpointLayout(fileroot, experiment_name, trial_name).root()

    @staticmethod
    def get_model_save_root(experiment_name: str, trial_name: str, fileroot: str, name: str = "default") -> str:
        return CheckpointLayout(fileroot, experiment_name, trial_name).model(name)

    @staticmethod
    def get_model_save_path(experiment_name: str, trial_name: str, fileroot: str, epoch: int, step: int,
                            globalstep: int, name: str = "default") -> str:
        return CheckpointLayout(fileroot, experiment_name, trial_name).step(name, epoch, step, globalstep)

    @staticmethod
    def get_recover_checkpoint_path(experiment_name: str, trial_name: str, fileroot: str, name: str = "default") -> str:
        return CheckpointLayout(fileroot, experiment_name, trial_name).recover(name)

    def save(self, engine, epoch: int, step: int, global_step: int, name: str = "default", tokenizer=None,
             processor=None, base_model_path: Optional[str] = None) -> Optional[str]:
        """Write a HF checkpoint of `engine` if one is due at this step; -> its directory or None."""
        if not self.due(step):
            return None
        c = self.config
        target = CheckpointLayout(c.fileroot, c.experiment_name, c.trial_name).step(name, epoch, step, global_step)
        engine.save(SaveLoadMeta(path=target, weight_format="hf", with_optim=False, tokenizer=tokenizer,
                                 processor=processor, base_model_path=base_model_path))
        return target
