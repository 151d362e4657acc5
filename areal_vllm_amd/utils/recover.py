"""Checkpoint-and-recover: recover info files + recover checkpoint (weights and optimizer).

File set and semantics of ``areal/utils/recover.py:28-383`` in the reference: ``step_info.json``,
``saver_info.json``, ``evaluator_info.json``, ``stats_logger_info.json``, ``checkpoint_info.json``,
``dataloader_info.pkl`` under ``{save_root}/recover_info``; ``RecoverHandler.load`` restores the trainer,
data position and frequency controllers, re-uploads the weights to the rollout engine and restores the
weight versions; ``check_if_recover`` implements the launcher-side modes disabled/auto/fault/resume.
"""
import dataclasses
import json
import os
import pickle
from typing import Dict, List, Optional, Union

import torch.distributed as dist

from ..api.cli_args import RecoverConfig
from ..api.engine_api import InferenceEngine, TrainEngine
from ..api.io_struct import FinetuneSpec, SaveLoadMeta, StepInfo, WeightUpdateMeta
from . import logging, timeutil
from .evaluator import Evaluator
from .saver import Saver
from .stats_logger import StatsLogger

logger = logging.getLogger("recover")


class InValidRecoverInfo(Exception):
    pass


@dataclasses.dataclass
class RecoverInfo:
    last_step_info: StepInfo
    saver_info: Dict
    evaluator_info: Dict
    stats_logger_info: Dict
    dataloader_info: Union[Dict, List[Dict]]
    checkpoint_info: Dict

    def dump(self, dump_dir: str):
        if dist.is_initialized():
            infos = [None] * dist.get_world_size()
            dist.all_gather_object(infos, self.dataloader_info)
            if dist.get_rank() != 0:
                return
        else:
            infos = self.dataloader_info
        os.makedirs(dump_dir, exist_ok=True)

        def wj(name, obj):
            tmp = os.path.join(dump_dir, name + ".tmp")
            with open(tmp, "w") as f:
                json.dump(obj, f, indent=4)
            os.replace(tmp, os.path.join(dump_dir, name))

        wj("saver_info.json", self.saver_info)
        wj("evaluator_info.json", self.evaluator_info)
        wj("stats_logger_info.json", self.stats_logger_info)
        wj("checkpoint_info.json", self.checkpoint_info)
        tmp = os.path.join(dump_dir, "dataloader_info.pkl.tmp")
        with open(tmp, "wb") as f:
            pickle.dump(infos, f)
        os.replace(tmp, os.path.join(dump_dir, "dataloader_info.pkl"))
        # step_info.json is written last: its presence marks a complete dump
        wj("step_info.json", dataclasses.asdict(self.last_step_info))

    @classmethod
    def load(cls, load_dir: str) -> "RecoverInfo":
        if not os.path.exists(load_dir):
            raise FileNotFoundError(f"Recover info directory {load_dir} does not exist.")
        try:
            def rj(name):
                with open(os.path.join(load_dir, name)) as f:
                    return json.load(f)

            step = StepInfo(**rj("step_info.json"))
            with open(os.path.join(load_dir, "dataloader_info.pkl"), "rb") as f:
                dl = pickle.load(f)  # written by this framework's own dump()
            if isinstance(dl, list) and dist.is_initialized():
                assert dist.get_world_size() == len(dl), (
                    f"Dataloader info list length {len(dl)} does not match the world size {dist.get_world_size()}.")
                dl = dl[dist.get_rank()]
            elif isinstance(dl, list) and len(dl) == 1:
                dl = dl[0]
            return cls(last_step_info=step, saver_info=rj("saver_info.json"), evaluator_info=rj("evaluator_info.json"),
                       stats_logger_info=rj("stats_logger_info.json"), dataloader_info=dl,
                       checkpoint_info=rj("checkpoint_info.json"))
        except Exception as e:
            logger.error(f"Failed to load recover info from {load_dir}: {e}")
            raise InValidRecoverInfo(f"Invalid recover info in {load_dir}") from e


class RecoverHandler:
    def __init__(self, config: RecoverConfig, ft_spec: FinetuneSpec):
        self.config = config
        self.ft_spec = ft_spec
        self.last_step_info = StepInfo(epoch=-1, epoch_step=-1, global_step=-1, steps_per_epoch=ft_spec.steps_per_epoch)
        self.freq_ctl = timeutil.EpochStepTimeFreqCtl(freq_epoch=config.freq_epochs, freq_step=config.freq_steps,
                                                      freq_sec=config.freq_secs)

    @staticmethod
    def recover_info_path(experiment_name: str, trial_name: str, fileroot: str) -> str:
        return os.path.join(Saver.get_save_root(experiment_name, trial_name, fileroot), "recover_info")

    def dump(self, engine, step_info: StepInfo, saver: Saver, evaluator: Evaluator, stats_logger: StatsLogger,
             dataloader, tokenizer=None, processor=None, base_model_path: Optional[str] = None, force: bool = False):
        if self.config.mode == "disabled":
            return
        if not force and not self.freq_ctl.check(
                epochs=int(step_info.epoch_step == self.ft_spec.steps_per_epoch - 1), steps=1):
            return
        if isinstance(engine, TrainEngine):
            engine = {"default": engine}
        for name, e in engine.items():
            path = Saver.get_recover_checkpoint_path(self.config.experiment_name, self.config.trial_name,
                                                     self.config.fileroot, name=name)
            e.save(SaveLoadMeta(path=path, weight_format="hf", with_optim=True, tokenizer=tokenizer,
                                processor=processor, base_model_path=base_model_path))
            logger.info(f"Saved recover checkpoint to {path}")
        self.last_step_info = step_info
        RecoverInfo(last_step_info=step_info, saver_info=saver.state_dict(), evaluator_info=evaluator.state_dict(),
                    stats_logger_info=stats_logger.state_dict(), dataloader_info=dataloader.state_dict(),
                    checkpoint_info=self.freq_ctl.state_dict()).dump(
            self.recover_info_path(self.config.experiment_name, self.config.trial_name, self.config.fileroot))

    def load(self, engine, saver: Saver, evaluator: Evaluator, stats_logger: StatsLogger, dataloader,
             inference_engine: Optional[InferenceEngine] = None, weight_update_meta: Optional[WeightUpdateMeta] = None,
             inference_engine_update_from: str = "default") -> Optional[RecoverInfo]:
        if self.config.mode == "disabled":
            return None
        if os.environ.get("AREAL_RECOVER_RUN", "0") != "1":
            return None
        if inference_engine is not None:
            assert weight_update_meta is not None, "Inference engine requires weight update meta for recovery."
        if isinstance(engine, TrainEngine):
            engine = {"default": engine}
        path = self.recover_info_path(self.config.experiment_name, self.config.trial_name, self.config.fileroot)
        logger.info(f"Loading recover info from {path}")
        try:
            info = RecoverInfo.load(path)
        except (FileNotFoundError, InValidRecoverInfo):
            logger.warning(f"Resume info not found at {path}. This should not be a resumed experiment!")
            return None
        logger.info(f"Recovering from {info.last_step_info.next()}.")
        saver.load_state_dict(info.saver_info)
        self.freq_ctl.load_state_dict(info.checkpoint_info)
        evaluator.load_state_dict(info.evaluator_info)
        stats_logger.load_state_dict(info.stats_logger_info)
        dataloader.load_state_dict(info.dataloader_info)
        for name, e in engine.items():
            ckpt = Saver.get_recover_checkpoint_path(self.config.experiment_name, self.config.trial_name,
                                                     self.config.fileroot, name=name)
            if not os.path.exists(ckpt):
                raise FileNotFoundError(f"Checkpoint path {ckpt} does not exist.")
            e.load(SaveLoadMeta(path=ckpt, weight_format="hf", with_optim=True))
        global_step = info.last_step_info.global_step
        self.last_step_info = info.last_step_info
        if weight_update_meta is not None:
            # Every trainer rank takes part (a tensor-parallel trainer gathers its shards inside
            # `upload_weights`, and the barrier is a world collective); only the ranks that drive a rollout
            # engine - the data-parallel heads - talk to it.
            upd = engine[inference_engine_update_from]
            is0 = (not dist.is_initialized()) or dist.get_rank() == 0
            fut = None
            if inference_engine is not None:
                inference_engine.pause()
                if is0 or weight_update_meta.type == "inproc":  # every colocated head drives its own engine
                    fut = inference_engine.update_weights(weight_update_meta)
            upd.upload_weights(weight_update_meta)
            if fut is not None:
                fut.result()
            if dist.is_initialized():
                dist.barrier()
            upd.set_version(global_step + 1)
            if inference_engine is not None:
                inference_engine.resume()
                inference_engine.set_version(global_step + 1)
        return info


def check_if_auto_recover(config: RecoverConfig) -> bool:
    path = RecoverHandler.recover_info_path(config.experiment_name, config.trial_name, config.fileroot)
    logger.info(f"Searching for recover info file in {path}.")
    if not os.path.exists(str(path)):
        logger.warning(f"Recover info not found at: {path}")
        return False
    try:
        info = RecoverInfo.load(path)
    except Exception as e:
        logger.warning(f"Failed to load recover info from {path}: {e}")
        return False
    if info.last_step_info.epoch < 0:
        logger.warning(f"Recover checkpoint is not valid (epoch {info.last_step_info.epoch} < 0)")
        return False
    save_root = Saver.get_save_root(config.experiment_name, config.trial_name, config.fileroot)
    for name in os.listdir(save_root):
        if name == "recover_info" or not os.path.isdir(os.path.join(save_root, name)):
            continue
        ckpt = os.path.join(save_root, name, "recover_checkpoint")
        if not os.path.exists(ckpt):
            logger.warning(f"Recover checkpoint for model {name} does not exist.")
            return False
    return True


def check_if_recover(config: RecoverConfig, run_id: int) -> bool:
    if config.mode == "disabled":
        return False
    if config.mode == "auto":
        return check_if_auto_recover(config)
    if config.mode == "fault":
        return run_id > 0
    if config.mode == "resume":
        return True
    raise ValueError(f"Unknown recover mode: {config.mode}")
