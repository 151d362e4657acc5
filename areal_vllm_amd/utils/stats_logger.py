"""Rank-0 statistics sink: console table + JSONL file, and wandb / swanlab / tensorboard when installed.

Counterpart of ``areal/utils/stats_logger.py:18-102`` in the reference; log directory
``{fileroot}/logs/{user}/{exp}/{trial}``. Optional back ends are imported lazily and skipped (with a
warning) when the package is missing; a ``stats.jsonl`` file is always written so runs stay inspectable
offline.
"""
import getpass
import json
import os
import time
from typing import Dict, List, Optional

import torch.distributed as dist

from ..api.cli_args import StatsLoggerConfig
from ..api.io_struct import FinetuneSpec
from . import logging

logger = logging.getLogger("StatsLogger")


class StatsLogger:
    def __init__(self, config: StatsLoggerConfig, ft_spec: FinetuneSpec):
        self.config = config
        self.ft_spec = ft_spec
        self._wandb = None
        self._swanlab = None
        self._tb = None
        self._jsonl = None
        self._last_commit_step = -1
        self.start_time = time.perf_counter()
        self.init()

    def _is_sink_rank(self) -> bool:
        return not dist.is_initialized() or dist.get_rank() == 0

    def init(self):
        if not self._is_sink_rank():
            return
        log_dir = self.get_log_path(self.config)
        os.makedirs(log_dir, exist_ok=True)
        self._jsonl = open(os.path.join(log_dir, "stats.jsonl"), "a")
        c = self.config
        if c.wandb.mode != "disabled":
            try:
                import wandb

                wandb.init(mode=c.wandb.mode, entity=c.wandb.entity, project=c.wandb.project or c.experiment_name,
                           name=c.wandb.name or c.trial_name, job_type=c.wandb.job_type,
                           group=c.wandb.group or f"{c.experiment_name}_{c.trial_name}", notes=c.wandb.notes,
                           tags=c.wandb.tags, config=c.wandb.config, dir=log_dir, force=True,
                           id=f"{c.experiment_name}_{c.trial_name}_train", resume="allow")
                self._wandb = wandb
            except ImportError:
                logger.warning("wandb is not installed; skipping the wandb sink")
        if c.swanlab.mode not in (None, "disabled") and c.swanlab.project:
            try:
                import swanlab

                swanlab.init(project=c.swanlab.project, experiment_name=c.swanlab.name or c.trial_name,
                             config=c.swanlab.config, logdir=c.swanlab.logdir or log_dir, mode=c.swanlab.mode)
                self._swanlab = swanlab
            except ImportError:
                logger.warning("swanlab is not installed; skipping the swanlab sink")
        if c.tensorboard.path is not None:
            try:
                from torch.utils.tensorboard import SummaryWriter

                self._tb = SummaryWriter(log_dir=c.tensorboard.path)
            except Exception:
                logger.warning("tensorboard is not installed; skipping the tensorboard sink")

    def state_dict(self):
        return {"last_commit_step": self._last_commit_step}

    def load_state_dict(self, sd):
        self._last_commit_step = sd.get("last_commit_step", -1)

    def close(self):
        if not self._is_sink_rank():
            return
        logger.info(f"Training completes! Total time elapsed {time.perf_counter() - self.start_time:.2f}s.")
        if self._wandb is not None:
            self._wandb.finish()
        if self._swanlab is not None:
            self._swanlab.finish()
        if self._tb is not None:
            self._tb.close()
        if self._jsonl is not None:
            self._jsonl.close()
            self._jsonl = None

    def commit(self, epoch: int, step: int, global_step: int, data):
        if not self._is_sink_rank():
            return
        if isinstance(data, dict):
            data = [data]
        logger.info(f"Epoch {epoch + 1}/{self.ft_spec.total_train_epochs} "
                    f"Step {step + 1}/{self.ft_spec.steps_per_epoch} "
                    f"Train step {global_step + 1}/{self.ft_spec.total_train_steps} done.")
        log_step = max(global_step, self._last_commit_step + 1)
        for i, item in enumerate(data):
            logger.info(f"Stats ({i + 1}/{len(data)}):\n" + self.get_stats_table(item))
            if self._jsonl is not None:
                self._jsonl.write(json.dumps(dict(global_step=global_step, minibatch=i, **item)) + "\n")
                self._jsonl.flush()
            if self._wandb is not None:
                self._wandb.log(item, step=log_step + i)
            if self._swanlab is not None:
                self._swanlab.log(item, step=log_step + i)
            if self._tb is not None:
                for k, v in item.items():
                    self._tb.add_scalar(k, v, log_step + i)
        self._last_commit_step = log_step + len(data) - 1

    @staticmethod
    def get_stats_table(stats: Dict[str, float]) -> str:
        try:
            from tabulate import tabulate

            return tabulate([(k, f"{v:.4e}" if isinstance(v, float) else v) for k, v in sorted(stats.items())],
                            headers=["key", "value"], tablefmt="github")
        except ImportError:
            return "\n".join(f"{k}: {v}" for k, v in sorted(stats.items()))

    @staticmethod
    def get_log_path(config: StatsLoggerConfig) -> str:
        path = f"{config.fileroot}/logs/{getpass.getuser()}/{config.experiment_name}/{config.trial_name}"
        os.makedirs(path, exist_ok=True)
        return path
