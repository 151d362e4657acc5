"""Supervised fine-tuning trainer shared by the SFT example scripts (what a step does follows the reference's
``examples/math/gsm8k_sft.py:21-171``): a step is `fetch` (this rank's padded rows, handed from the data-parallel
head to the ranks of its replica) -> `update` (one `train_lm`) -> the periodic tail shared with the RL trainer
(checkpoint / evaluation / recover checkpoint / statistics reduced over data-parallel ranks)."""
import os
from typing import List

from ..api.alloc_mode import AllocationMode
from ..api.cli_args import SFTConfig, load_expr_config
from ..engine.sft.lm_engine import NativeLMEngine
from ..utils import logging, seeding, stats_tracker
from ..utils.data import broadcast_tensor_container, pad_sequences_to_tensors, to_device
from .common import PeriodicActions, StepBudget, build_dataloaders

logger = logging.getLogger("sft-trainer")


class SFTTrainer:
    def __init__(self, argv, config_cls=SFTConfig, processor_needed: bool = False):
        self.config, _ = load_expr_config(argv, config_cls)
        config = self.config
        rank = int(os.getenv("RANK", "0"))
        self.tokenizer = self.processor = None
        if config.tokenizer_path and os.path.isdir(config.tokenizer_path):
            from ..utils.hf_utils import load_hf_processor_and_tokenizer, load_hf_tokenizer

            if processor_needed:
                self.processor, self.tokenizer = load_hf_processor_and_tokenizer(config.tokenizer_path)
            else:
                self.tokenizer = load_hf_tokenizer(config.tokenizer_path)
        seeding.set_random_seed(config.seed, f"trainer{rank}")
        self.engine = NativeLMEngine(config=config.model)
        self.engine.create_process_group(parallel_strategy=AllocationMode.from_str(config.allocation_mode).train)
        self.train_dl, self.valid_dl, ft_spec = build_dataloaders(
            config, self.engine, "sft", self.tokenizer, self.processor, collate_fn=pad_sequences_to_tensors)
        self.engine.initialize(None, ft_spec)
        self.periodic = PeriodicActions(config, ft_spec)
        self.budget = StepBudget(config, len(self.train_dl), self.periodic.load(self.engine, self.train_dl))
        self._rows = self._endless(self.train_dl)

    @staticmethod
    def _endless(loader):
        while True:
            yield from loader

    def _to_replica(self, data):
        e = self.engine
        return broadcast_tensor_container(to_device(data, e.device), src_rank=e.current_data_parallel_head(),
                                          group=e.context_and_model_parallel_group)

    def fetch(self):
        return self._to_replica(next(self._rows))

    def update(self, data):
        with stats_tracker.record_timing("train_step"), stats_tracker.scope("sft"):
            stats = self.engine.train_lm(data)
            self.engine.step_lr_scheduler()
        return stats

    def evaluate(self):
        if self.valid_dl is not None:
            for rows in self.valid_dl:
                self.engine.evaluate_lm(self._to_replica(rows))

    def run(self) -> List[float]:
        losses = []
        for info in self.budget:
            stats = self.update(self.fetch())
            out = self.periodic.after_step(self.engine, info, stats, self.train_dl, evaluate_fn=self.evaluate,
                                           eval_scope="sft-eval", tokenizer=self.tokenizer, processor=self.processor)
            losses.append(out.get("sft/loss/avg", out.get("sft/loss", float("nan"))))
        return losses

    def close(self):
        self.periodic.close()
        self.engine.destroy()


def run_sft(argv, config_cls=SFTConfig, processor_needed: bool = False):
    trainer = SFTTrainer(argv, config_cls, processor_needed)
    losses = trainer.run()
    trainer.close()
    return losses
