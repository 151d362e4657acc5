"""Supervised fine-tuning loop shared by the SFT example scripts (reference control flow:
``examples/math/gsm8k_sft.py:21-171``): per step one `train_lm` over a padded batch of this rank's rows,
periodic save / evaluation / recover checkpoints, statistics reduced over data-parallel ranks."""
import os
from typing import Optional

import torch.distributed as dist

from ..api.alloc_mode import AllocationMode
from ..api.cli_args import SFTConfig, load_expr_config
from ..api.io_struct import FinetuneSpec, StepInfo
from ..dataset import get_custom_dataset
from ..engine.sft.lm_engine import NativeLMEngine
from ..platforms import current_platform
from ..utils import logging, seeding, stats_tracker
from ..utils.data import broadcast_tensor_container, pad_sequences_to_tensors, to_device
from ..utils.dataloader import StatefulDataLoader
from ..utils.evaluator import Evaluator
from ..utils.recover import RecoverHandler
from ..utils.saver import Saver
from ..utils.stats_logger import StatsLogger

logger = logging.getLogger("sft-trainer")


def run_sft(argv, config_cls=SFTConfig, processor_needed: bool = False):
    config, _ = load_expr_config(argv, config_cls)
    rank = int(os.getenv("RANK", "0"))
    tokenizer = processor = None
    if config.tokenizer_path and os.path.isdir(config.tokenizer_path):
        from ..utils.hf_utils import load_hf_processor_and_tokenizer, load_hf_tokenizer

        if processor_needed:
            processor, tokenizer = load_hf_processor_and_tokenizer(config.tokenizer_path)
        else:
            tokenizer = load_hf_tokenizer(config.tokenizer_path)
    seeding.set_random_seed(config.seed, f"trainer{rank}")
    am = AllocationMode.from_str(config.allocation_mode)
    engine = NativeLMEngine(config=config.model)
    engine.create_process_group(parallel_strategy=am.train)
    kw = dict(rank=engine.data_parallel_rank, world_size=engine.data_parallel_world_size, tokenizer=tokenizer,
              processor=processor)
    train_ds = get_custom_dataset(path=config.train_dataset.path, split="train", type=config.train_dataset.type or "sft",
                                  max_length=config.train_dataset.max_length, **kw)
    valid_ds = None
    if config.valid_dataset is not None and config.valid_dataset.path:
        valid_ds = get_custom_dataset(path=config.valid_dataset.path, split="test",
                                      type=config.valid_dataset.type or "sft",
                                      max_length=config.valid_dataset.max_length, **kw)
    dp = engine.data_parallel_world_size
    train_dl = StatefulDataLoader(train_ds, batch_size=max(1, config.train_dataset.batch_size // dp),
                                  shuffle=config.train_dataset.shuffle, drop_last=config.train_dataset.drop_last,
                                  collate_fn=pad_sequences_to_tensors, seed=config.seed)
    valid_dl = None
    if valid_ds is not None:
        valid_dl = StatefulDataLoader(valid_ds, batch_size=max(1, config.valid_dataset.batch_size // dp), shuffle=False,
                                      drop_last=config.valid_dataset.drop_last, collate_fn=pad_sequences_to_tensors)
    ft_spec = FinetuneSpec(total_train_epochs=config.total_train_epochs,
                           dataset_size=len(train_dl) * config.train_dataset.batch_size,
                           train_batch_size=config.train_dataset.batch_size)
    engine.initialize(None, ft_spec)
    saver, stats_logger = Saver(config.saver, ft_spec), StatsLogger(config.stats_logger, ft_spec)
    evaluator, recover_handler = Evaluator(config.evaluator, ft_spec), RecoverHandler(config.recover, ft_spec)
    recover_info = recover_handler.load(engine, saver, evaluator, stats_logger, train_dl)
    start_step = recover_info.last_step_info.next().global_step if recover_info is not None else 0
    steps_per_epoch = len(train_dl)
    max_steps = config.total_train_epochs * steps_per_epoch
    if config.total_train_steps is not None:
        max_steps = min(max_steps, config.total_train_steps)

    def batches():
        while True:
            yield from train_dl

    gen = batches()
    losses = []
    for global_step in range(start_step, max_steps):
        epoch, step = divmod(global_step, steps_per_epoch)
        data = to_device(next(gen), engine.device)
        data = broadcast_tensor_container(data, src_rank=engine.current_data_parallel_head(),
                                          group=engine.context_and_model_parallel_group)
        with stats_tracker.record_timing("train_step"), stats_tracker.scope("sft"):
            stats = engine.train_lm(data)
            engine.step_lr_scheduler()
        with stats_tracker.record_timing("save"):
            saver.save(engine, epoch, step, global_step, tokenizer=tokenizer, processor=processor)
        with stats_tracker.record_timing("eval"), stats_tracker.scope("sft-eval"):
            def evaluate_fn():
                if valid_dl is None:
                    return
                for vd in valid_dl:
                    vd = to_device(vd, engine.device)
                    vd = broadcast_tensor_container(vd, src_rank=engine.current_data_parallel_head(),
                                                    group=engine.context_and_model_parallel_group)
                    engine.evaluate_lm(vd)

            evaluator.evaluate(evaluate_fn, epoch, step, global_step)
        with stats_tracker.record_timing("checkpoint_for_recover"):
            recover_handler.dump(engine, StepInfo(global_step=global_step, epoch=epoch, epoch_step=step,
                                                  steps_per_epoch=steps_per_epoch), saver, evaluator, stats_logger,
                                 train_dl, tokenizer=tokenizer)
        if dist.is_initialized():
            dist.barrier()
        current_platform.synchronize()
        out = dict(stats) if isinstance(stats, dict) else {}
        out.update(engine.share_pipeline_stats(stats_tracker.export_all(reduce_group=engine.data_parallel_group)))
        stats_logger.commit(epoch, step, global_step, out)
        losses.append(out.get("sft/loss/avg", out.get("sft/loss", float("nan"))))
    stats_logger.close()
    engine.destroy()
    return losses
