"""Pieces shared by the RL and SFT trainers: the per-rank data loaders and the bundle of periodic actions
(checkpoint, evaluation, recover checkpoint, statistics) that closes every training step."""
from typing import Any, Callable, Optional

import torch.distributed as dist

from ..api.io_struct import FinetuneSpec, StepInfo
from ..dataset import get_custom_dataset
from ..platforms import current_platform
from ..utils import stats_tracker
from ..utils.dataloader import StatefulDataLoader
from ..utils.evaluator import Evaluator
from ..utils.recover import RecoverHandler
from ..utils.saver import Saver
from ..utils.stats_logger import StatsLogger


def sync_ranks():
    """Every rank has issued, and its device has finished, everything before this point."""
    if dist.is_initialized():
        dist.barrier()
    current_platform.synchronize()


def build_dataloaders(config, engine, default_type: str, tokenizer=None, processor=None, collate_fn=None):
    """-> (train loader, validation loader | None, FinetuneSpec). Every data-parallel rank reads its own slice
    of the dataset; the configured batch size is the global one."""
    kw = dict(rank=engine.data_parallel_rank, world_size=engine.data_parallel_world_size, tokenizer=tokenizer)
    if processor is not None:
        kw["processor"] = processor
    dp = engine.data_parallel_world_size
    extra = dict(collate_fn=collate_fn) if collate_fn is not None else {}
    tc, vc = config.train_dataset, config.valid_dataset
    train = get_custom_dataset(path=tc.path, split="train", max_length=tc.max_length, type=tc.type or default_type, **kw)
    train_dl = StatefulDataLoader(train, batch_size=max(1, tc.batch_size // dp), shuffle=tc.shuffle,
                                  drop_last=tc.drop_last, seed=config.seed, **extra)
    valid_dl = None
    if vc is not None and vc.path:
        valid = get_custom_dataset(path=vc.path, split="test", max_length=vc.max_length, type=vc.type or default_type,
                                   **kw)
        valid_dl = StatefulDataLoader(valid, batch_size=max(1, vc.batch_size // dp), shuffle=False,
                                      drop_last=vc.drop_last, **extra)
    spec = FinetuneSpec(total_train_epochs=config.total_train_epochs, dataset_size=len(train_dl) * tc.batch_size,
                        train_batch_size=tc.batch_size)
    return train_dl, valid_dl, spec


class StepBudget:
    """How many steps a run takes and where a recovered run picks up."""

    def __init__(self, config, steps_per_epoch: int, recover_info):
        self.steps_per_epoch = steps_per_epoch
        self.start = recover_info.last_step_info.next().global_step if recover_info is not None else 0
        self.stop = config.total_train_epochs * steps_per_epoch
        if config.total_train_steps is not None:
            self.stop = min(self.stop, config.total_train_steps)

    def __iter__(self):
        for g in range(self.start, self.stop):
            epoch, step = divmod(g, self.steps_per_epoch)
            yield StepInfo(global_step=g, epoch=epoch, epoch_step=step, steps_per_epoch=self.steps_per_epoch)


class PeriodicActions:
    """Checkpoint writer, evaluator, recover handler and statistics logger of one run. `after_step` is the
    tail of a training step: save if due -> evaluate if due -> recover checkpoint if due -> reduce and log the
    step's statistics. `on_no_eval` runs first when no evaluation is due at this step (the RL trainer resumes
    generation there: nothing else in the tail needs the rollout engines quiet)."""

    def __init__(self, config, ft_spec: FinetuneSpec):
        self.saver = Saver(config.saver, ft_spec)
        self.stats_logger = StatsLogger(config.stats_logger, ft_spec)
        self.evaluator = Evaluator(config.evaluator, ft_spec)
        self.recover = RecoverHandler(config.recover, ft_spec)

    def load(self, engine, dataloader, **kw):
        return self.recover.load(engine, self.saver, self.evaluator, self.stats_logger, dataloader, **kw)

    def after_step(self, engine, info: StepInfo, stats: Any, dataloader, evaluate_fn: Optional[Callable[[], Any]] = None,
                   on_no_eval: Optional[Callable[[], Any]] = None, eval_scope: Optional[str] = None, tokenizer=None,
                   processor=None):
        epoch, step, g = info.epoch, info.epoch_step, info.global_step
        eval_due = self.evaluator.due(step)
        if not eval_due and on_no_eval is not None:
            on_no_eval()
        with stats_tracker.record_timing("save"):
            self.saver.save(engine, epoch, step, g, tokenizer=tokenizer, processor=processor)
        with stats_tracker.record_timing("eval"):
            if eval_due and evaluate_fn is not None:
                if eval_scope:
                    with stats_tracker.scope(eval_scope):
                        evaluate_fn()
                else:
                    evaluate_fn()
        with stats_tracker.record_timing("checkpoint_for_recover"):
            self.recover.dump(engine, info, self.saver, self.evaluator, self.stats_logger, dataloader,
                              tokenizer=tokenizer)
        sync_ranks()
        reduced = engine.share_pipeline_stats(stats_tracker.export_all(reduce_group=engine.data_parallel_group))
        if isinstance(stats, list):
            stats[0].update(reduced)
            out = stats
        else:
            out = dict(stats) if isinstance(stats, dict) else {}
            out.update(reduced)
        self.stats_logger.commit(epoch, step, g, out)
        sync_ranks()
        return out

    def close(self):
        self.stats_logger.close()
