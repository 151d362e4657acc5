"""ZeRO-style sharded AdamW over flat parameter / gradient buffers.

MI355X-first choice: with 288 GB of HBM per GPU the bf16 parameters and fp32 gradient accumulators are
kept replicated (ZeRO-1) and only the fp32 master weights and Adam moments are sharded over the
data-parallel group. A step is then ONE reduce-scatter of the flat fp32 gradient buffer and ONE
all-gather of the flat bf16 parameter buffer (large, bandwidth-efficient xGMI transfers) instead of the
per-layer all-gather / reduce-scatter traffic of FSDP2 that the reference uses
(``areal/utils/fsdp.py:253-285``). The reduce-scatter can hide behind the last backward of the step
(`arm_overlap`): every rank's shard range is cut into a few buckets, and as soon as the gradients of all
parameters that touch a bucket are final, the bucket is reduced to its owner asynchronously - the same bytes over
the same links, issued while the remaining layers still compute. A 7B model costs 14 GB (bf16 params) + 28 GB (fp32 grads) replicated
plus 84/dp GB of sharded optimizer state per GPU; larger models add tensor / pipeline parallelism.

The whole step (norm -> clip coefficient -> non-finite check -> AdamW -> bf16 shadow) runs on device
with no host synchronisation; the gradient norm is read back lazily for logging.
"""
import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from .. import ops
from ..api.cli_args import OptimizerConfig
from ..models.transformer import FlatParams
from . import comm


class ShardedAdamW:
    def __init__(self, flat: FlatParams, cfg: OptimizerConfig, dp_group=None, mp_groups: Tuple = (),
                 replicated_ranges: Optional[List[Tuple[int, int, int]]] = None):
        """`dp_group`: group over which parameters are replicated (gradients are summed over it).
        `mp_groups`: model-parallel groups (tp, pp) over which the squared gradient norm is summed.
        `replicated_ranges`: (lo, hi, rep) flat ranges of parameters held by `rep` ranks of the tp group
        (norm weights, embeddings, KV projections when tp exceeds the KV heads); their norm contribution
        is counted once."""
        self.flat = flat
        self.cfg = cfg
        self.group = dp_group
        self.mp_groups = [g for g in mp_groups if g is not None and comm.group_size(g) > 1]
        self.n = comm.group_size(dp_group)
        self.rank = comm.group_rank(dp_group)
        assert flat.numel % self.n == 0, "flat buffer must be padded to a multiple of the dp size"
        self.shard_numel = flat.numel // self.n
        self.lo = self.rank * self.shard_numel
        self.hi = self.lo + self.shard_numel
        dev = flat.flat.device
        self.master = flat.flat[self.lo:self.hi].float().clone()
        self.m = torch.zeros_like(self.master)
        self.v = torch.zeros_like(self.master)
        self.state = torch.zeros(ops.optim.STATE_SIZE, dtype=torch.float32, device=dev)
        self.grad_shard = (torch.empty(self.shard_numel, dtype=torch.float32, device=dev)
                           if self.n > 1 else None)
        self.replicated_ranges = sorted(replicated_ranges or [])
        self._norm_buf = torch.zeros(2, dtype=torch.float32, device=dev)
        self.lr = cfg.lr
        # ---- overlapped reduction: buckets = `buckets_per_shard` equal slices of every rank's shard range
        self.buckets: List[Tuple[int, int, int]] = []  # (lo, hi, owner rank in the group)
        self._bucket_of: Dict[int, List[int]] = {}  # id(parameter) -> buckets its flat range touches
        self._bucket_params: List[int] = []
        self._armed = False
        self._pending: List[int] = []
        self._works: List = []
        self._launched: List[bool] = []
        self.overlapped_buckets = 0  # buckets that left during a backward pass (all steps so far)
        if self.n > 1:
            per = max(1, 4)
            for r in range(self.n):
                step = (self.shard_numel + per - 1) // per
                for b in range(per):
                    lo = r * self.shard_numel + b * step
                    hi = min((r + 1) * self.shard_numel, lo + step)
                    if hi > lo:
                        self.buckets.append((lo, hi, r))
            self._bucket_params = [0] * len(self.buckets)
            starts = [b[0] for b in self.buckets]
            import bisect

            for name, (o, numel) in flat.offsets.items():
                p = flat._named[name]
                first = max(0, bisect.bisect_right(starts, o) - 1)
                ids = [i for i in range(first, len(self.buckets))
                       if self.buckets[i][0] < o + numel and self.buckets[i][1] > o]
                self._bucket_of[id(p)] = ids
                # An embedding table reports twice when it is also the (tied) LM head: once through the
                # accumulation hook (head GEMM) and once from the lookup's backward, which adds its rows to the
                # fp32 buffer directly (ops/linear.py). Its buckets wait for both; an untied table never fires
                # the hook, so its buckets leave with the flush after the backward.
                events = 2 if name.split(".")[-1] in ("embed", "pos_embed") else 1
                for i in ids:
                    self._bucket_params[i] += events
                p._after_fold = self._param_done

    # ------------------------------------------------------------- overlapped gradient reduction
    def arm_overlap(self):
        """Call right before the LAST backward pass of a step: from now on a bucket is reduced to the rank that
        owns it as soon as every parameter touching it has folded its gradient into the fp32 buffer."""
        if self.n == 1:
            return
        self._armed = True
        self._pending = list(self._bucket_params)
        self._launched = [False] * len(self.buckets)
        self._works = []

    def _launch(self, i: int):
        lo, hi, owner = self.buckets[i]
        dst = dist.get_global_rank(self.group, owner) if self.group is not None else owner
        self._works.append(dist.reduce(self.flat.main_grad[lo:hi], dst=dst, op=dist.ReduceOp.SUM, group=self.group,
                                       async_op=True))
        self._launched[i] = True

    def _param_done(self, p):
        if not self._armed:
            return
        for i in self._bucket_of.get(id(p), ()):
            self._pending[i] -= 1
            if self._pending[i] == 0:
                self._launch(i)
                self.overlapped_buckets += 1

    # ------------------------------------------------------------------ gradient reduction
    def reduce_gradients(self):
        g = self.flat.main_grad
        if self.n == 1:
            return g
        if self._armed:
            # buckets some of whose parameters got no gradient in the last backward (and the padding at the end
            # of the buffer, when it is not empty) were never launched: reduce them now, then wait for all
            self._armed = False
            for i, done in enumerate(self._launched):
                if not done and self._bucket_params[i] > 0:
                    self._launch(i)
            for w in self._works:
                w.wait()
            self._works = []
            return g[self.lo:self.hi]
        comm.reduce_scatter_sum(self.grad_shard, g, self.group)
        return self.grad_shard

    def _sumsq(self, gshard: torch.Tensor):
        """Squared norm of the full (all ranks, all model-parallel shards) gradient in state[0].
        A range replicated `rep` times over the tp group contributes 1/rep of its square sum on every
        replica, so the plain sum over tp counts it exactly once."""
        buf = self._norm_buf
        buf.zero_()
        if not self.mp_groups or not self.replicated_ranges:
            ops.sumsq_(gshard, buf[0:1])
        else:
            cur = self.lo
            for (a, b, rep) in self.replicated_ranges:
                a2, b2 = max(a, self.lo), min(b, self.hi)
                if a2 >= b2:
                    continue
                if cur < a2:
                    ops.sumsq_(gshard[cur - self.lo:a2 - self.lo], buf[0:1])
                buf[1:2].zero_()
                ops.sumsq_(gshard[a2 - self.lo:b2 - self.lo], buf[1:2])
                buf[0:1].add_(buf[1:2], alpha=1.0 / rep)
                cur = b2
            if cur < self.hi:
                ops.sumsq_(gshard[cur - self.lo:], buf[0:1])
        total = buf[0:1]
        comm.all_reduce_sum(total, self.group)
        for g in self.mp_groups:
            comm.all_reduce_sum(total, g)
        self.state[0:1].copy_(total)

    # -------------------------------------------------------------------------------- step
    def prepare(self) -> torch.Tensor:
        """Reduce gradients and leave the global squared gradient norm in ``state[0]``; returns it (a view)
        so that several optimizers over disjoint parameter sets can sum their norms before `apply`."""
        gshard = self.reduce_gradients()
        if self.n == 1:
            gshard = gshard[self.lo:self.hi]
        assert gshard.numel() == self.shard_numel
        self._sumsq(gshard)
        self._gshard = gshard
        return self.state[0:1]

    def step(self, lr: Optional[float] = None, grad_prescale: float = 1.0) -> None:
        self.prepare()
        self.apply(lr, grad_prescale)

    def apply(self, lr: Optional[float] = None, grad_prescale: float = 1.0) -> None:
        cfg = self.cfg
        lr = self.lr if lr is None else lr
        gshard = self._gshard
        ops.adam_prepare_(self.state, float(cfg.gradient_clipping or 0.0), grad_prescale, cfg.beta1, cfg.beta2)
        shadow = self.flat.flat[self.lo:self.hi]
        ops.adamw_(self.master, self.m, self.v, gshard, shadow if shadow.dtype == torch.bfloat16 else None,
                   self.state, lr, cfg.beta1, cfg.beta2, cfg.eps, cfg.weight_decay)
        if shadow.dtype != torch.bfloat16:
            shadow.copy_(self.master)
        comm.all_gather_flat(self.flat.flat, shadow, self.group)

    def grad_norm(self) -> float:
        return float(self.state[1].item())

    def update_successful(self) -> bool:
        return bool(self.state[3].item() > 0)

    # ------------------------------------------------------------------------ checkpointing
    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {"master": self.master.cpu(), "m": self.m.cpu(), "v": self.v.cpu(), "state": self.state.cpu(),
                "shard": torch.tensor([self.rank, self.n, self.shard_numel])}

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        r, n, sn = sd["shard"].tolist()
        assert (r, n, sn) == (self.rank, self.n, self.shard_numel), "optimizer shard layout mismatch"
        self.master.copy_(sd["master"])
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.state.copy_(sd["state"])

    def resync_master_from_params(self):
        """Re-derive fp32 master weights from the bf16 parameters (after loading a checkpoint)."""
        self.master.copy_(self.flat.flat[self.lo:self.hi].float())


def lr_lambda(step: int, total_steps: int, cfg: OptimizerConfig) -> float:
    """Multiplier of the base LR: warm-up then constant / linear / cosine decay to min_lr_ratio."""
    warmup = int(cfg.warmup_steps_proportion * total_steps)
    if warmup > 0 and step < warmup:
        return float(step + 1) / float(warmup)
    if cfg.lr_scheduler_type == "constant":
        return 1.0
    progress = min(1.0, max(0.0, (step - warmup) / max(1, total_steps - warmup)))
    if cfg.lr_scheduler_type == "linear":
        return cfg.min_lr_ratio + (1 - cfg.min_lr_ratio) * (1 - progress)
    if cfg.lr_scheduler_type == "cosine":
        return cfg.min_lr_ratio + (1 - cfg.min_lr_ratio) * 0.5 * (1 + math.cos(math.pi * progress))
    raise ValueError(f"unknown lr scheduler {cfg.lr_scheduler_type}")
