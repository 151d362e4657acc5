"""ZeRO-style sharded AdamW over flat parameter / gradient buffers.

MI355X-first choice: with 288 GB of HBM per GPU the bf16 parameters and fp32 gradient accumulators are
kept replicated (ZeRO-1) and only the fp32 master weights and Adam moments are sharded over the
data-parallel group. A step is then a handful of LARGE reduce-scatters of the flat fp32 gradient buffer and as many
all-gathers of the flat bf16 parameter buffer (bandwidth-efficient xGMI transfers) instead of the per-layer
all-gather / reduce-scatter traffic of FSDP2 that the reference uses (``areal/utils/fsdp.py:253-285``).

Ownership is block-cyclic: the flat buffer is cut into `n_buckets` contiguous buckets and every bucket into one
piece per rank; a rank owns its piece of EVERY bucket. A bucket is therefore exactly the input of one
``reduce_scatter_tensor`` (and the output of one ``all_gather_into_tensor``), and because the backward pass finishes
gradients from the end of the buffer to its start, the reduce-scatter of bucket g can leave - asynchronously, on the
communicator's stream - while the layers below it still compute (`arm_overlap`). Buckets are launched strictly in
descending index order on every rank, whatever order their parameters report in, so the ranks' collective sequences
always match; whatever has not left when the backward ends leaves then, in the same order. A 7B model costs 14 GB
(bf16 params) + 28 GB (fp32 grads) replicated plus 84/dp GB of sharded optimizer state per GPU; larger models add
tensor / pipeline parallelism.

The whole step (norm -> clip coefficient -> non-finite check -> AdamW -> bf16 shadow) runs on device
with no host synchronisation; the gradient norm is read back lazily for logging.
"""
import bisect
import math
import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from .. import ops
from ..api.cli_args import OptimizerConfig
from ..models.transformer import FlatParams
from . import comm

# buckets per flat buffer: the last one (embeddings, first layers) is reduced after the backward has ended, so its
# share of the traffic, 1 / N_BUCKETS, is what the overlap cannot hide
N_BUCKETS = int(os.environ.get("AREAL_ZERO_BUCKETS", "8"))


class _StreamWork:
    """Stand-in for a communicator's work handle: `wait()` orders the current stream after what a side stream holds."""

    def __init__(self, stream):
        self.event = stream.record_event()

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


class ShardedAdamW:
    def __init__(self, flat: FlatParams, cfg: OptimizerConfig, dp_group=None, mp_groups: Tuple = (),
                 replicated_ranges: Optional[List[Tuple[int, int, int]]] = None, n_buckets: Optional[int] = None,
                 peer=None):
        """`dp_group`: group over which parameters are replicated (gradients are summed over it).
        `mp_groups`: model-parallel groups (tp, pp) over which the squared gradient norm is summed.
        `replicated_ranges`: (lo, hi, rep) flat ranges of parameters held by `rep` ranks of the tp group
        (norm weights, embeddings, KV projections when tp exceeds the KV heads); their norm contribution
        is counted once.
        `peer`: an `ops.peer_collectives.PeerCollectives` over `dp_group` used by nobody else: bucket
        reduce-scatters and parameter all-gathers then run as peer-memory kernels on a side stream of this
        object instead of the communicator's."""
        self.flat = flat
        self.cfg = cfg
        self.group = dp_group
        self.mp_groups = [g for g in mp_groups if g is not None and comm.group_size(g) > 1]
        self.n = comm.group_size(dp_group)
        self.rank = comm.group_rank(dp_group)
        self.solo = comm.trivial(dp_group)  # nothing to reduce or gather: the fp32 gradient buffer is the shard
        self.peer = peer if (peer is not None and flat.flat.is_cuda and not self.solo) else None
        self._side = torch.cuda.Stream(device=flat.flat.device) if self.peer is not None else None
        assert flat.numel % self.n == 0, "flat buffer must be padded to a multiple of the dp size"
        self.shard_numel = flat.numel // self.n
        # ---- block-cyclic layout: bucket g = flat[n * cut[g] : n * cut[g + 1]], this rank's piece of it starts
        # `rank` piece lengths into the bucket; in the rank's shard (master, moments, reduced gradients) the pieces
        # follow each other, piece g at shard[cut[g] : cut[g + 1]]
        G = 1 if self.solo else max(1, min(int(n_buckets or N_BUCKETS), self.shard_numel // 8 or 1))
        cuts = sorted({min(self.shard_numel, (g * self.shard_numel // G + 7) // 8 * 8) for g in range(G)} | {0})
        self.cuts: List[int] = cuts + [self.shard_numel]
        self.buckets: List[Tuple[int, int]] = [(self.n * a, self.n * b) for a, b in zip(self.cuts, self.cuts[1:])]
        self.pieces: List[Tuple[int, int]] = [(lo + self.rank * (b - a), lo + (self.rank + 1) * (b - a))
                                              for (lo, _), a, b in zip(self.buckets, self.cuts, self.cuts[1:])]
        dev = flat.flat.device
        self.master = torch.cat([flat.flat[a:b].float() for a, b in self.pieces])
        self.m = torch.zeros_like(self.master)
        self.v = torch.zeros_like(self.master)
        self.state = torch.zeros(ops.optim.STATE_SIZE, dtype=torch.float32, device=dev)
        self.grad_shard = (torch.empty(self.shard_numel, dtype=torch.float32, device=dev)
                           if not self.solo else None)
        self.replicated_ranges = sorted(replicated_ranges or [])
        self._norm_buf = torch.zeros(2, dtype=torch.float32, device=dev)
        self.lr = cfg.lr
        # ---- overlapped reduction
        self._bucket_of: Dict[int, List[int]] = {}  # id(parameter) -> buckets its flat range touches
        self._bucket_params: List[int] = [0] * len(self.buckets)  # gradient folds a bucket waits for
        self._events: Dict[int, int] = {}  # id(parameter) -> folds it reports per backward pass
        self._got: Dict[int, int] = {}  # ... and how many it has reported since the overlap was armed
        self._armed = False
        self._pending: List[int] = []
        self._works: List = []
        self._next = -1  # next bucket to launch (descending)
        self.overlapped_buckets = 0  # buckets that left during a backward pass (all steps so far)
        if not self.solo:
            starts = [b[0] for b in self.buckets]
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
                self._events[id(p)] = events
                for i in ids:
                    self._bucket_params[i] += events
                p._after_fold = self._param_done

    def owned_ranges(self) -> List[Tuple[int, int]]:
        """Flat ranges whose master weights / moments this rank holds, in the order they sit in its shard."""
        return list(self.pieces)

    def _shard_piece(self, t: torch.Tensor, g: int) -> torch.Tensor:
        return t[self.cuts[g]:self.cuts[g + 1]]

    # ------------------------------------------------------------- overlapped gradient reduction
    def arm_overlap(self):
        """Call right before the LAST backward pass of a step: from now on a bucket's reduce-scatter leaves as
        soon as every parameter touching it - and every bucket above it - has folded its gradient into the fp32
        buffer."""
        if self.solo:
            return
        self._armed = True
        self._pending = list(self._bucket_params)
        self._got = {}
        self._next = len(self.buckets) - 1
        self._works = []

    def _launch(self, g: int, async_op: bool):
        lo, hi = self.buckets[g]
        src, out = self.flat.main_grad[lo:hi], self._shard_piece(self.grad_shard, g)
        if self.peer is not None and (hi - lo) // self.n % 4 == 0:
            # everything folded into the bucket so far is queued on the compute stream: the side stream starts there
            from ..ops.peer_collectives import OVERLAP_BLOCKS

            cur = torch.cuda.current_stream()
            self._side.wait_stream(cur)
            with torch.cuda.stream(self._side):
                self.peer.reduce_scatter_into(out, src, max_blocks=OVERLAP_BLOCKS if async_op else 0)
            if async_op:
                self._works.append((_StreamWork(self._side), None))
            else:
                cur.wait_stream(self._side)
        elif comm._is_nccl(src):
            w = dist.reduce_scatter_tensor(out, src, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            if async_op:
                self._works.append((w, None))
        else:  # host backend: no reduce-scatter; sum the bucket in place, keep this rank's piece
            w = dist.all_reduce(src, group=self.group, async_op=async_op)
            a, b = self.pieces[g]
            if async_op:
                self._works.append((w, (out, self.flat.main_grad[a:b])))
            else:
                out.copy_(self.flat.main_grad[a:b])

    def _drain_ready(self) -> int:
        """Launch, in descending order, every bucket that is final and has no unfinished bucket above it."""
        k = 0
        while self._next >= 0 and self._pending[self._next] <= 0:
            self._launch(self._next, async_op=True)
            self._next -= 1
            k += 1
        return k

    def _param_done(self, p):
        if not self._armed:
            return
        k = self._got.get(id(p), 0) + 1
        # one report too many would stand in for another parameter's missing one and send the bucket off early
        assert k <= self._events.get(id(p), 0), "a parameter folded its gradient more often than its buckets expect"
        self._got[id(p)] = k
        for i in self._bucket_of.get(id(p), ()):
            self._pending[i] -= 1
        self.overlapped_buckets += self._drain_ready()

    # ------------------------------------------------------------------ gradient reduction
    def reduce_gradients(self):
        g = self.flat.main_grad
        if self.solo:
            return g
        if self._armed:
            # buckets some of whose parameters got no gradient in the last backward (a vision tower on a rank
            # without images, the padding at the end of the buffer), and everything below them, have not left:
            # they leave now, in the same descending order on every rank; then wait for all
            self._armed = False
            while self._next >= 0:
                self._launch(self._next, async_op=True)
                self._next -= 1
            for w, post in self._works:
                w.wait()
                if post is not None:
                    post[0].copy_(post[1])
            self._works = []
            return self.grad_shard
        for i in range(len(self.buckets) - 1, -1, -1):
            self._launch(i, async_op=False)
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
            for gi, (lo, hi) in enumerate(self.pieces):  # flat range [lo, hi) sits at shard[cuts[gi]:]
                part = self._shard_piece(gshard, gi)
                cur = lo
                for (a, b, rep) in self.replicated_ranges:
                    a2, b2 = max(a, lo), min(b, hi)
                    if a2 >= b2:
                        continue
                    if cur < a2:
                        ops.sumsq_(part[cur - lo:a2 - lo], buf[0:1])
                    buf[1:2].zero_()
                    ops.sumsq_(part[a2 - lo:b2 - lo], buf[1:2])
                    buf[0:1].add_(buf[1:2], alpha=1.0 / rep)
                    cur = b2
                if cur < hi:
                    ops.sumsq_(part[cur - lo:], buf[0:1])
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
        for g, (a, b) in enumerate(self.pieces):
            shadow = self.flat.flat[a:b]
            master = self._shard_piece(self.master, g)
            ops.adamw_(master, self._shard_piece(self.m, g), self._shard_piece(self.v, g), self._shard_piece(gshard, g),
                       shadow if shadow.dtype == torch.bfloat16 else None, self.state, lr, cfg.beta1, cfg.beta2,
                       cfg.eps, cfg.weight_decay)
            if shadow.dtype != torch.bfloat16:
                shadow.copy_(master)
            # the new parameters of bucket g travel while the next piece is still being updated
            lo, hi = self.buckets[g]
            if self.peer is not None and (b - a) * shadow.element_size() % 16 == 0:
                self._side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self._side):
                    self.peer.all_gather_into(self.flat.flat[lo:hi], shadow)
            else:
                comm.all_gather_flat(self.flat.flat[lo:hi], shadow, self.group)
        if self.peer is not None:
            torch.cuda.current_stream().wait_stream(self._side)

    def step_stats(self) -> Tuple[float, bool]:
        """(gradient norm, was the update applied) of the last step: ONE read-back of the device-side state."""
        st = self.state[:4].tolist()
        return float(st[1]), bool(st[3] > 0)

    def grad_norm(self) -> float:
        return float(self.state[1].item())

    def update_successful(self) -> bool:
        return bool(self.state[3].item() > 0)

    # ------------------------------------------------------------------------ checkpointing
    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {"master": self.master.cpu(), "m": self.m.cpu(), "v": self.v.cpu(), "state": self.state.cpu(),
                "shard": torch.tensor([self.rank, self.n, self.shard_numel]), "cuts": torch.tensor(self.cuts)}

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        r, n, sn = sd["shard"].tolist()
        assert (r, n, sn) == (self.rank, self.n, self.shard_numel), "optimizer shard layout mismatch"
        cuts = sd["cuts"].tolist() if "cuts" in sd else [0, sn]  # files written before buckets: one contiguous piece
        assert n == 1 or cuts == self.cuts, "optimizer shard was written with another bucket layout (AREAL_ZERO_BUCKETS)"
        self.master.copy_(sd["master"])
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.state.copy_(sd["state"])

    def resync_master_from_params(self):
        """Re-derive fp32 master weights from the bf16 parameters (after loading a checkpoint)."""
        for g, (a, b) in enumerate(self.pieces):
            self._shard_piece(self.master, g).copy_(self.flat.flat[a:b].float())


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
