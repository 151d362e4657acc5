"""ZeRO-3: parameters, gradients and optimizer state all sharded over the data-parallel group (of every tensor-parallel
rank, when the model is also split over tp).

The default on MI355X is ZeRO-1 (``parallel/zero.py``): 288 GB of HBM hold replicated bf16 weights and fp32
gradient accumulators of everything up to ~30B parameters, and a step costs two large collectives. This module
is for what does not fit that way: every rank keeps 1/n of each parameter at rest; a transformer block's
weights exist in full only while that block computes. Semantics of the reference's FSDP2 path
(``areal/utils/fsdp.py:253-285``: ``fully_shard`` per decoder layer, ``reshard_after_forward``, CPU offload
policy; ``areal/engine/fsdp_engine.py``):

* unit = one decoder block; everything outside the blocks (embeddings, final norm, LM / value head, vision
  tower) forms one more unit that stays whole for the duration of a micro-batch;
* before a block runs (forward, backward, or the recomputation of gradient checkpointing) its shards are
  all-gathered into a buffer whose storage is allocated for the occasion; the next block's gather is started
  at the same time on a communication stream, so the xGMI transfer of block i+1 hides behind the GEMMs of
  block i; after the block the storage is dropped again (`reshard_after_forward=False` keeps it until the
  backward is through);
* as soon as the last gradient of a block has been produced, its fp32 gradient buffer is reduce-scattered and
  added to this rank's gradient shard, and the buffer is freed;
* AdamW runs on the shards (fp32 master weights and moments, 12 bytes per parameter / n per rank) with the same
  fused device step as ZeRO-1; `offload_optimizer` keeps those three buffers in pinned host memory and streams
  them through the device in slabs.

Parameters are `.data` views of their unit's buffer; dropping the storage (``resize_(0)``) keeps the views -
and the tensors autograd saved for the backward - alive, and gathering into the same storage makes them valid
again.
"""
import math
from typing import Any, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from .. import ops
from ..api.cli_args import OptimizerConfig
from . import comm

ALIGN = 16  # elements: parameter offsets inside a unit stay 32-byte aligned for bf16


class ShardMap:
    """Where the elements of a model's (virtual) contiguous parameter buffer live once ZeRO-3 has cut every unit over
    `n` ranks: computed from names and shapes alone (a model on the meta device will do), with the arithmetic of
    `Zero3.__init__` - the test suite holds the two against each other."""

    def __init__(self, model, n: int):
        import numpy as np

        self.n = n
        units: Dict[str, List] = {}
        order: List[str] = [f"layers.{i}" for i in range(len(model.layers))] + ["outer"]
        sizes = {u: 0 for u in order}
        rows = []  # (flat offset, numel, unit index, offset in unit)
        seen = set()
        for name, p in model.named_parameters():
            if id(p) in seen:
                continue
            seen.add(id(p))
            parts = name.split(".")
            u = ".".join(parts[:2]) if parts[0] == "layers" else "outer"
            rows.append((model.flat.offsets[name][0], p.numel(), order.index(u), sizes[u]))
            sizes[u] += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        quantum = n * ALIGN
        self.unit_numel = [(sizes[u] + quantum - 1) // quantum * quantum for u in order]
        self.shard_off = list(np.cumsum([0] + [x // n for x in self.unit_numel[:-1]]))
        self.shard_numel = sum(x // n for x in self.unit_numel)
        rows.sort()
        self.rows = np.asarray(rows, dtype=np.int64)

    def translate(self, src_intervals, dst_intervals):
        """[n, 2] ranges of the contiguous buffer (each inside one parameter) with their destination ranges ->
        {rank: int64 [m, 3] (offset in that rank's shard buffer, destination offset, length)}."""
        import numpy as np

        out: Dict[int, List] = {}
        starts = self.rows[:, 0]
        for (a, b), (d0, _d1) in zip(np.asarray(src_intervals).tolist(), np.asarray(dst_intervals).tolist()):
            i = int(np.searchsorted(starts, a, side="right")) - 1
            o_f, numel, ui, off = self.rows[i].tolist()
            assert o_f <= a and b <= o_f + numel, "a source range must lie inside one parameter"
            per = self.unit_numel[ui] // self.n
            lo, hi = off + a - o_f, off + b - o_f
            for z in range(lo // per, (hi - 1) // per + 1):
                x0, x1 = max(lo, z * per), min(hi, (z + 1) * per)
                out.setdefault(z, []).append((self.shard_off[ui] + x0 - z * per, d0 + (x0 - lo), x1 - x0))
        return {z: np.asarray(v, dtype=np.int64) for z, v in out.items()}


def shard_map(model, n: int) -> ShardMap:
    return ShardMap(model, n)


class _Unit:
    def __init__(self, name: str):
        self.name = name
        self.params: List[Tuple[str, torch.nn.Parameter, int]] = []  # (name, parameter, offset in the unit)
        self.numel = 0  # padded to a multiple of n * ALIGN
        self.full: Optional[torch.Tensor] = None  # [numel], storage present only while gathered
        self.gathered = False
        self.event = None
        self.shard_off = 0  # offset of this unit's shard in the rank's flat shard buffers
        self.grad_full: Optional[torch.Tensor] = None
        self.pending = 0


class Zero3:
    def __init__(self, model, cfg: Optional[OptimizerConfig], group=None, reshard_after_forward: bool = True,
                 offload_optimizer: bool = False, offload_slab: int = 1 << 26, tp_group=None,
                 tp_replicated: Optional[List[Tuple[str, int, int, int]]] = None,
                 tp_partial_grads: Optional[List[Tuple[str, int, int, int]]] = None):
        """`group`: the ranks that shard this rank's parameters (data x context parallel). With tensor parallelism
        every tp rank shards ITS slice of the model over its own `group`; `tp_group` joins the ranks whose slices
        differ. `tp_replicated`: (parameter, first element, end element, rep) - element ranges that `rep`
        neighbouring tp ranks hold alike (they count once in the gradient norm); `tp_partial_grads`: those of them
        whose gradient on one rank is only a partial sum, to be summed inside every set of `rep` neighbours."""
        self.model = model
        self.cfg = cfg
        self.group = group
        self.tp_group = tp_group if comm.group_size(tp_group) > 1 else None
        self.n = comm.group_size(group)
        self.solo = comm.trivial(group)
        self.rank = comm.group_rank(group)
        self.reshard_after_forward = reshard_after_forward
        self.offload = offload_optimizer
        self.offload_slab = offload_slab
        self.in_backward = False
        self.lr = cfg.lr if cfg is not None else 0.0
        dev = next(model.parameters()).device
        self.device, self.dtype = dev, model.dtype
        self.cuda = dev.type == "cuda"
        self.comm_stream = torch.cuda.Stream(device=dev) if self.cuda else None
        self._events: List[Any] = []

        # ---- units: decoder blocks in order, then everything else
        by_name: Dict[str, _Unit] = {}
        self.units: List[_Unit] = []
        for i in range(len(model.layers)):
            by_name[f"layers.{i}"] = _Unit(f"layers.{i}")
            self.units.append(by_name[f"layers.{i}"])
        self.outer = _Unit("outer")
        self.units.append(self.outer)
        seen = set()
        for name, p in model.named_parameters():
            if id(p) in seen:
                continue
            seen.add(id(p))
            parts = name.split(".")
            u = by_name[".".join(parts[:2])] if parts[0] == "layers" else self.outer
            u.params.append((name, p, u.numel))
            u.numel += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self._unit_of: Dict[int, Tuple[_Unit, int]] = {}
        soff = 0
        for u in self.units:
            quantum = self.n * ALIGN
            u.numel = (u.numel + quantum - 1) // quantum * quantum
            u.shard_off = soff
            soff += u.numel // self.n
            for _name, p, off in u.params:
                self._unit_of[id(p)] = (u, off)
        self.shard_numel = soff

        # ---- move the weights: unit buffer <- parameters, shard <- own slice, parameters -> views of the buffer
        self.param_shard = torch.zeros(self.shard_numel, dtype=self.dtype, device=dev)
        for u in self.units:
            u.full = torch.zeros(u.numel, dtype=self.dtype, device=dev)
            for _name, p, off in u.params:
                view = u.full[off:off + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                if hasattr(p, "main_grad"):
                    del p.main_grad
            u.gathered = True
            self._shard_view(u).copy_(u.full.view(self.n, -1)[self.rank])
        if model.flat is not None:  # the contiguous ZeRO-1 buffer is gone; names / shapes stay for checkpoints
            model.flat.flat = torch.empty(0, dtype=self.dtype, device=dev)
            model.flat.main_grad = None
        for u in self.units:
            self._release(u)

        # ---- optimizer state on the shards
        self.grad_shard = torch.zeros(self.shard_numel, dtype=torch.float32, device=dev)
        self.state = torch.zeros(ops.optim.STATE_SIZE, dtype=torch.float32, device=dev)
        self._norm_buf = torch.zeros(1, dtype=torch.float32, device=dev)
        self.master = self.m = self.v = None
        if cfg is not None:
            host = dict(device="cpu", pin_memory=self.cuda) if self.offload else dict(device=dev)
            self.master = torch.empty(self.shard_numel, dtype=torch.float32, **host)
            self.master.copy_(self.param_shard.float())
            self.m = torch.zeros(self.shard_numel, dtype=torch.float32, **host)
            self.v = torch.zeros(self.shard_numel, dtype=torch.float32, **host)

        # ---- tensor parallelism: where the replicated / partially summed parameters sit in this rank's shard
        self.tp_rank = comm.group_rank(self.tp_group)
        self._rep_ranges = self._shard_ranges(tp_replicated or []) if self.tp_group is not None else []
        self._partial_ranges = self._shard_ranges(tp_partial_grads or []) if self.tp_group is not None else []

        # ---- hooks
        for i, blk in enumerate(model.layers):
            blk.register_forward_pre_hook(lambda _m, _a, i=i: self._pre_forward(i))
            blk.register_forward_hook(lambda _m, _a, _o, i=i: self._post_forward(i))
            blk.register_full_backward_pre_hook(lambda _m, _g, i=i: self._pre_backward(i))
        for u in self.units:
            for _name, p, _off in u.params:
                if p.requires_grad and cfg is not None:
                    p.register_post_accumulate_grad_hook(self._grad_ready)

    def _shard_ranges(self, spans: List[Tuple[str, int, int, int]]) -> List[Tuple[int, int, int]]:
        """(lo, hi, rep) ranges of the rank's flat shard buffers that hold the given element spans of parameters.
        The layout is the same on every tp rank (same shapes), so equal ranges of tp peers hold the same elements."""
        by_name: Dict[str, List[Tuple[int, int, int]]] = {}
        for name, a, b, rep in spans:
            by_name.setdefault(name, []).append((a, b, rep))
        out = []
        for u in self.units:
            per = u.numel // self.n
            s0 = self.rank * per
            for name, p, off in u.params:
                for pa, pb, rep in by_name.get(name, ()):
                    a, b = max(off + pa, s0), min(off + pb, s0 + per)
                    if a < b:
                        out.append((u.shard_off + a - s0, u.shard_off + b - s0, rep))
        return out

    # ---------------------------------------------------------------------------- storage
    def _shard_view(self, u: _Unit) -> torch.Tensor:
        return self.param_shard[u.shard_off:u.shard_off + u.numel // self.n]

    def _gather(self, u: _Unit):
        """Start the all-gather of a unit (no-op when it is whole)."""
        if u.gathered:
            return
        u.full.untyped_storage().resize_(u.numel * u.full.element_size())
        if self.cuda:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                comm.all_gather_flat(u.full, self._shard_view(u), self.group)
                u.event = torch.cuda.Event()
                u.event.record(self.comm_stream)
        else:
            comm.all_gather_flat(u.full, self._shard_view(u), self.group)
        u.gathered = True

    def _need(self, u: _Unit):
        """The unit's weights are about to be read on the current stream."""
        self._gather(u)
        if u.event is not None:
            torch.cuda.current_stream().wait_event(u.event)
            u.event = None

    def _release(self, u: _Unit):
        if u.gathered:
            if u.event is not None:  # a prefetch nobody consumed: its writes must land before the memory is reused
                torch.cuda.current_stream().wait_event(u.event)
            u.full.untyped_storage().resize_(0)  # allocated and last used on the compute stream
            u.gathered = False
            u.event = None

    # ------------------------------------------------------------------------------ hooks
    def _pre_forward(self, i: int):
        self._need(self.units[i])
        if not self.in_backward and i + 1 < len(self.model.layers):
            self._gather(self.units[i + 1])  # prefetch: overlaps with this block's compute

    def _post_forward(self, i: int):
        if self.reshard_after_forward and not self.in_backward:
            self._release(self.units[i])

    def _pre_backward(self, i: int):
        self._need(self.units[i])
        if i > 0:
            self._gather(self.units[i - 1])

    def _grad_ready(self, p: torch.nn.Parameter):
        if p.grad is None:
            return
        u, off = self._unit_of[id(p)]
        if u.grad_full is None:
            u.grad_full = torch.zeros(u.numel, dtype=torch.float32, device=self.device)
            u.pending = sum(1 for _n, q, _o in u.params if q.requires_grad)
        u.grad_full[off:off + p.numel()].add_(p.grad.reshape(-1))
        p.grad = None
        u.pending -= 1
        if u.pending == 0 and u is not self.outer:
            self._reduce(u)
            if self.in_backward:
                self._release(u)

    def _reduce(self, u: _Unit):
        """grad shard += reduce-scatter(sum) of the unit's fp32 gradients; the full buffer is freed."""
        g, u.grad_full = u.grad_full, None
        if g is None:
            return
        dst = self.grad_shard[u.shard_off:u.shard_off + u.numel // self.n]
        if self.cuda and not self.solo:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                tmp = torch.empty_like(dst)
                comm.reduce_scatter_sum(tmp, g, self.group)
                dst.add_(tmp)
                ev = torch.cuda.Event()
                ev.record(self.comm_stream)
            g.record_stream(self.comm_stream)
            self._events.append(ev)
        elif not self.solo:
            tmp = torch.empty_like(dst)
            comm.reduce_scatter_sum(tmp, g, self.group)
            dst.add_(tmp)
        else:
            dst.add_(g[: dst.numel()])

    # -------------------------------------------------------------------- micro-batch scope
    def begin_microbatch(self):
        """Embeddings / head / final norm are whole from here to `end_microbatch`; the first block is on its way."""
        self._need(self.outer)
        if self.model.layers:
            self._gather(self.units[0])

    def end_microbatch(self):
        self.in_backward = False
        for u in self.units:
            if u.grad_full is not None:  # the outer unit, and blocks some of whose parameters got no gradient
                self._reduce(u)
            self._release(u)

    class _Full:
        def __init__(self, z):
            self.z = z

        def __enter__(self):
            for u in self.z.units:
                self.z._need(u)
            return self.z

        def __exit__(self, *exc):
            for u in self.z.units:
                self.z._release(u)
            return False

    def full_params(self):
        """Context in which every parameter is whole (checkpoint writers, weight pushes): costs the memory of
        the unsharded model for its duration."""
        return Zero3._Full(self)

    def reshard_from_full(self):
        """Inside `full_params()`, after the parameters were overwritten (checkpoint load): refresh the shards."""
        for u in self.units:
            assert u.gathered
            self._shard_view(u).copy_(u.full.view(self.n, -1)[self.rank])
        self.resync_master_from_params()

    # ---------------------------------------------------------------------------- optimizer
    def zero_grad(self):
        self.grad_shard.zero_()
        for u in self.units:
            u.grad_full = None

    def _wait_reductions(self):
        for ev in self._events:
            torch.cuda.current_stream().wait_event(ev)
        self._events.clear()

    def step(self, lr: Optional[float] = None, grad_prescale: float = 1.0) -> None:
        cfg = self.cfg
        lr = self.lr if lr is None else lr
        self._wait_reductions()
        if self._partial_ranges:  # e.g. norm weights under sequence parallelism: every tp rank saw its own tokens
            tp = comm.group_size(self.tp_group)
            for rep in sorted({r for _, _, r in self._partial_ranges}):
                sel = [(a, b) for a, b, r in self._partial_ranges if r == rep]
                packed = torch.cat([self.grad_shard[a:b] for a, b in sel])
                if rep == tp:
                    comm.all_reduce_sum(packed, self.tp_group)
                else:  # K/V projections shared by `rep` neighbouring ranks (more tp ranks than KV heads)
                    s0 = self.tp_rank // rep * rep
                    packed = comm.all_gather_cat(packed.unsqueeze(0), self.tp_group)[s0:s0 + rep].sum(0)
                o = 0
                for a, b in sel:
                    self.grad_shard[a:b].copy_(packed[o:o + b - a])
                    o += b - a
        self._norm_buf.zero_()
        ops.sumsq_(self.grad_shard, self._norm_buf)
        for a, b, rep in self._rep_ranges:  # held by `rep` tp ranks: count once over the tp group
            self._norm_buf.sub_(self.grad_shard[a:b].pow(2).sum() * (1.0 - 1.0 / rep))
        comm.all_reduce_sum(self._norm_buf, self.group)
        if self.tp_group is not None:
            comm.all_reduce_sum(self._norm_buf, self.tp_group)
        self.state[0:1].copy_(self._norm_buf)
        ops.adam_prepare_(self.state, float(cfg.gradient_clipping or 0.0), grad_prescale, cfg.beta1, cfg.beta2)
        bf16 = self.param_shard.dtype == torch.bfloat16
        if not self.offload:
            ops.adamw_(self.master, self.m, self.v, self.grad_shard, self.param_shard if bf16 else None, self.state, lr,
                       cfg.beta1, cfg.beta2, cfg.eps, cfg.weight_decay)
            if not bf16:
                self.param_shard.copy_(self.master)
            return
        # optimizer state lives on the host: stream it through the device slab by slab
        for lo in range(0, self.shard_numel, self.offload_slab):
            hi = min(self.shard_numel, lo + self.offload_slab)
            w, m, v = (t[lo:hi].to(self.device, non_blocking=True) for t in (self.master, self.m, self.v))
            shadow = self.param_shard[lo:hi]
            ops.adamw_(w, m, v, self.grad_shard[lo:hi], shadow if bf16 else None, self.state, lr, cfg.beta1, cfg.beta2,
                       cfg.eps, cfg.weight_decay)
            if not bf16:
                shadow.copy_(w)
            self.master[lo:hi].copy_(w, non_blocking=True)
            self.m[lo:hi].copy_(m, non_blocking=True)
            self.v[lo:hi].copy_(v, non_blocking=True)
        if self.cuda:
            torch.cuda.current_stream().synchronize()  # the pinned buffers are complete before anyone reads them

    def grad_norm(self) -> float:
        return float(self.state[1].item())

    def update_successful(self) -> bool:
        return bool(self.state[3].item() > 0)

    def resync_master_from_params(self):
        if self.master is not None:
            self.master.copy_(self.param_shard.float())

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

    # --------------------------------------------------------------------------- accounting
    def resident_param_bytes(self) -> int:
        """Bytes of parameter storage this rank holds right now (shards + whatever is gathered)."""
        e = self.param_shard.element_size()
        return self.shard_numel * e + sum(u.full.untyped_storage().size() for u in self.units)

    def full_param_bytes(self) -> int:
        return sum(u.numel for u in self.units) * self.param_shard.element_size()
