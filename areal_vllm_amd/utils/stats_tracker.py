"""Scoped, distributed statistics tracker.

API and key conventions of ``areal/utils/stats_tracker.py:17-310`` in the reference (scopes,
bool-mask denominators, reduce types AVG_MIN_MAX/AVG/SUM/MIN/MAX/SCALAR, ``timeperf/`` timings, named
trackers, ``export_all``). The export path differs: local partials of ALL keys are packed into three
vectors and reduced with one SUM, one MIN and one MAX all-reduce, and the result is read back with a
single device-to-host copy (the reference all-reduces and synchronises per key).
"""
import os
import time
from collections import defaultdict
from contextlib import contextmanager
from enum import Enum, auto
from threading import Lock
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from . import logging

logger = logging.getLogger("StatsTracker")


_ROCTX = os.environ.get("AREAL_ROCTX", "0") == "1"


class ReduceType(Enum):
    AVG_MIN_MAX = auto()
    AVG = auto()
    SUM = auto()
    MIN = auto()
    MAX = auto()
    SCALAR = auto()


class DistributedStatsTracker:
    def __init__(self, name: str = ""):
        self.lock = Lock()
        self.scope_stack: List[str] = []
        if name:
            self.scope_stack.append(name.strip("/"))
        self.denominators: Dict[str, str] = {}
        self.reduce_types: Dict[str, ReduceType] = {}
        self.stats = defaultdict(list)

    class Scope:
        def __init__(self, tracker, name):
            self.tracker = tracker
            self.name = name.strip("/")

        def __enter__(self):
            self.tracker.scope_stack.append(self.name)
            return self

        def __exit__(self, *exc):
            self.tracker.scope_stack.pop()

    def scope(self, name):
        return self.Scope(self, name)

    def _full(self, key):
        return "/".join(self.scope_stack + [key]) if self.scope_stack else key

    @contextmanager
    def disable_scope(self):
        tmp, self.scope_stack = self.scope_stack, []
        try:
            yield
        finally:
            self.scope_stack = tmp

    @contextmanager
    def record_timing(self, key):
        """Wall time of a loop phase under ``timeperf/<key>``. With ``AREAL_ROCTX=1`` the phase is also a
        roctx range (through the marker API PyTorch-ROCm exposes as ``torch.cuda.nvtx``), so a
        ``rocprofv3 --marker-trace`` timeline shows the kernels grouped by rollout / recompute / train /
        weight update (the reference times the same phases, ``areal/utils/stats_tracker.py:71-80``)."""
        marked = _ROCTX and torch.cuda.is_available()
        if marked:
            torch.cuda.nvtx.range_push(str(key))
        t0 = time.perf_counter()
        try:
            yield
        finally:
            if marked:
                torch.cuda.nvtx.range_pop()
            with self.lock:
                fk = f"timeperf/{key}"
                self.reduce_types[fk] = ReduceType.SCALAR
                self.stats[fk].append(time.perf_counter() - t0)

    def denominator(self, **kwargs):
        with self.lock:
            for key, value in kwargs.items():
                if not torch.is_tensor(value) or value.dtype != torch.bool:
                    raise ValueError(f"`{key}` must be a pytorch bool tensor: {getattr(value, 'dtype', type(value))}")
                if value.numel() == 0:
                    raise ValueError(f"`{key}` must be non-empty")
                fk = self._full(key)
                self.reduce_types[fk] = ReduceType.SUM
                self.stats[fk].append(value.detach().clone())

    def scalar(self, **kwargs):
        with self.lock:
            for key, value in kwargs.items():
                fk = self._full(key)
                self.reduce_types[fk] = ReduceType.SCALAR
                self.stats[fk].append(float(value))

    def stat(self, denominator: str, reduce_type: Optional[ReduceType] = None, **kwargs):
        with self.lock:
            for key, value in kwargs.items():
                if not torch.is_tensor(value) or value.dtype != torch.float:
                    raise ValueError(f"`{key}` should be a pytorch float tensor: {getattr(value, 'dtype', type(value))}")
                if value.numel() == 0:
                    raise ValueError(f"`{key}` should be non-empty")
                if reduce_type == ReduceType.SCALAR:
                    raise ValueError("Cannot use the scalar reduce type for a tensor")
                fk = self._full(key)
                dk = self._full(denominator)
                if dk not in self.stats or not self.stats[dk]:
                    raise ValueError(f"Denominator `{dk}` does not exist")
                for x, y in zip(self.stats[dk], self.stats[fk] + [value]):
                    assert x.shape == y.shape, (x.shape, y.shape)
                self.denominators[fk] = dk
                self.reduce_types[fk] = reduce_type or ReduceType.AVG_MIN_MAX
                self.stats[fk].append(value.detach().clone())

    # ------------------------------------------------------------------------------ export
    def _local_partials(self, key, device):
        """(sum, count, min, max) of one key as 0-d tensors on `device`."""
        rt = self.reduce_types.get(key, ReduceType.SCALAR)
        vals = self.stats.get(key, [])
        z = torch.zeros((), dtype=torch.float64, device=device)
        inf = torch.full((), float("inf"), dtype=torch.float64, device=device)
        if rt == ReduceType.SCALAR:
            return (torch.tensor(float(sum(vals)), dtype=torch.float64, device=device),
                    torch.tensor(float(len(vals)), dtype=torch.float64, device=device), inf, -inf)
        if not vals:
            return z, z, inf, -inf
        if key not in self.denominators:  # a denominator itself (SUM of a bool mask)
            s = sum(v.sum().to(device=device, dtype=torch.float64) for v in vals)
            return s, z + 1, inf, -inf
        dk = self.denominators[key]
        if dk not in self.stats:
            raise ValueError(f"Denominator `{dk}` not set for key `{key}`.")
        s, c, mn, mx = z, z, inf, -inf
        for v, d in zip(vals, self.stats[dk]):
            vd = v.to(torch.float64)
            s = s + torch.where(d, vd, 0.0).sum().to(device)
            c = c + d.sum().to(device=device, dtype=torch.float64)
            mn = torch.minimum(mn, torch.where(d, vd, float("inf")).min().to(device))
            mx = torch.maximum(mx, torch.where(d, vd, -float("inf")).max().to(device))
        return s, c, mn, mx

    def export(self, key=None, reduce_group=None, reset=True) -> Dict[str, float]:
        with self.lock:
            if key is not None:
                keys = [self._full(key)]
            else:
                keys = list(self.stats.keys())
                if reduce_group is not None and dist.is_initialized() and dist.get_world_size(reduce_group) > 1:
                    all_keys = [None] * dist.get_world_size(reduce_group)
                    dist.all_gather_object(all_keys, keys, group=reduce_group)
                    rts = [None] * dist.get_world_size(reduce_group)
                    dist.all_gather_object(rts, {k: v.name for k, v in self.reduce_types.items()}, group=reduce_group)
                    for d in rts:
                        for k, v in d.items():
                            self.reduce_types.setdefault(k, ReduceType[v])
                    keys = sorted(set(k for ks in all_keys for k in ks))
            device = self._device(reduce_group)
            if keys:
                parts = [self._local_partials(k, device) for k in keys]
                sums = torch.stack([p[0] for p in parts] + [p[1] for p in parts])
                mins = torch.stack([p[2] for p in parts])
                maxs = torch.stack([p[3] for p in parts])
                if reduce_group is not None and dist.is_initialized() and dist.get_world_size(reduce_group) > 1:
                    dist.all_reduce(sums, group=reduce_group)
                    dist.all_reduce(mins, op=dist.ReduceOp.MIN, group=reduce_group)
                    dist.all_reduce(maxs, op=dist.ReduceOp.MAX, group=reduce_group)
                packed = torch.cat([sums, mins, maxs]).cpu().tolist()
            else:
                packed = []
            n = len(keys)
            results: Dict[str, float] = {}
            for i, k in enumerate(keys):
                s, c, mn, mx = packed[i], packed[n + i], packed[2 * n + i], packed[3 * n + i]
                rt = self.reduce_types.get(k, ReduceType.SCALAR)
                if rt == ReduceType.AVG_MIN_MAX:
                    if c > 0:
                        results[f"{k}/avg"] = s / c
                    if mn != float("inf"):
                        results[f"{k}/min"] = mn
                    if mx != -float("inf"):
                        results[f"{k}/max"] = mx
                elif rt == ReduceType.AVG:
                    if c > 0:
                        results[k] = s / c
                elif rt == ReduceType.SUM:
                    results[k] = s
                elif rt == ReduceType.MIN:
                    if mn != float("inf"):
                        results[k] = mn
                elif rt == ReduceType.MAX:
                    if mx != -float("inf"):
                        results[k] = mx
                elif rt == ReduceType.SCALAR:
                    if c > 0:
                        results[k] = s / c
            if reset:
                if key is not None:
                    fk = keys[0]
                    self.denominators.pop(fk, None)
                    self.reduce_types.pop(fk, None)
                    self.stats.pop(fk, None)
                else:
                    self.denominators = {}
                    self.reduce_types = {}
                    self.stats = defaultdict(list)
            return results

    @staticmethod
    def _device(reduce_group):
        if reduce_group is not None and dist.is_initialized() and "nccl" in str(dist.get_backend(reduce_group)):
            return torch.device("cuda", torch.cuda.current_device())
        return torch.device("cpu")


DEFAULT_TRACKER = DistributedStatsTracker()
stat = DEFAULT_TRACKER.stat
denominator = DEFAULT_TRACKER.denominator
export = DEFAULT_TRACKER.export
scope = DEFAULT_TRACKER.scope
scalar = DEFAULT_TRACKER.scalar
record_timing = DEFAULT_TRACKER.record_timing

TRACKERS = {"": DEFAULT_TRACKER}
LOCK = Lock()


def get(name: str = "") -> DistributedStatsTracker:
    with LOCK:
        if name not in TRACKERS:
            TRACKERS[name] = DistributedStatsTracker(name)
        return TRACKERS[name]


def export_all(reduce_group=None, reset=True) -> Dict[str, float]:
    out: Dict[str, float] = {}
    dup = set()
    names = list(TRACKERS.keys())
    if reduce_group is not None and dist.is_initialized() and dist.get_world_size(reduce_group) > 1:
        # named trackers are created lazily (e.g. by the first rollout that reports a reward): every rank
        # must walk the same list in the same order or the per-tracker collectives would pair up wrongly
        gathered = [None] * dist.get_world_size(reduce_group)
        dist.all_gather_object(gathered, names, group=reduce_group)
        names = sorted(set(n for ns in gathered for n in ns))
    for name in names:
        tracker = get(name)
        x = tracker.export(reduce_group=reduce_group, reset=reset)
        dup |= set(x) & set(out)
        out.update(x)
    if dup:
        logger.warning(f"Duplicate stat keys detected: {sorted(dup)}")
    return out
