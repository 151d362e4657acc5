"""Per-model-worker sample storage and the execution of redistribution plans.

Samples stay on the device of the worker that produced them (dataset loader or a model function call);
only their metadata travels to the master. Before a call runs, every worker walks the plan the master
derived and moves the missing pieces directly between owners and consumers. Role of
``realhf/system/data_manager.py:38-455``; transfers are batched point-to-point sends over the model
workers' world group (RCCL over xGMI on GPUs, gloo in CPU runs) instead of per-step collectives.
"""
from typing import Dict, Hashable, List, Optional, Sequence

import torch
import torch.distributed as dist

from ..utils import logging
from .data_api import SequenceSample
from .redistributor import RedistribStep

logger = logging.getLogger("data-manager")

# gloo cannot send bool tensors: they travel as uint8
_WIRE = {torch.bool: torch.uint8}


class DataManager:
    def __init__(self, worker_index: int, device="cpu", group=None):
        self.worker_index = worker_index
        self.device = torch.device(device)
        self.group = group
        self.storage: Dict[Hashable, SequenceSample] = {}

    # ------------------------------------------------------------------------- storage
    def store(self, x: SequenceSample):
        """Insert or amend: items already present gain the new keys."""
        for item in x.unpack():
            i = item.ids[0]
            item.to_device(self.device)
            if i in self.storage:
                self.storage[i].update_(item)
            else:
                self.storage[i] = item

    def has_data(self, id_: Hashable, key: Optional[str] = None) -> bool:
        s = self.storage.get(str(id_))
        return s is not None and (key is None or key in s.keys)

    def get(self, ids: Sequence[Hashable], keys: Optional[Sequence[str]] = None) -> SequenceSample:
        items = [self.storage[str(i)] for i in ids]
        return SequenceSample.gather(items, keys=keys)

    def get_meta(self, ids: Sequence[Hashable], keys: Optional[Sequence[str]] = None) -> SequenceSample:
        return self.get(ids, keys).meta()

    def clear_data(self, ids: Optional[Sequence[Hashable]] = None):
        if ids is None:
            self.storage.clear()
            return
        for i in ids:
            self.storage.pop(str(i), None)

    def __len__(self):
        return len(self.storage)

    # ---------------------------------------------------------------------- redistribution
    def redistribute(self, meta: SequenceSample, plan: List[RedistribStep]):
        """Execute this worker's part of `plan`. `meta` carries lengths / shapes / dtypes of every
        (id, key) in the plan so that receivers can allocate their buffers."""
        me = self.worker_index
        mine = [s for s in plan if s.involves(me)]
        if not mine:
            return
        pos = {i: k for k, i in enumerate(meta.ids)}
        ops, recvs = [], []
        keepalive = []
        for tag, s in enumerate(plan):
            if not s.involves(me):
                continue
            dtype = meta.dtypes[s.key]
            wire = _WIRE.get(dtype, dtype)
            if s.src == me:
                buf = torch.cat([self.storage[str(i)].data[s.key] for i in s.ids], 0).to(wire).contiguous()
                keepalive.append(buf)
                for d in s.dsts:
                    if d != me:
                        ops.append(dist.P2POp(dist.isend, buf, d, group=self.group, tag=tag))
            else:
                lens = [sum(meta.seqlens[s.key][pos[str(i)]]) for i in s.ids]
                shape = (sum(lens),) + tuple(meta.trailing_shapes[s.key] or ())
                buf = torch.empty(shape, dtype=wire, device=self.device)
                ops.append(dist.P2POp(dist.irecv, buf, s.src, group=self.group, tag=tag))
                recvs.append((s, buf, lens, dtype))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for s, buf, lens, dtype in recvs:
            off = 0
            for i, n in zip(s.ids, lens):
                piece = buf[off:off + n].to(dtype)
                off += n
                k = pos[str(i)]
                item = SequenceSample(keys={s.key}, ids=[i], seqlens={s.key: [meta.seqlens[s.key][k]]},
                                      data={s.key: piece}, trailing_shapes={s.key: meta.trailing_shapes[s.key]},
                                      dtypes={s.key: dtype},
                                      metadata={mk: [mv[k]] for mk, mv in meta.metadata.items()})
                if str(i) in self.storage:
                    self.storage[str(i)].update_(item)
                else:
                    self.storage[str(i)] = item
