"""Asynchronous sequence buffer of the master worker.

Samples (meta only: ids, keys, lengths) enter with the keys the dataset provides; model function calls
amend them with their output keys. An MFC can take a batch as soon as `n_seqs` samples own all of its
input keys; samples are handed out oldest first and leave the buffer after every MFC has consumed them
once. Contract of ``realhf/system/buffer.py:34-408``; storage is a dict keyed by sample id with a
per-RPC consumed set instead of fixed slots + readiness matrix.
"""
import asyncio
import dataclasses
import time
from typing import Dict, List, Optional, Sequence, Set, Tuple

from .data_api import SequenceSample


@dataclasses.dataclass
class _Entry:
    sample: SequenceSample
    birth_time: float
    consumed_by: Set[str] = dataclasses.field(default_factory=set)


class BufferFull(Exception):
    pass


class AsyncIOSequenceBuffer:
    def __init__(self, rpcs: Sequence, max_size: int):
        """`rpcs`: objects with `.name`, `.input_keys`, `.n_seqs` (model function call definitions)."""
        self.rpcs = {r.name: r for r in rpcs}
        self.max_size = max_size
        self._entries: Dict[str, _Entry] = {}
        self._cond: Optional[asyncio.Condition] = None
        self._birth_counter = 0

    def _c(self) -> asyncio.Condition:
        if self._cond is None:
            self._cond = asyncio.Condition()
        return self._cond

    @property
    def size(self) -> int:
        return len(self._entries)

    def _ready_ids(self, rpc_name: str) -> List[str]:
        need = set(self.rpcs[rpc_name].input_keys)
        ids = [i for i, e in self._entries.items() if need <= e.sample.keys and rpc_name not in e.consumed_by]
        ids.sort(key=lambda i: self._entries[i].birth_time)
        return ids

    async def put_batch(self, samples: List[SequenceSample], birth_times: Optional[List[float]] = None) -> List[str]:
        """Insert single-item meta samples; returns their ids."""
        async with self._c():
            if len(self._entries) + len(samples) > self.max_size:
                raise BufferFull(f"buffer holds {len(self._entries)} of {self.max_size}, cannot add {len(samples)}")
            out = []
            for j, s in enumerate(samples):
                assert s.bs == 1, "the buffer stores single items"
                self._birth_counter += 1
                t = birth_times[j] if birth_times is not None else time.monotonic() + 1e-9 * self._birth_counter
                self._entries[s.ids[0]] = _Entry(sample=s, birth_time=t)
                out.append(s.ids[0])
            self._c().notify_all()
            return out

    async def amend_batch(self, ids: List[str], samples: List[SequenceSample]) -> None:
        """Add the output keys of an MFC to stored samples."""
        async with self._c():
            for i, s in zip(ids, samples):
                self._entries[i].sample.update_(s)
            self._c().notify_all()

    async def get_batch_for_rpc(self, rpc, timeout: Optional[float] = None) -> Tuple[List[str], SequenceSample]:
        """Wait until `rpc.n_seqs` samples are ready for `rpc`, mark them consumed and return
        (ids, gathered meta sample restricted to the rpc's input keys)."""
        name = rpc.name

        def ready():
            return len(self._ready_ids(name)) >= rpc.n_seqs

        async with self._c():
            if timeout is None:
                await self._c().wait_for(ready)
            else:
                await asyncio.wait_for(self._c().wait_for(ready), timeout)
            ids = self._ready_ids(name)[: rpc.n_seqs]
            for i in ids:
                self._entries[i].consumed_by.add(name)
            sample = SequenceSample.gather([self._entries[i].sample for i in ids], keys=rpc.input_keys)
            done = [i for i in ids if self._entries[i].consumed_by >= set(self.rpcs)]
            for i in done:
                self._entries.pop(i)
            return ids, sample

    def n_ready(self, rpc_name: str) -> int:
        return len(self._ready_ids(rpc_name))
