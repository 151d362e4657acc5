"""All-reduce over peer-mapped memory (``csrc/hip/ipc_allreduce.hip``) for decode-sized messages: one-shot for small
ones, two-shot (reduce-scatter + all-gather) from a few hundred KB on groups of four or more ranks.

A tensor-parallel decode step all-reduces ``[batch, hidden]`` bf16 twice per layer: a few hundred KB to ~2 MB, far
below the size at which a ring collective reaches link bandwidth, so its cost is launch + protocol latency. Here every
rank maps the staging regions of its peers once (hipIpc handles exchanged through the group's host backend) and one
kernel per call copies the input into the rank's own region, signals the peers with flags in THEIR memory, and sums
all regions with direct loads over xGMI - no communicator, capturable in a hipGraph, bitwise identical results on
every rank (fixed summation order). In the two-shot form a rank sums only its 1/n of every chunk and pushes the result
into the peers' result slots: (n-1)/n of the message read and written per rank instead of (n-1) messages read, at the
price of a second flag round trip. Messages beyond the staging capacity (prefill) stay on RCCL.

Role in the reference: the custom all-reduce inside its serving engines (``disable_custom_all_reduce``,
``areal/api/cli_args.py:398-454``).
"""
import ctypes
import os
from typing import List, Optional

import torch
import torch.distributed as dist

from . import _lib as L

MAX_RANKS = 8
# two-shot from this many bytes on (groups of >= TWO_SHOT_MIN_RANKS ranks; with two or three ranks it moves as many
# bytes over the links as one-shot and only adds a flag round trip)
TWO_SHOT_MIN_BYTES = int(os.environ.get("AREAL_AR_TWO_SHOT_BYTES", 1 << 20))
TWO_SHOT_MIN_RANKS = 4
# seconds a kernel waits for a peer's flag before it raises the error word (a host stall of a few seconds on one rank,
# e.g. a follower that is late to launch its plan, must not trip it)
TIMEOUT_S = float(os.environ.get("AREAL_AR_TIMEOUT_S", "60"))


class AllReduceTimeout(RuntimeError):
    """A peer did not arrive within the time limit: the data of the affected call (and everything computed from it)
    is garbage, the ranks' sequence numbers may have diverged. The engine that owns the group must stop."""


class IpcAllReduce:
    def __init__(self, group, device: torch.device, max_elems: int):
        assert device.type == "cuda"
        self.group, self.device = group, device
        self.n, self.rank = dist.get_world_size(group), dist.get_rank(group)
        assert 2 <= self.n <= MAX_RANKS, "peer-memory all-reduce serves 2..8 ranks of one node"
        self.cap = (int(max_elems) + 7) // 8 * 8
        lib = L.lib()
        lib.areal_ipc_allreduce_region_bytes.restype = ctypes.c_long
        lib.areal_ipc_allreduce_state_offset.restype = ctypes.c_long
        lib.areal_ipc_allreduce_set_timeout.restype = None
        lib.areal_ipc_allreduce_set_timeout(ctypes.c_double(TIMEOUT_S))
        self.uncached = True
        self.nbytes = int(lib.areal_ipc_allreduce_region_bytes(ctypes.c_long(self.cap)))
        with torch.cuda.device(device):
            p = ctypes.c_void_p(0)
            # uncached: peers store into this region (flags, result slots) while this device's kernels read it
            if lib.areal_ipc_alloc_uncached(ctypes.c_long(self.nbytes), ctypes.byref(p)) != 0:
                # ordinary memory: a peer's stores are not guaranteed to be seen through this device's L2. The
                # start-up self-check decides whether the kernel may be used at all; say so either way
                import logging

                logging.getLogger("ipc-allreduce").warning(
                    "uncached device memory is unavailable: the peer-memory all-reduce region is an ordinary allocation")
                self.uncached = False
                p = ctypes.c_void_p(0)
                L.check(lib.areal_ipc_alloc(ctypes.c_long(self.nbytes), ctypes.byref(p)), "ipc_alloc")
            self._ptr = p.value
            zeros = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)
            L.check(lib.areal_memcpy_d2d_async(p, ctypes.c_void_p(zeros.data_ptr()), ctypes.c_long(self.nbytes),
                                               L.stream_ptr()), "memcpy_d2d_async")
            torch.cuda.current_stream().synchronize()
            handle = (ctypes.c_ubyte * 64)()
            off, size = ctypes.c_long(0), ctypes.c_long(0)
            L.check(lib.areal_ipc_export(p, handle, ctypes.byref(off), ctypes.byref(size)), "ipc_export")
            assert off.value == 0
            descs: List = [None] * self.n
            dist.all_gather_object(descs, dict(handle=bytes(handle).hex(), nbytes=self.nbytes, cap=self.cap), group=group)
            assert all(d["cap"] == self.cap for d in descs), "ranks disagree on the staging capacity"
            self._mapped = []
            bases = (ctypes.c_void_p * MAX_RANKS)()
            for r, d in enumerate(descs):
                if r == self.rank:
                    bases[r] = self._ptr
                    continue
                raw = (ctypes.c_ubyte * 64).from_buffer_copy(bytes.fromhex(d["handle"]))
                base = ctypes.c_void_p(0)
                L.check(lib.areal_ipc_open(raw, ctypes.byref(base)), "ipc_open")
                self._mapped.append(base.value)
                bases[r] = base.value
            for r in range(self.n, MAX_RANKS):
                bases[r] = self._ptr
            self._bases = bases
        dist.barrier(group=group)  # nobody signals a peer whose region is not zeroed and mapped yet

    def fits(self, x: torch.Tensor) -> bool:
        return (x.dtype == torch.bfloat16 and x.is_cuda and x.is_contiguous() and x.numel() <= self.cap
                and x.data_ptr() % 16 == 0)

    def all_reduce_(self, x: torch.Tensor, two_shot: Optional[bool] = None) -> torch.Tensor:
        """In-place sum over the group on the current stream. Every rank must call it with the same shape (and the
        same `two_shot` choice: by default a function of size and group size only), in the same order relative to
        its other calls."""
        assert self.fits(x), "message does not fit the staging region (bf16, contiguous, <= capacity)"
        if two_shot is None:
            two_shot = self.n >= TWO_SHOT_MIN_RANKS and x.numel() * 2 >= TWO_SHOT_MIN_BYTES
        fn = L.lib().areal_ipc_allreduce_two_shot if two_shot else L.lib().areal_ipc_allreduce
        L.check(fn(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(x.data_ptr()), ctypes.c_long(x.numel()), self._bases,
                   ctypes.c_int(self.n), ctypes.c_int(self.rank), ctypes.c_long(self.cap), L.stream_ptr()),
                "ipc_allreduce")
        return x

    def self_check(self) -> bool:
        """Collective: reduce known values twice (both staging slots) and agree on whether every rank got the right
        sums. A caller falls back to the group's ordinary collective when this says no."""
        ok = True
        for rep in range(4):  # both kernels, both staging slots
            x = torch.full((min(self.cap, 4096 + 8 * rep),), float(self.rank + 1), dtype=torch.bfloat16,
                           device=self.device)
            self.all_reduce_(x, two_shot=rep >= 2)
            ok = ok and bool((x.float() == self.n * (self.n + 1) / 2).all().item())
        ok = ok and not self.error()
        votes: List = [None] * self.n
        dist.all_gather_object(votes, ok, group=self.group)
        return all(votes)

    def _error_word_ptr(self) -> int:
        return self._ptr + int(L.lib().areal_ipc_allreduce_state_offset(ctypes.c_long(self.cap))) + 8  # third state word

    def error(self) -> bool:
        """True if a call gave up waiting for a peer (its result is then garbage). Synchronises the stream."""
        word = torch.empty(1, dtype=torch.int32, device=self.device)
        L.check(L.lib().areal_memcpy_d2d_async(ctypes.c_void_p(word.data_ptr()), ctypes.c_void_p(self._error_word_ptr()),
                                               ctypes.c_long(4), L.stream_ptr()), "memcpy_d2d_async")
        return bool(word.item())

    def read_error_async(self, pinned: torch.Tensor):
        """Queue a copy of the error word into `pinned` (int32[1], pinned host memory) on the current stream: whoever
        waits for the work queued so far (a decode plan's completion event) then reads it without another
        synchronisation."""
        assert pinned.dtype == torch.int32 and pinned.is_pinned()
        L.check(L.lib().areal_memcpy_async(ctypes.c_void_p(pinned.data_ptr()), ctypes.c_void_p(self._error_word_ptr()),
                                           ctypes.c_long(4), L.stream_ptr()), "memcpy_async")

    def raise_if_error(self):
        if self.error():
            raise AllReduceTimeout("tensor-parallel all-reduce: a peer did not arrive within "
                                   f"{TIMEOUT_S:.0f} s (AREAL_AR_TIMEOUT_S)")

    def close(self):
        for b in self._mapped:
            L.lib().areal_ipc_close(ctypes.c_void_p(b))
        self._mapped = []
        if self._ptr:
            L.lib().areal_ipc_free(ctypes.c_void_p(self._ptr))
            self._ptr = 0
