"""Trainer collectives over peer-mapped memory (``csrc/hip/peer_collectives.hip``).

One kernel - an all-to-all of equal pieces with strided rows on both sides, pushed straight into the peers' staging
slots - gives the collectives a training step issues inside one node:

* ``seq_to_head`` / ``head_to_seq``: the Ulysses exchange, with the [tokens, ranks, heads] <-> [ranks, tokens, heads]
  transposes folded into the kernel's addressing (the communicator path needs a ``.contiguous()`` copy on each side),
* ``all_gather_rows`` / ``reduce_scatter_rows``: sequence parallelism around tensor-parallel linears,
* ``all_gather_into`` / ``reduce_scatter_into``: parameter all-gather and fp32 gradient reduce-scatter of the sharded
  optimizer, chunked through the staging slot.

Every call is one kernel launch on the current stream with no host synchronisation, so it can be captured and it
overlaps with other streams' kernels (``max_blocks`` bounds the compute units it occupies). All calls on one
``PeerCollectives`` object must come from one stream per rank, in the same order on every rank. Sums are taken in rank
order: every rank gets the same bits, run after run.

The object is opt-in (``AREAL_PEER_COLLECTIVES=1`` or the engine's ``peer_collectives`` option) and checks itself at
start-up; groups it cannot serve (more than 8 ranks, ranks on several nodes, mapping failures) stay on RCCL. Its
bandwidth over xGMI has not been measured: the only hardware it has run on is one GPU shared by the ranks' processes.

Role in the reference: NCCL collectives of ``areal/utils/ulysses.py:125-149``,
``realhf/impl/model/parallelism/tensor_parallel/modules.py:242-402`` and ``areal/utils/fsdp.py:253-285``.
"""
import ctypes
import logging
import os
from typing import List, Optional

import torch
import torch.distributed as dist

from . import _lib as L

MAX_RANKS = 8
ENABLED = os.environ.get("AREAL_PEER_COLLECTIVES", "0") == "1"
# bytes of one staging slot (the region holds two): one exchange moves at most this much INTO a rank
SLOT_BYTES = int(os.environ.get("AREAL_PEER_SLOT_MB", "256")) << 20
TIMEOUT_S = float(os.environ.get("AREAL_PEER_TIMEOUT_S", "120"))
# workgroups of an exchange that runs beside compute kernels (gradient buckets during the backward pass)
OVERLAP_BLOCKS = int(os.environ.get("AREAL_PEER_OVERLAP_BLOCKS", "32"))

SCATTER, SUM_F32, SUM_BF16 = 0, 1, 2
logger = logging.getLogger("peer-collectives")


class PeerTimeout(RuntimeError):
    """A peer did not arrive at an exchange within the time limit; the data of that call is garbage."""


class PeerCollectives:
    def __init__(self, group, device: torch.device, slot_bytes: Optional[int] = None):
        assert device.type == "cuda"
        self.group, self.device = group, device
        self.n, self.rank = dist.get_world_size(group), dist.get_rank(group)
        assert 2 <= self.n <= MAX_RANKS, "peer-memory collectives serve 2..8 ranks of one node"
        self.slot_bytes = (int(slot_bytes or SLOT_BYTES) + 4095) // 4096 * 4096
        assert 2 * self.slot_bytes < (1 << 31), "a mapped region must stay below 2 GiB"
        lib = L.lib()
        lib.areal_peer_region_bytes.restype = ctypes.c_long
        lib.areal_peer_state_offset.restype = ctypes.c_long
        lib.areal_peer_set_timeout.restype = None
        lib.areal_peer_set_timeout(ctypes.c_double(TIMEOUT_S))
        self.nbytes = int(lib.areal_peer_region_bytes(ctypes.c_long(self.slot_bytes)))
        self._state_off = int(lib.areal_peer_state_offset(ctypes.c_long(self.slot_bytes)))
        self._mapped: List[int] = []
        self._ptr = 0
        self.calls = 0  # kernels launched so far
        bases = (ctypes.c_void_p * MAX_RANKS)()
        desc, err = None, ""
        with torch.cuda.device(device):
            try:
                desc = self._allocate(lib)
            except Exception as e:  # this rank still takes part in the exchanges below, so nobody waits for it
                err = f"rank {self.rank}: {e}"
            descs: List = [None] * self.n
            dist.all_gather_object(descs, desc, group=group)
            if all(d is not None for d in descs):
                try:
                    assert all(d["slot"] == self.slot_bytes for d in descs), "ranks disagree on the slot size"
                    for r, d in enumerate(descs):
                        if r == self.rank:
                            bases[r] = self._ptr
                            continue
                        raw = (ctypes.c_ubyte * 64).from_buffer_copy(bytes.fromhex(d["handle"]))
                        base = ctypes.c_void_p(0)
                        L.check(lib.areal_ipc_open(raw, ctypes.byref(base)), "ipc_open")
                        self._mapped.append(base.value)
                        bases[r] = base.value
                except Exception as e:
                    err = f"rank {self.rank}: {e}"
            elif not err:
                err = "a peer could not allocate its region"
            for r in range(self.n, MAX_RANKS):
                bases[r] = self._ptr
            self._bases = bases
        # doubles as the barrier before the first signal: every region is zeroed and mapped once all have reported
        errs: List = [None] * self.n
        dist.all_gather_object(errs, err, group=group)
        if any(errs):
            self.close()
            raise RuntimeError("; ".join(e for e in errs if e))

    def _allocate(self, lib):
        p = ctypes.c_void_p(0)
        # uncached: peers store into this region while this device's kernels read it
        if lib.areal_ipc_alloc_uncached(ctypes.c_long(self.nbytes), ctypes.byref(p)) != 0:
            logger.warning("uncached device memory is unavailable: the staging region is an ordinary allocation")
            p = ctypes.c_void_p(0)
            L.check(lib.areal_ipc_alloc(ctypes.c_long(self.nbytes), ctypes.byref(p)), "ipc_alloc")
        self._ptr = p.value
        # flags and state start at zero (the slots need no initial value)
        tail = self.nbytes - 2 * self.slot_bytes
        zeros = torch.zeros(tail, dtype=torch.uint8, device=self.device)
        L.check(lib.areal_memcpy_d2d_async(ctypes.c_void_p(self._ptr + 2 * self.slot_bytes),
                                           ctypes.c_void_p(zeros.data_ptr()), ctypes.c_long(tail), L.stream_ptr()),
                "memcpy_d2d_async")
        torch.cuda.current_stream().synchronize()
        handle = (ctypes.c_ubyte * 64)()
        off, size = ctypes.c_long(0), ctypes.c_long(0)
        L.check(lib.areal_ipc_export(p, handle, ctypes.byref(off), ctypes.byref(size)), "ipc_export")
        assert off.value == 0
        return dict(handle=bytes(handle).hex(), slot=self.slot_bytes)

    # ------------------------------------------------------------------ the one primitive
    def exchange(self, inp: torch.Tensor, out: torch.Tensor, piece_bytes: int, row_bytes: int, in_piece_stride: int,
                 in_row_stride: int, out_piece_stride: int, out_row_stride: int, mode: int = SCATTER,
                 max_blocks: int = 0, in_offset: int = 0, out_offset: int = 0):
        """One kernel: piece d of `inp` goes to rank d, the piece from rank s lands at piece s of `out` (or all
        received pieces are summed into `out`). Strides and offsets in bytes; see the kernel's header."""
        assert inp.is_cuda and out.is_cuda and self.n * piece_bytes <= self.slot_bytes
        self.calls += 1
        L.check(L.lib().areal_peer_exchange(
            ctypes.c_void_p(inp.data_ptr() + in_offset), ctypes.c_void_p(out.data_ptr() + out_offset),
            ctypes.c_long(piece_bytes), ctypes.c_long(row_bytes), ctypes.c_long(in_piece_stride),
            ctypes.c_long(in_row_stride), ctypes.c_long(out_piece_stride), ctypes.c_long(out_row_stride),
            ctypes.c_int(mode), self._bases, ctypes.c_int(self.n), ctypes.c_int(self.rank),
            ctypes.c_long(self.slot_bytes), ctypes.c_int(max_blocks), L.stream_ptr()), "peer_exchange")

    @staticmethod
    def usable(t: torch.Tensor) -> bool:
        return t.is_cuda and t.is_contiguous() and t.data_ptr() % 16 == 0

    def _rows_per_call(self, row_bytes: int) -> int:
        return max(1, self.slot_bytes // (self.n * row_bytes))

    # ------------------------------------------------------------------ Ulysses
    def seq_to_head(self, x: torch.Tensor) -> torch.Tensor:
        """[t, n, h, D] (this rank's tokens, head groups by destination) -> [n * t, h, D] (everyone's tokens in rank
        order, this rank's head group)."""
        t, n, h, D = x.shape
        assert n == self.n and self.usable(x)
        row = h * D * x.element_size()
        assert row % 16 == 0
        out = x.new_empty((n * t, h, D))
        step = self._rows_per_call(row)
        for r0 in range(0, t, step):
            r = min(step, t - r0)
            self.exchange(x, out, r * row, row, row, n * row, t * row, row, in_offset=r0 * n * row,
                          out_offset=r0 * row)
        return out

    def head_to_seq(self, x: torch.Tensor) -> torch.Tensor:
        """[n * t, h, D] -> [t, n, h, D]: the inverse exchange."""
        T, h, D = x.shape
        n = self.n
        assert T % n == 0 and self.usable(x)
        t = T // n
        row = h * D * x.element_size()
        assert row % 16 == 0
        out = x.new_empty((t, n, h, D))
        step = self._rows_per_call(row)
        for r0 in range(0, t, step):
            r = min(step, t - r0)
            self.exchange(x, out, r * row, row, t * row, row, row, n * row, in_offset=r0 * row,
                          out_offset=r0 * n * row)
        return out

    # ------------------------------------------------------------------ flat buffers
    def _flat_calls(self, piece_bytes: int):
        step = self.slot_bytes // self.n // 16 * 16
        for lo in range(0, piece_bytes, step):
            yield lo, min(step, piece_bytes - lo)

    def all_gather_into(self, full: torch.Tensor, shard: torch.Tensor, max_blocks: int = 0):
        """full = concatenation of every rank's `shard` in rank order (`shard` may be this rank's slice of `full`)."""
        nb = shard.numel() * shard.element_size()
        assert self.usable(full) and self.usable(shard) and full.dtype == shard.dtype
        assert full.numel() == self.n * shard.numel() and nb % 16 == 0
        for lo, ln in self._flat_calls(nb):
            self.exchange(shard, full, ln, ln, 0, ln, nb, ln, SCATTER, max_blocks, in_offset=lo, out_offset=lo)

    def reduce_scatter_into(self, out: torch.Tensor, full: torch.Tensor, max_blocks: int = 0):
        """out = sum over ranks of their `full[rank * len(out) : (rank + 1) * len(out)]` (fp32 or bf16)."""
        nb = out.numel() * out.element_size()
        assert self.usable(full) and self.usable(out) and full.dtype == out.dtype
        assert full.numel() == self.n * out.numel() and nb % 16 == 0
        mode = {torch.float32: SUM_F32, torch.bfloat16: SUM_BF16}[out.dtype]
        for lo, ln in self._flat_calls(nb):
            self.exchange(full, out, ln, ln, nb, ln, 0, ln, mode, max_blocks, in_offset=lo, out_offset=lo)

    # ------------------------------------------------------------------ rows (sequence parallelism)
    def all_gather_rows(self, x: torch.Tensor) -> torch.Tensor:
        """Concatenate `x` from every rank along dim 0."""
        out = x.new_empty((self.n * x.shape[0],) + tuple(x.shape[1:]))
        self.all_gather_into(out, x)
        return out

    def reduce_scatter_rows(self, x: torch.Tensor) -> torch.Tensor:
        """Sum `x` over the ranks and return this rank's 1/n of the rows."""
        assert x.shape[0] % self.n == 0
        out = x.new_empty((x.shape[0] // self.n,) + tuple(x.shape[1:]))
        self.reduce_scatter_into(out, x)
        return out

    def serves_rows(self, x: torch.Tensor) -> bool:
        return (self.usable(x) and x.dtype in (torch.float32, torch.bfloat16)
                and (x.numel() // max(1, self.n)) * x.element_size() % 16 == 0)

    # ------------------------------------------------------------------ health
    def error(self) -> bool:
        """True if a call gave up waiting for a peer. Synchronises the stream."""
        word = torch.empty(1, dtype=torch.int32, device=self.device)
        L.check(L.lib().areal_memcpy_d2d_async(ctypes.c_void_p(word.data_ptr()),
                                               ctypes.c_void_p(self._ptr + self._state_off + 8), ctypes.c_long(4),
                                               L.stream_ptr()), "memcpy_d2d_async")
        return bool(word.item())

    def raise_if_error(self):
        if self.error():
            raise PeerTimeout(f"peer-memory collective: a peer did not arrive within {TIMEOUT_S:.0f} s "
                              "(AREAL_PEER_TIMEOUT_S)")

    def self_check(self) -> bool:
        """Collective: run every kind of exchange on known values (both slots) and agree on the outcome."""
        n, r, dev = self.n, self.rank, self.device
        ok = True
        for rep in range(2):
            x = (torch.arange(6 * n * 2 * 8, device=dev, dtype=torch.float32).view(6, n, 2, 8) + 1000 * r + rep)
            x = x.to(torch.bfloat16)
            y = self.seq_to_head(x)
            want = torch.cat([(torch.arange(6 * n * 2 * 8, device=dev, dtype=torch.float32).view(6, n, 2, 8)
                               + 1000 * s + rep).to(torch.bfloat16)[:, r] for s in range(n)], 0)
            ok = ok and bool((y == want).all().item())
            ok = ok and bool((self.head_to_seq(y) == x).all().item())
            g = torch.full((n * 64,), float(r + 1 + rep), device=dev)
            o = torch.empty(64, device=dev)
            self.reduce_scatter_into(o, g)
            ok = ok and bool((o == n * (n + 1) / 2 + n * rep).all().item())
            full = torch.empty(n * 64, device=dev)
            self.all_gather_into(full, o + r)
            ok = ok and bool((full.view(n, 64)[:, 0] == n * (n + 1) / 2 + n * rep
                              + torch.arange(n, device=dev)).all().item())
        ok = ok and not self.error()
        votes: List = [None] * n
        dist.all_gather_object(votes, ok, group=self.group)
        return all(votes)

    def close(self):
        for b in self._mapped:
            L.lib().areal_ipc_close(ctypes.c_void_p(b))
        self._mapped = []
        if self._ptr:
            L.lib().areal_ipc_free(ctypes.c_void_p(self._ptr))
            self._ptr = 0


def open_for(group, device, slot_bytes: Optional[int] = None) -> Optional[PeerCollectives]:
    """A checked `PeerCollectives` over `group`, or None when the group cannot be served (then the caller keeps the
    communicator). Collective over the group."""
    if not dist.is_initialized() or group is None and dist.get_world_size() < 2:
        return None
    n = dist.get_world_size(group)
    if n < 2 or n > MAX_RANKS or torch.device(device).type != "cuda":
        return None
    try:
        pc = PeerCollectives(group, torch.device(device), slot_bytes)
    except RuntimeError as e:  # raised on every rank alike (see the constructor)
        logger.warning("peer-memory collectives unavailable for this group: %s", e)
        return None
    if not pc.self_check():
        logger.warning("peer-memory collectives failed their self-check: the group stays on the communicator")
        pc.close()
        return None
    return pc
