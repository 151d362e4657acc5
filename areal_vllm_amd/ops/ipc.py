"""Peer-memory buffers for weight shipping between processes (``csrc/hip/ipc.hip``).

``export_buffer(t)`` describes the memory behind a tensor so that ANOTHER process can map it;
``PeerBuffer(desc)`` maps it there and ``copy_into`` pulls bytes with an asynchronous device-to-device copy
on the caller's stream. Between two GPUs of a node the copy rides xGMI; two processes on one GPU share the
same mechanism (the handle resolves to local memory).

On the CPU platform (plumbing tests without a GPU) the same interface is backed by POSIX shared memory: the
exporter owns a mirror segment and ``refresh()`` copies the tensor into it before announcing an update.

Role in the reference: the NCCL broadcast funnel of ``areal/engine/fsdp_engine.py:399-433``.
"""
import ctypes
import os
from multiprocessing import shared_memory
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib as L


CHUNK_BYTES = 1 << 30  # exported pieces stay well below 2 GiB: mapping a larger allocation hangs (see ipc.hip)


_LOCAL_EXPORTS: Dict[str, "ExportedBuffer"] = {}  # device snapshots exported by this process, by descriptor key


class ExportedBuffer:
    """Exporter side: a SNAPSHOT area other processes can map, refreshed from the tensor on demand.

    Device tensors: the snapshot is a list of dedicated device allocations of at most `CHUNK_BYTES` each
    (allocated outside the framework's caching allocator so that a handle never names a multi-GiB segment);
    ``refresh()`` copies the tensor into them on the current stream (an on-device copy, ~1.4 TB/s) and waits.
    Peers then read a consistent image while the owner goes on mutating the tensor. Host tensors: the
    snapshot is one POSIX shared-memory segment. Keep the object alive as long as peers may read."""

    def __init__(self, t: torch.Tensor):
        assert t.is_contiguous()
        self.tensor = t
        self.nbytes = t.numel() * t.element_size()
        self._shm: Optional[shared_memory.SharedMemory] = None
        self._chunks = []  # (device pointer, nbytes)
        if t.is_cuda:
            lib = L.lib()
            pieces = []
            for lo in range(0, self.nbytes, CHUNK_BYTES):
                n = min(CHUNK_BYTES, self.nbytes - lo)
                p = ctypes.c_void_p(0)
                L.check(lib.areal_ipc_alloc(ctypes.c_long(n), ctypes.byref(p)), "ipc_alloc")
                handle = (ctypes.c_ubyte * 64)()
                off, size = ctypes.c_long(0), ctypes.c_long(0)
                L.check(lib.areal_ipc_export(p, handle, ctypes.byref(off), ctypes.byref(size)), "ipc_export")
                assert off.value == 0 and size.value < (1 << 31)
                self._chunks.append((p.value, n))
                pieces.append(dict(handle=bytes(handle).hex(), nbytes=n))
            self.desc: Dict[str, Any] = dict(kind="hip", chunks=pieces, nbytes=self.nbytes, pid=os.getpid(),
                                             device=int(t.device.index or 0), key=pieces[0]["handle"] if pieces else "")
            _LOCAL_EXPORTS[self.desc["key"]] = self
        else:
            self._shm = shared_memory.SharedMemory(create=True, size=max(1, self.nbytes))
            self.desc = dict(kind="shm", name=self._shm.name, nbytes=self.nbytes, device=-1, key=self._shm.name)

    def refresh(self):
        """Take a new snapshot of the tensor (returns when it is complete)."""
        if self._shm is not None:
            src = self.tensor.detach().contiguous().view(torch.uint8).numpy()
            self._shm.buf[: self.nbytes] = src.tobytes()
            return
        base = self.tensor.data_ptr()
        lo = 0
        with torch.cuda.device(self.tensor.device):
            for p, n in self._chunks:
                L.check(L.lib().areal_memcpy_d2d_async(ctypes.c_void_p(p), ctypes.c_void_p(base + lo), ctypes.c_long(n),
                                                       L.stream_ptr()), "memcpy_d2d_async")
                lo += n
            torch.cuda.current_stream().synchronize()

    def close(self):
        _LOCAL_EXPORTS.pop(self.desc.get("key"), None)
        for p, _ in self._chunks:
            L.lib().areal_ipc_free(ctypes.c_void_p(p))
        self._chunks = []
        if self._shm is not None:
            try:
                self._shm.close()
                self._shm.unlink()
            except FileNotFoundError:
                pass
            self._shm = None


def export_buffer(t: torch.Tensor) -> ExportedBuffer:
    return ExportedBuffer(t)


class PeerBuffer:
    """Importer side: a mapping of another process's exported snapshot."""

    def __init__(self, desc: Dict[str, Any]):
        self.desc = dict(desc)
        self.nbytes = int(desc["nbytes"])
        self._maps = []  # (mapped base pointer, nbytes)
        self._own = False  # the pointers belong to an export of this process: nothing to unmap
        self._shm = None
        local = _LOCAL_EXPORTS.get(desc.get("key")) if desc.get("pid") == os.getpid() else None
        if desc["kind"] == "hip" and local is not None:
            # exported by THIS process (a server that lives next to one of the trainer's ranks): the runtime refuses to
            # open a handle in the process that made it, and there is nothing to map - use the snapshot's own pointers
            self._maps = [(p, n) for p, n in local._chunks]
            self._own = True
        elif desc["kind"] == "hip":
            for piece in desc["chunks"]:
                raw = bytes.fromhex(piece["handle"])
                assert len(raw) == 64
                handle = (ctypes.c_ubyte * 64).from_buffer_copy(raw)
                base = ctypes.c_void_p(0)
                L.check(L.lib().areal_ipc_open(handle, ctypes.byref(base)), "ipc_open")
                self._maps.append((base.value, int(piece["nbytes"])))
        else:
            self._shm = shared_memory.SharedMemory(name=desc["name"])

    def copy_into(self, dst: torch.Tensor, nbytes: Optional[int] = None):
        """Asynchronous copy of the first `nbytes` bytes into `dst` on the current stream (device snapshots);
        a plain memcpy for the shared-memory backing."""
        n = self.nbytes if nbytes is None else int(nbytes)
        assert n <= self.nbytes and n <= dst.numel() * dst.element_size() and dst.is_contiguous()
        if self._shm is not None:
            src = torch.frombuffer(self._shm.buf, dtype=torch.uint8, count=n)
            dst.view(torch.uint8).view(-1)[:n].copy_(src)
            return
        assert dst.is_cuda
        lo = 0
        for base, size in self._maps:
            take = min(size, n - lo)
            if take <= 0:
                break
            L.check(L.lib().areal_memcpy_d2d_async(ctypes.c_void_p(dst.data_ptr() + lo), ctypes.c_void_p(base),
                                                   ctypes.c_long(take), L.stream_ptr()), "memcpy_d2d_async")
            lo += take

    def prepare_moves(self, triples: np.ndarray, elt_bytes: int, device) -> List[Tuple[int, Any, int, int]]:
        """Turn ``[n, 3]`` (source offset, destination offset, length) ELEMENT ranges of the exporter's tensor into
        what `move_intervals` launches: per mapped piece of the snapshot the byte ranges that fall into it (ranges
        that straddle two pieces are cut), as a device array. Weight-pull plans are the same for every update, so
        callers prepare once and replay."""
        t = np.ascontiguousarray(triples, dtype=np.int64).reshape(-1, 3) * int(elt_bytes)
        t = t[t[:, 2] > 0]
        if self._shm is not None:
            return [(-1, t, int(t.shape[0]), int(t[:, 2].max()) if len(t) else 0)]
        out, lo = [], 0
        for ci, (_base, size) in enumerate(self._maps):
            hi = lo + size
            s0, s1 = np.maximum(t[:, 0], lo), np.minimum(t[:, 0] + t[:, 2], hi)
            keep = s1 > s0
            if keep.any():
                cut = np.stack([s0[keep] - lo, t[keep, 1] + (s0[keep] - t[keep, 0]), s1[keep] - s0[keep]], 1)
                dev = torch.from_numpy(np.ascontiguousarray(cut)).to(device)
                out.append((ci, dev, int(cut.shape[0]), int(cut[:, 2].max())))
            lo = hi
        return out

    def move_intervals(self, dst: torch.Tensor, prepared: List[Tuple[int, Any, int, int]]):
        """Copy the prepared ranges into `dst` (a contiguous local buffer) on the current stream: one kernel launch
        per mapped piece of the snapshot (``csrc/hip/ipc.hip::interval_move_kernel``); plain memory copies for the
        shared-memory backing."""
        assert dst.is_contiguous()
        if self._shm is not None:
            src = torch.frombuffer(self._shm.buf, dtype=torch.uint8, count=self.nbytes)
            flat = dst.view(torch.uint8).view(-1)
            for _ci, t, _n, _m in prepared:
                for so, do, ln in t.tolist():
                    flat[do:do + ln].copy_(src[so:so + ln])
            return
        assert dst.is_cuda
        for ci, dev, n, max_len in prepared:
            L.check(L.lib().areal_interval_move(ctypes.c_void_p(self._maps[ci][0]), ctypes.c_void_p(dst.data_ptr()),
                                                ctypes.c_void_p(dev.data_ptr()), ctypes.c_int(n), ctypes.c_long(max_len),
                                                L.stream_ptr()), "interval_move")

    def close(self):
        for base, _ in ([] if self._own else self._maps):
            L.lib().areal_ipc_close(ctypes.c_void_p(base))
        self._maps = []
        if self._shm is not None:
            self._shm.close()
            self._shm = None
