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
from multiprocessing import shared_memory
from typing import Any, Dict, Optional

import torch

from . import _lib as L


class ExportedBuffer:
    """Exporter side. Keep it alive as long as peers may read."""

    def __init__(self, t: torch.Tensor):
        assert t.is_contiguous()
        self.tensor = t
        self.nbytes = t.numel() * t.element_size()
        self._shm: Optional[shared_memory.SharedMemory] = None
        if t.is_cuda:
            L.lib()
            handle = (ctypes.c_ubyte * 64)()
            off, size = ctypes.c_long(0), ctypes.c_long(0)
            L.check(L.lib().areal_ipc_export(L.ptr(t), handle, ctypes.byref(off), ctypes.byref(size)), "ipc_export")
            self.desc: Dict[str, Any] = dict(kind="hip", handle=bytes(handle).hex(), offset=int(off.value),
                                             nbytes=self.nbytes, device=int(t.device.index or 0))
        else:
            self._shm = shared_memory.SharedMemory(create=True, size=max(1, self.nbytes))
            self.desc = dict(kind="shm", name=self._shm.name, offset=0, nbytes=self.nbytes, device=-1)

    def refresh(self):
        """Make the current contents visible to peers: device buffers are live (the caller synchronises its
        stream), the host mirror is rewritten."""
        if self._shm is not None:
            src = self.tensor.detach().contiguous().view(torch.uint8).numpy()
            self._shm.buf[: self.nbytes] = src.tobytes()
        elif self.tensor.is_cuda:
            torch.cuda.current_stream(self.tensor.device).synchronize()

    def close(self):
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
    """Importer side: a mapping of another process's exported buffer."""

    def __init__(self, desc: Dict[str, Any]):
        self.desc = dict(desc)
        self.nbytes = int(desc["nbytes"])
        self._base = None
        self._shm = None
        if desc["kind"] == "hip":
            raw = bytes.fromhex(desc["handle"])
            assert len(raw) == 64
            handle = (ctypes.c_ubyte * 64).from_buffer_copy(raw)
            base = ctypes.c_void_p(0)
            L.check(L.lib().areal_ipc_open(handle, ctypes.byref(base)), "ipc_open")
            self._base = base.value
            self.ptr = self._base + int(desc["offset"])
        else:
            self._shm = shared_memory.SharedMemory(name=desc["name"])
            self.ptr = None

    def copy_into(self, dst: torch.Tensor, nbytes: Optional[int] = None):
        """Asynchronous copy of the first `nbytes` bytes into `dst` on the current stream (device buffers);
        a plain memcpy for the shared-memory backing."""
        n = self.nbytes if nbytes is None else int(nbytes)
        assert n <= self.nbytes and n <= dst.numel() * dst.element_size() and dst.is_contiguous()
        if self._shm is not None:
            src = torch.frombuffer(self._shm.buf, dtype=torch.uint8, count=n)
            dst.view(torch.uint8).view(-1)[:n].copy_(src)
            return
        assert dst.is_cuda
        L.check(L.lib().areal_memcpy_d2d_async(L.ptr(dst), ctypes.c_void_p(self.ptr), ctypes.c_long(n), L.stream_ptr()),
                "memcpy_d2d_async")

    def close(self):
        if self._base is not None:
            L.lib().areal_ipc_close(ctypes.c_void_p(self._base))
            self._base = None
        if self._shm is not None:
            self._shm.close()
            self._shm = None
