"""ctypes loader for the in-tree HIP kernel library (``_C/libareal_hip.so``).

Kernels are exported with a C ABI (raw device pointers + ``hipStream_t``); the wrappers in this
package pass ``tensor.data_ptr()`` and the current torch stream, so kernel launches are captured
by hipGraph capture like any other stream work.

On a machine with a GPU the library MUST load: ops raise instead of silently falling back to
PyTorch. On CPU-only machines (unit tests, the gloo plumbing config) the wrappers use their
PyTorch reference implementations.
"""
import ctypes
import os
from pathlib import Path

import torch

_LIB = None
_LIB_ERR = None
_LIB_PATH = Path(__file__).resolve().parent.parent / "_C" / "libareal_hip.so"

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_int64 = ctypes.c_int64
c_uint64 = ctypes.c_uint64
c_float = ctypes.c_float


def _try_load():
    global _LIB, _LIB_ERR
    if _LIB is not None or _LIB_ERR is not None:
        return
    try:
        _LIB = ctypes.CDLL(str(_LIB_PATH))
    except OSError as e:  # missing file or missing ROCm runtime
        _LIB_ERR = e


def lib():
    """Return the loaded kernel library or raise with build instructions."""
    _try_load()
    if _LIB is None:
        raise RuntimeError(
            f"HIP kernel library not available ({_LIB_ERR}). Build it with `python build_ext.py` "
            f"(expected at {_LIB_PATH})."
        )
    return _LIB


def available() -> bool:
    _try_load()
    return _LIB is not None


def use_hip(t: torch.Tensor) -> bool:
    """True when `t` lives on the GPU, in which case the HIP kernel is mandatory."""
    if not t.is_cuda:
        return False
    lib()  # raises loudly if the extension is missing on a GPU box
    return True


def stream_ptr():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return c_void_p(0)
    return c_void_p(t.data_ptr())


def check(code: int, name: str):
    if code != 0:
        raise RuntimeError(f"HIP kernel launcher {name} failed with code {code}")


def dtype_flag(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return 0
    if t.dtype == torch.float32:
        return 1
    raise TypeError(f"unsupported dtype {t.dtype}")
