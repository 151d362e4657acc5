"""Platform abstraction: MI355X/ROCm when a GPU is visible, CPU otherwise.

The reference only detects NVIDIA devices (``areal/platforms/__init__.py:13-33``); this framework targets
gfx950 and keeps a CPU/gloo platform for unit tests and the plumbing configuration.
"""
import os

import torch

from .platform import CpuPlatform, Platform, RocmPlatform


def _detect() -> Platform:
    if os.environ.get("AREAL_PLATFORM", "").lower() == "cpu":  # forced host run (tests, plumbing configuration)
        return CpuPlatform()
    try:
        if torch.cuda.is_available() and torch.cuda.device_count() > 0:
            return RocmPlatform()
    except Exception:
        pass
    return CpuPlatform()


class _Lazy:
    """Resolve the platform on first use so that importing this package never initialises the GPU."""

    _p = None

    def _get(self) -> Platform:
        if _Lazy._p is None:
            _Lazy._p = _detect()
        return _Lazy._p

    def __getattr__(self, name):
        return getattr(self._get(), name)


current_platform = _Lazy()
