"""Device memory statistics logging (``areal/utils/device.py`` in the reference)."""
import torch
import torch.distributed as dist

from . import logging

logger = logging.getLogger("device")


def log_gpu_stats(head: str, rank: int = 0):
    if not torch.cuda.is_available():
        return
    if dist.is_initialized() and dist.get_rank() != rank:
        return
    free, total = torch.cuda.mem_get_info()
    gb = 1024**3
    logger.info(f"Memory-Usage {head}: allocated {torch.cuda.memory_allocated() / gb:.2f} GB, "
                f"reserved {torch.cuda.memory_reserved() / gb:.2f} GB, peak {torch.cuda.max_memory_allocated() / gb:.2f} "
                f"GB, device free {free / gb:.2f} / {total / gb:.2f} GB")


def clear_memory():
    import gc

    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
