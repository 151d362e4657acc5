"""Fake-cluster harness: run a function on N local processes that form a gloo world on the CPU platform,
with an NFS-style name_resolve root under a temporary directory - the way every distributed component
here is tested without GPUs. Role of ``realhf/base/testing.py`` (``StandaloneTestingProcess`` /
``LocalMultiProcessTest``).

    def body(rank, world, shared):          # runs in every process; torch.distributed is initialised
        ...
        return rank * 2                     # picklable result, collected by the parent

    results = LocalMultiProcessTest(4, body, shared=dict(x=1)).launch()
"""
import os
import socket
import tempfile
import traceback
from typing import Any, Callable, Dict, List, Optional

import torch.multiprocessing as mp

TESTING_MODEL_VOCAB_SIZE = 512
TESTING_MODEL_PRESET = "tiny"


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _entry(rank: int, world: int, port: int, nr_root: str, fn: Callable, shared: Any, q, init_dist: bool):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo", AREAL_PLATFORM="cpu",
                      HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    try:
        import torch
        import torch.distributed as dist

        from ..utils import name_resolve

        torch.set_num_threads(1)
        name_resolve.reconfigure("nfs", nfs_record_root=nr_root)
        if init_dist:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        res = fn(rank, world, shared)
        if init_dist:
            dist.barrier()
            dist.destroy_process_group()
        q.put((rank, True, res))
    except Exception:
        q.put((rank, False, traceback.format_exc()))


class LocalMultiProcessTest:
    def __init__(self, world_size: int, fn: Callable[[int, int, Any], Any], shared: Any = None,
                 init_distributed: bool = True, timeout: float = 600.0, name_resolve_root: Optional[str] = None):
        self.world_size, self.fn, self.shared = world_size, fn, shared
        self.init_distributed, self.timeout = init_distributed, timeout
        self.nr_root = name_resolve_root or tempfile.mkdtemp(prefix="areal_nr_")

    def launch(self) -> List[Any]:
        """Run `fn` on every rank; returns the results ordered by rank or raises with the first traceback."""
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_entry, args=(r, self.world_size, port, self.nr_root, self.fn, self.shared, q,
                                                  self.init_distributed)) for r in range(self.world_size)]
        for p in procs:
            p.start()
        out: Dict[int, Any] = {}
        try:
            for _ in range(self.world_size):
                rank, ok, res = q.get(timeout=self.timeout)
                if not ok:
                    raise RuntimeError(f"rank {rank} failed:\n{res}")
                out[rank] = res
        finally:
            for p in procs:
                p.join(10)
                if p.is_alive():
                    p.terminate()
        return [out[r] for r in range(self.world_size)]
