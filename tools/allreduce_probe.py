"""Latency of the peer-memory all-reduce (ops/ipc_allreduce.py) at decode message sizes.

    python tools/allreduce_probe.py [world=2]

Starts `world` processes on GPU 0 (all that a one-GPU box allows: the peers' regions are then local memory, so the
figure is protocol + kernel time without the xGMI hop, with `world` kernels sharing one device) and times a captured
graph of 64 back-to-back calls of the one-shot and of the two-shot kernel.
"""
import os
import socket
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist

    from areal_vllm_amd.ops.ipc_allreduce import IpcAllReduce

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ar = IpcAllReduce(dist.group.WORLD, dev, 512 * 8192)
    for rows, hidden in [(1, 1536), (64, 1536), (256, 1536), (256, 3584), (256, 5120), (512, 8192)]:
        x = torch.randn(rows, hidden, device=dev).bfloat16()
        took = {}
        for two_shot in (False, True):
            ar.all_reduce_(x, two_shot=two_shot)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(64):
                    ar.all_reduce_(x, two_shot=two_shot)
            x.zero_()
            dist.barrier()
            g.replay()
            torch.cuda.synchronize()
            dist.barrier()
            t = time.perf_counter()
            for _ in range(5):
                g.replay()
            torch.cuda.synchronize()
            took[two_shot] = (time.perf_counter() - t) / (5 * 64)
        if rank == 0:
            print(f"{world} ranks  [{rows:4d}, {hidden:5d}] bf16 ({rows * hidden * 2 / 1024:8.1f} KiB): one-shot "
                  f"{took[False] * 1e6:7.1f} us, two-shot {took[True] * 1e6:7.1f} us per all-reduce", flush=True)
    assert not ar.error()
    dist.barrier()
    ar.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp

    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(worker, args=(world, port), nprocs=world, join=True)
