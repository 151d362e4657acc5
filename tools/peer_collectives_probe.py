"""Time the peer-memory exchange kernels between processes that share ONE GPU.

    python tools/peer_collectives_probe.py --ranks 2

All "peer" stores land in local HBM here, so the figures bound the kernels' own cost (staging copy in, flag round trip,
copy out) from below; they say nothing about xGMI. Sizes: the Ulysses exchange of a 32k-token micro-batch of the 1.5B
and 7B models, a 772 MB fp32 gradient bucket, and its bf16 parameter all-gather."""
import argparse
import os
import socket
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port, blocks):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist

    from areal_vllm_amd.ops.peer_collectives import open_for

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pc = open_for(dist.group.WORLD, dev)
    assert pc is not None

    def timed(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        dist.barrier()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps

    rows = []
    for name, T, heads in [("ulysses 1.5B (12 q + 2x2 kv heads)", 32768, 12 + 2 * max(2, world)),
                           ("ulysses 7B (28 q + 2x4 kv heads)", 32768, 28 + 2 * max(4, world))]:
        t, h = T // world, heads // world
        x = torch.randn(t, world, h, 128, device=dev).bfloat16()
        dt = timed(lambda: pc.seq_to_head(x))
        y = pc.seq_to_head(x)
        dt2 = timed(lambda: pc.head_to_seq(y))
        nb = x.numel() * 2
        rows.append(f"{name}: {nb / 2**20:.0f} MiB per rank  seq->head {dt * 1e6:7.0f} us ({nb / dt / 1e9:6.0f} GB/s)  "
                    f"head->seq {dt2 * 1e6:7.0f} us ({nb / dt2 / 1e9:6.0f} GB/s)")
    n = 193_000_000 // world * world
    g = torch.randn(n, device=dev)
    out = torch.empty(n // world, device=dev)
    for mb in (0, blocks):
        dt = timed(lambda: pc.reduce_scatter_into(out, g, max_blocks=mb), reps=5)
        rows.append(f"reduce-scatter fp32 {n * 4 / 2**20:.0f} MiB, max_blocks={mb or 128}: {dt * 1e3:6.2f} ms "
                    f"({n * 4 / dt / 1e9:6.0f} GB/s of input)")
    p = torch.randn(n, device=dev).bfloat16()
    dt = timed(lambda: pc.all_gather_into(p, p[rank * (n // world):(rank + 1) * (n // world)]), reps=5)
    rows.append(f"all-gather bf16 {n * 2 / 2**20:.0f} MiB: {dt * 1e3:6.2f} ms ({n * 2 / dt / 1e9:6.0f} GB/s of output)")
    assert not pc.error()
    if rank == 0:
        print(f"{world} processes on one GPU (local memory, not xGMI)")
        print("\n".join(rows), flush=True)
    dist.barrier()
    pc.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--blocks", type=int, default=32)
    a = ap.parse_args()
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=worker, args=(r, a.ranks, port, a.blocks)) for r in range(a.ranks)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    bad = [p for p in procs if p.is_alive() or p.exitcode != 0]
    for p in procs:
        if p.is_alive():
            p.kill()
    sys.exit(1 if bad else 0)
