"""Micro-benchmark: fused skinny GEMM vs the library GEMM at decode shapes (run on the GPU box)."""
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from areal_vllm_amd import ops  # noqa: E402


def timeit(fn, iters=50):
    g = torch.cuda.CUDAGraph()
    fn()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


def main():
    dev = torch.device("cuda")
    torch.manual_seed(0)
    shapes = [("qkv", 2048, 1536), ("o", 1536, 1536), ("gate_up", 17920, 1536), ("down", 1536, 8960),
              ("lm_head", 151936, 1536)]
    ws = ops.SkinnyWorkspace(256, 151936, dev)
    for M in (16, 64, 128, 256):
        for name, N, K in shapes:
            x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
            w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
            ref = F.linear(x, w)
            t_lib = timeit(lambda: F.linear(x, w))
            best = None
            for nt in (1, 2):
                for ks in (1, 2, 4, 8, 16):
                    if K // ks < 128:
                        continue
                    out = ops.skinny_linear(x, w, ksplit=ks, nt=nt, workspace=ws)
                    err = (out.float() - ref.float()).abs().max().item()
                    t = timeit(lambda: ops.skinny_linear(x, w, ksplit=ks, nt=nt, workspace=ws))
                    if best is None or t < best[0]:
                        best = (t, nt, ks, err)
            gbs = N * K * 2 / best[0] / 1e3
            print(f"M={M:4d} {name:8s} N={N:6d} K={K:5d}  lib {t_lib:7.1f}us  skinny {best[0]:7.1f}us (nt={best[1]}, "
                  f"ksplit={best[2]}, {gbs:6.0f} GB/s, maxerr {best[3]:.3g})", flush=True)


if __name__ == "__main__":
    main()
