"""Time the grouped GEMM of the routed experts (forward, dx, dW) against a loop of library GEMMs.

    python tools/grouped_gemm_probe.py                 # Mixtral-8x7B and Qwen3-30B-A3B expert shapes, 16k tokens
    python tools/grouped_gemm_probe.py 8192 64 1408 2048 6   # tokens, experts, expert width, hidden, top-k

Rows are routed uniformly at random (multinomial counts), sorted by expert as the MoE layer does it.
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import areal_vllm_amd.ops  # noqa: E402,F401

G = sys.modules["areal_vllm_amd.ops.grouped_gemm"]


def bench(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


def probe(name, T, E, F, H, topk):
    dev = torch.device("cuda")
    torch.manual_seed(0)
    rows = T * topk
    counts = torch.bincount(torch.randint(0, E, (rows,)), minlength=E).tolist()
    for label, N, K in (("gate|up", 2 * F, H), ("down", H, F)):
        x = (0.1 * torch.randn(rows, K, device=dev)).bfloat16().requires_grad_(True)
        w = (0.1 * torch.randn(E, N, K, device=dev)).bfloat16().requires_grad_(True)
        fl = 2.0 * rows * N * K
        with torch.no_grad():
            t_f = bench(lambda: G.grouped_gemm(x, w, counts))
            t_ref = bench(lambda: G.grouped_gemm_ref(x, w, counts))
        y = G.grouped_gemm(x, w, counts)
        g = torch.randn_like(y)

        def fb():
            x.grad = w.grad = None
            G.grouped_gemm(x, w, counts).backward(g)

        t_fb = bench(fb, 5)
        print(f"{name} {label:8s} rows={rows} E={E} N={N} K={K}: fwd {t_f * 1e6:8.0f} us ({fl / t_f / 1e12:6.0f} TF/s)  "
              f"library loop fwd {t_ref * 1e6:8.0f} us ({fl / t_ref / 1e12:6.0f} TF/s)  fwd+bwd {t_fb * 1e6:8.0f} us "
              f"({3 * fl / t_fb / 1e12:6.0f} TF/s)", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        T, E, F, H, topk = (int(a) for a in sys.argv[1:6])
        probe("custom", T, E, F, H, topk)
    else:
        probe("mixtral-8x7b", 16384, 8, 14336, 4096, 2)
        probe("qwen3-30b-a3b", 16384, 128, 768, 2048, 8)
