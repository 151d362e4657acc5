"""Experiment: forward flash-attention rate over sequence-length mixes at a fixed token count, to separate
the per-iteration cost of the main loop from per-workgroup fixed costs and load imbalance."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from areal_vllm_amd.ops import attention as A  # noqa: E402

dev = torch.device("cuda")
Hq, Hkv, D = 12, 2, 128
for name, lens in (("25 x ~1150", torch.randint(1050, 1216, (25,), generator=torch.Generator().manual_seed(0)).tolist()),
                   ("224 x 128", [128] * 224), ("56 x 512", [512] * 56), ("14 x 2048", [2048] * 14),
                   ("4 x 8192", [8192] * 4), ("1 x 32768", [32768])):
    T = sum(lens)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev)
    qkv = torch.randn(T, Hq + 2 * Hkv, D, device=dev, dtype=torch.bfloat16)
    q, k, v = qkv[:, :Hq], qkv[:, Hq:Hq + Hkv], qkv[:, Hq + Hkv:]
    f = lambda: A._flash_fwd_hip(q, k, v, cu, cu, max(lens), D ** -0.5, True)  # noqa: E731
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t) / 20 * 1e6
    fl = sum(4 * l * l / 2 * Hq * D for l in lens)
    tiles = sum(sum(2 * (i + 1) for i in range((l + 127) // 128)) for l in lens) * Hq  # workgroup iterations
    print(f"{name:12s} T={T:6d}: {us:8.1f} us  {fl / us / 1e6:6.0f} TF/s (useful)  {tiles} wg-iterations -> "
          f"{us * 512 / tiles * 1000:6.0f} ns per iteration at 512 resident workgroups", flush=True)
