"""Time the decode-shaped (skinny M) GEMMs through torch (hipBLASLt), optionally with TunableOp."""
import os, sys, time, torch
import torch.nn.functional as F
M = int(sys.argv[1]) if len(sys.argv) > 1 else 128
shapes = [("qkv", 1536, 2048, True), ("o", 1536, 1536, False), ("gate_up", 1536, 17920, False), ("down", 8960, 1536, False), ("lm_head", 1536, 151936, False)]
dev = "cuda"
def bench(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
for name, K, N, bias in shapes:
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    b = torch.zeros(N, device=dev).bfloat16() if bias else None
    wt = w.t().contiguous()
    # cycle several weight copies so the weights come from HBM, not the 256 MB infinity cache
    ws = [w.clone() for _ in range(max(1, int(600e6 // (w.numel() * 2))))]
    i = [0]
    def f():
        i[0] = (i[0] + 1) % len(ws); return F.linear(x, ws[i[0]], b)
    t_nt = bench(f)
    wts = [t.t().contiguous() for t in ws]
    def g():
        i[0] = (i[0] + 1) % len(wts); return torch.mm(x, wts[i[0]])
    t_nn = bench(g)
    gb = w.numel() * 2 / 1e9
    print(f"M={M} {name:8s} K={K} N={N}: linear(NT) {t_nt:7.1f} us ({gb/t_nt*1e6/1e3:5.2f} TB/s)  mm(NN) {t_nn:7.1f} us ({gb/t_nn*1e6/1e3:5.2f} TB/s)", flush=True)
