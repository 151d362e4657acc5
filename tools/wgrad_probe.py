"""Time weight-gradient GEMM formulations for the trainer's shapes (T tokens per micro-batch)."""
import sys, time, torch
dev = "cuda"
T = int(sys.argv[1]) if len(sys.argv) > 1 else 29440
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
for name, K, N in (("qkv", 1536, 2048), ("o", 1536, 1536), ("gate_up", 1536, 17920), ("down", 8960, 1536)):
    x = torch.randn(T, K, device=dev).bfloat16(); dy = torch.randn(T, N, device=dev).bfloat16()
    acc = torch.zeros(N, K, device=dev, dtype=torch.float32)
    fl = 2.0 * T * K * N
    res = {}
    res["dyT@x"] = bench(lambda: torch.mm(dy.t(), x))                       # autograd's form, [N, K]
    res["(xT@dy)T"] = bench(lambda: torch.mm(x.t(), dy))                    # [K, N] (transposed result)
    xt = x.t().contiguous(); dyt = dy.t().contiguous()
    res["dyTc@x"] = bench(lambda: torch.mm(dyt, x))                         # pre-transposed dy (cost of transpose excluded)
    res["dyTc@xTcT"] = bench(lambda: torch.mm(dyt, xt.t()))                 # both K-contiguous (NT form)
    res["transpose dy"] = bench(lambda: dy.t().contiguous())
    res["transpose x"] = bench(lambda: x.t().contiguous())
    try:
        res["f32out"] = bench(lambda: torch.mm(dy.t(), x, out_dtype=torch.float32))
    except Exception as e:
        res["f32out"] = float("nan")
    try:
        res["addmm_f32"] = bench(lambda: torch.addmm(acc, dy.t(), x, out_dtype=torch.float32, out=acc))
    except Exception as e:
        res["addmm_f32"] = float("nan")
    res["add_"] = bench(lambda: acc.add_(torch.empty(N, K, device=dev, dtype=torch.bfloat16)))
    print(f"T={T} {name:8s} " + " | ".join(f"{k} {v:7.1f}us ({fl / v / 1e6:6.0f}TF)" if "@" in k or "out" in k or "addmm" in k else f"{k} {v:7.1f}us" for k, v in res.items()), flush=True)
