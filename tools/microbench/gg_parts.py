"""Split the backward of the grouped GEMM into its parts (transposed weight copy, dx kernel, dW kernel)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import areal_vllm_amd.ops  # noqa
G = sys.modules["areal_vllm_amd.ops.grouped_gemm"]
L = G.L
dev = torch.device("cuda")
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
for (rows, E, N, K) in ((131072, 128, 1536, 2048), (131072, 128, 2048, 768)):
    counts = torch.bincount(torch.randint(0, E, (rows,)), minlength=E).tolist()
    offsets = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32, device=dev)
    x = (0.1 * torch.randn(rows, K, device=dev)).bfloat16()
    w = (0.1 * torch.randn(E, N, K, device=dev)).bfloat16()
    dy = (0.1 * torch.randn(rows, N, device=dev)).bfloat16()
    wT = w.transpose(1, 2).contiguous()
    dw = torch.empty_like(w)
    t_tr = bench(lambda: w.transpose(1, 2).contiguous())
    t_dx = bench(lambda: G._launch(dy, wT, offsets))
    def wg():
        L.check(L.lib().areal_grouped_gemm_wgrad(L.ptr(dy), L.ptr(x), L.ptr(dw), L.ptr(offsets), L.c_int(rows), L.c_int(E),
                                                 L.c_int(N), L.c_int(K), L.c_int64(dw.stride(0)), L.c_int(0), L.stream_ptr()), "wgrad")
    t_wg = bench(wg)
    fl = 2.0 * rows * N * K
    print(f"rows={rows} E={E} N={N} K={K}: transpose copy {t_tr:.0f} us, dx kernel {t_dx:.0f} us ({fl/t_dx/1e6:.0f} TF/s), "
          f"dW kernel {t_wg:.0f} us ({fl/t_wg/1e6:.0f} TF/s)", flush=True)
