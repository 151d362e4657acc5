"""Does TunableOp find faster hipBLASLt / rocBLAS solutions than the default heuristic for the decoder's
skinny GEMMs? Every variant is timed inside a captured graph (50 back-to-back calls), so that launch
overhead does not hide the kernel time. Usage: tunable_decode_probe.py <out.csv> [M ...]"""
import sys
import time

import torch
import torch.nn.functional as F

dev = "cuda"
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tunable_decode.csv"
Ms = [int(x) for x in sys.argv[2:]] or [128, 256, 384]
LAYERS = (("qkv", 1536, 2048, True), ("o", 1536, 1536, False), ("gate_up", 1536, 17920, False), ("down", 8960, 1536, False))


def graph_time(fn, n=50, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / (n * reps) * 1e6


def cases(M):
    for name, K, N, bias in LAYERS:
        x = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
        b = torch.randn(N, device=dev).bfloat16() if bias else None
        yield name, 2.0 * M * K * N, (N * K + M * K + M * N) * 2, (lambda x=x, w=w, b=b: F.linear(x, w, b))


base = {}
for M in Ms:
    for name, fl, by, fn in cases(M):
        base[(M, name)] = graph_time(fn)
torch.cuda.tunable.enable(True)
torch.cuda.tunable.tuning_enable(True)
torch.cuda.tunable.set_max_tuning_duration(100)
torch.cuda.tunable.set_max_tuning_iterations(200)
torch.cuda.tunable.set_filename(out)
for M in Ms:
    for name, fl, by, fn in cases(M):
        t0 = time.perf_counter()
        fn()  # tunes this shape
        torch.cuda.synchronize()
        tune_s = time.perf_counter() - t0
        torch.cuda.tunable.tuning_enable(False)
        t = graph_time(fn)
        torch.cuda.tunable.tuning_enable(True)
        b = base[(M, name)]
        print(f"M={M:4d} {name:8s} default {b:7.1f}us ({by / b / 1e6:5.2f} TB/s, {fl / b / 1e6:5.0f} TF)  tuned {t:7.1f}us  "
              f"x{b / t:5.2f}  (tuning {tune_s:4.1f}s)", flush=True)
if hasattr(torch.cuda.tunable, "write_file"):
    torch.cuda.tunable.write_file(out)
