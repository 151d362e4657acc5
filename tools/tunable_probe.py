"""Compare hipBLASLt's default algorithm choice with TunableOp's pick for the trainer's and the decoder's
GEMM shapes. Usage: tunable_probe.py <T> <out.csv>; prints default vs tuned time per GEMM."""
import sys, time, torch
import torch.nn.functional as F

dev = "cuda"
T = int(sys.argv[1]) if len(sys.argv) > 1 else 29440
out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/tunable_probe.csv"
FWD_ONLY = len(sys.argv) > 3 and sys.argv[3] == "fwd"
LAYERS = (("qkv", 1536, 2048), ("o", 1536, 1536), ("gate_up", 1536, 17920), ("down", 8960, 1536))


def bench(fn, n=50 if FWD_ONLY else 10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


def cases():
    for name, K, N in LAYERS:
        x = torch.randn(T, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
        dy = torch.randn(T, N, device=dev).bfloat16()
        acc = torch.zeros(N, K, device=dev, dtype=torch.float32)
        fl = 2.0 * T * K * N
        yield f"{name}.fwd", fl, (lambda x=x, w=w: F.linear(x, w))
        if FWD_ONLY:
            continue
        yield f"{name}.dgrad", fl, (lambda dy=dy, w=w: torch.mm(dy, w))
        yield f"{name}.wgrad", fl, (lambda dy=dy, x=x, acc=acc: torch.addmm(acc, dy.t(), x, out_dtype=torch.float32, out=acc))


base = {}
for name, fl, fn in cases():
    base[name] = bench(fn)
torch.cuda.tunable.enable(True)
torch.cuda.tunable.tuning_enable(True)
torch.cuda.tunable.set_max_tuning_duration(40)
torch.cuda.tunable.set_max_tuning_iterations(50 if FWD_ONLY else 20)
torch.cuda.tunable.set_filename(out)
for name, fl, fn in cases():
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    tune_s = time.perf_counter() - t0
    t = bench(fn)
    print(f"T={T} {name:14s} default {base[name]:8.1f}us ({fl / base[name] / 1e6:6.0f}TF)  tuned {t:8.1f}us "
          f"({fl / t / 1e6:6.0f}TF)  x{base[name] / t:5.2f}  (tuning took {tune_s:5.1f}s)", flush=True)
if hasattr(torch.cuda.tunable, "write_file"):
    torch.cuda.tunable.write_file(out)
