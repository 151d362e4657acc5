"""Time the hand-written training GEMM against the library GEMM (hipBLASLt through torch) on trainer shapes.

    python tools/tgemm_probe.py [--rows 32768] [--model qwen2-1.5b|qwen2.5-7b]

Operands are uniform random in [-1, 1) scaled by 1/sqrt(K) on the weight side (random data, not zeros: zero-filled
operands let the chip clock higher and overstate every kernel). Variants are timed interleaved in one process,
`--rounds` rounds of `--iters` back-to-back calls each; the median round is printed. The first column checks the result
against an fp32 reference on a slice of the output."""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from areal_vllm_amd.ops import train_gemm as tg  # noqa: E402

SHAPES = {
    "qwen2-1.5b": dict(H=1536, F=8960, QKV=2048, V=151936),
    "qwen2.5-7b": dict(H=3584, F=18944, QKV=4608, V=152064),
}


def time_rounds(fns, iters, rounds):
    out = {k: [] for k in fns}
    for _ in range(rounds):
        for k, fn in fns.items():
            fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                fn()
            e.record()
            torch.cuda.synchronize()
            out[k].append(s.elapsed_time(e) / iters * 1e3)
    return {k: (statistics.median(v), min(v)) for k, v in out.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=32768)
    ap.add_argument("--model", default="qwen2-1.5b")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--lm-head", action="store_true")
    args = ap.parse_args()
    s = SHAPES[args.model]
    dev = torch.device("cuda")
    M = args.rows
    cases = [("qkv", s["QKV"], s["H"]), ("o", s["H"], s["H"]), ("gate_up", 2 * s["F"], s["H"]),
             ("down", s["H"], s["F"]), ("dgrad gate_up", s["H"], 2 * s["F"]), ("dgrad down", s["F"], s["H"])]
    if args.lm_head:
        cases.append(("lm_head (8192 rows)", s["V"], s["H"]))
    # gate|up + SwiGLU in one kernel against library GEMM + activation kernel
    import areal_vllm_amd.ops as ops
    import torch.nn.functional as Fn

    torch.manual_seed(0)
    a = (torch.rand(M, s["H"], device=dev) * 2 - 1).bfloat16()
    w = ((torch.rand(2 * s["F"], s["H"], device=dev) * 2 - 1) / s["H"] ** 0.5).bfloat16()
    out = torch.empty(M, s["F"], device=dev, dtype=torch.bfloat16)
    res = time_rounds({"lib+act": lambda: ops.swiglu(Fn.linear(a, w), "silu"),
                       "fused": lambda: tg.tgemm_swiglu(a, w, out=out)}, args.iters, args.rounds)
    print(f"M={M:6d} gate|up + SwiGLU  F={s['F']:6d} K={s['H']:6d} | "
          + " | ".join(f"{k} {med:8.1f} us (best {mn:8.1f})" for k, (med, mn) in res.items()), flush=True)
    for name, N, K in cases:
        m = 8192 if name.startswith("lm_head") else M
        torch.manual_seed(0)
        a = (torch.rand(m, K, device=dev) * 2 - 1).bfloat16()
        b = ((torch.rand(N, K, device=dev) * 2 - 1) / K ** 0.5).bfloat16()
        got = tg.tgemm_tn(a, b)
        got2 = tg.tgemm_tn(a, b, max_blocks=0)
        rows = torch.cat([torch.arange(0, 300, device=dev), torch.arange(m - 300, m, device=dev)])
        ref = a[rows].float() @ b.float().t()
        err = float((got[rows].float() - ref).abs().max())
        err2 = float((got2[rows].float() - ref).abs().max())
        lib_err = float(((a[rows] @ b.t()).float() - ref).abs().max())
        same = bool(torch.equal(got, got2))
        out = torch.empty(m, N, device=dev, dtype=torch.bfloat16)
        res = time_rounds({"lib": lambda: torch.mm(a, b.t(), out=out), "own": lambda: tg.tgemm_tn(a, b, out=out),
                           "own-1tile": lambda: tg.tgemm_tn(a, b, out=out, max_blocks=0)}, args.iters, args.rounds)
        fl = 2.0 * m * N * K
        line = f"M={m:6d} {name:18s} N={N:6d} K={K:6d}  err own {err:.4f}/{err2:.4f} lib {lib_err:.4f} same={same} |"
        for k, (med, mn) in res.items():
            line += f" {k} {med:8.1f} us ({fl / med / 1e6:6.0f} TF, best {fl / mn / 1e6:6.0f}) |"
        print(line, flush=True)


if __name__ == "__main__":
    main()
