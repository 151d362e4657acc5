"""Time the hand-written weight-gradient GEMM (fp32 accumulation into the gradient buffer) against the library call the
trainer used before (`torch.addmm(acc, dy.T, x, out_dtype=float32, out=acc)`), on trainer shapes, random operands,
variants interleaved in one process (median of rounds)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from areal_vllm_amd.ops import train_gemm as tg  # noqa: E402
from tgemm_probe import SHAPES, time_rounds  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=32768)
    ap.add_argument("--model", default="qwen2-1.5b")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--splits", default="0")
    args = ap.parse_args()
    s = SHAPES[args.model]
    dev = torch.device("cuda")
    M = args.rows
    for name, N, K in [("qkv", s["QKV"], s["H"]), ("o", s["H"], s["H"]), ("gate_up", 2 * s["F"], s["H"]),
                       ("down", s["H"], s["F"])]:
        torch.manual_seed(0)
        dy = ((torch.rand(M, N, device=dev) * 2 - 1) / M ** 0.5).bfloat16()
        x = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
        base = torch.randn(N, K, device=dev)
        ref = base[:256].double() + dy[:, :256].double().t() @ x.double()
        line = f"M={M:6d} {name:8s} dW[{N:6d},{K:6d}] |"
        fns = {}
        for sp in [int(v) for v in args.splits.split(",")]:
            got = tg.tgemm_wgrad_(base.clone(), dy, x, splits=sp)
            err = float((got[:256].double() - ref).abs().max())
            line += f" splits={sp} err {err:.2e}"
            acc = base.clone()
            fns[f"own/{sp}"] = (lambda acc=acc, sp=sp: tg.tgemm_wgrad_(acc, dy, x, splits=sp))
        lib = base.clone()
        torch.addmm(lib, dy.t(), x, out_dtype=torch.float32, out=lib)
        line += f" lib err {float((lib[:256].double() - ref).abs().max()):.2e} |"
        acc_l = base.clone()
        fns["lib"] = lambda: torch.addmm(acc_l, dy.t(), x, out_dtype=torch.float32, out=acc_l)
        res = time_rounds(fns, args.iters, args.rounds)
        fl = 2.0 * M * N * K
        for k, (med, mn) in res.items():
            line += f" {k} {med:8.1f} us ({fl / med / 1e6:6.0f} TF) |"
        print(line, flush=True)


if __name__ == "__main__":
    main()
