"""Balanced gate|up GEMM with SwiGLU epilogue (csrc/hip/decode_swiglu.hip) against the tiled kernel of
decode_gemm.hip and against the library GEMM + activation kernel, at decode batch sizes.

    python tools/swiglu_probe.py --model qwen2-1.5b --m 128,256,384,512

Weights rotate over enough copies (> 320 MB) that every launch streams them from HBM, as inside a decode step.
Every candidate is checked against an fp32 reference. Per M: library time, tiled-kernel time, and the balanced
kernel over column widths / stage counts.
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from areal_vllm_amd import ops  # noqa: E402
from areal_vllm_amd.models.config import preset  # noqa: E402
from areal_vllm_amd.ops import decode_gemm as dg  # noqa: E402
from tools.dgemm_probe import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen2-1.5b")
    ap.add_argument("--m", default="128,256,384,512")
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    cfg = preset(a.model)
    H, Fc = cfg.hidden_size, cfg.intermediate_size
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(0)
    n_copies = max(2, int(320e6 // (2 * Fc * H * 2)) + 1)
    W = (torch.randn(n_copies, 2 * Fc, H, device=dev, generator=g) * 0.03).to(torch.bfloat16)
    for M in [int(x) for x in a.m.split(",")]:
        X = torch.randn(n_copies, M, H, device=dev, generator=g).to(torch.bfloat16)
        ref = F.linear(X[0].float(), W[0].float())
        ref = ref.bfloat16().float()
        ref = (F.silu(ref[:, :Fc]) * ref[:, Fc:])
        out = torch.empty(M, Fc, dtype=torch.bfloat16, device=dev)

        def check(y, what):
            err = (y.float() - ref).abs().max().item()
            tol = 3e-2 * ref.abs().max().item() + 1e-3
            assert err <= tol, f"{what}: max err {err} > {tol}"

        t_lib = timeit(lambda c: ops.swiglu(F.linear(X[c], W[c]), "silu"), n_copies, a.iters)
        t_tiled = timeit(lambda c: dg.dgemm(X[c], W[c], epilogue=dg.EPI_SWIGLU, out=out), n_copies, a.iters)
        check(dg.dgemm(X[0], W[0], epilogue=dg.EPI_SWIGLU), "tiled")
        print(f"M={M:4d}  library+act {t_lib:7.1f} us   tiled kernel {t_tiled:7.1f} us   (weights {2 * Fc * H * 2 / 1e6:.0f} MB)")
        plan = dg.swiglu_plan(M, Fc)
        rows = []
        mbs = (1, 2) if 128 < M <= 256 else ((1,) if M <= 128 else (2,))
        for mblocks in mbs:
            for cols in (28, 32, 36, 40, 44, 48, 56, 64, 72, 76, 80):
                for stages in (2, 3, 4):
                    for nt in (True, False):
                        try:
                            y = dg.dgemm_swiglu_balanced(X[0], W[0], cols=cols, mblocks=mblocks, stages=stages, nt=nt)
                        except RuntimeError:
                            continue
                        check(y, f"balanced cols={cols} mb={mblocks} st={stages}")
                        t = timeit(lambda c: dg.dgemm_swiglu_balanced(X[c], W[c], out, cols=cols, mblocks=mblocks,
                                                                      stages=stages, nt=nt), n_copies, a.iters)
                        groups = -(-Fc // cols) * mblocks
                        rows.append((t, cols, mblocks, stages, nt, groups))
        rows.sort()
        for t, cols, mblocks, stages, nt, groups in rows[:6]:
            star = " <- default plan" if plan == (cols, mblocks) else ""
            print(f"        balanced {t:7.1f} us  cols={cols:3d} row blocks={mblocks} stages={stages} nt={int(nt)} "
                  f"workgroups={groups}{star}")
        if plan is not None:
            d = [r for r in rows if (r[1], r[2]) == plan]
            if d:
                print(f"        default plan cols={plan[0]} row blocks={plan[1]}: best {d[0][0]:.1f} us")


if __name__ == "__main__":
    main()
