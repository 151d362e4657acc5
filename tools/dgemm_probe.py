"""Sweep the decode GEMM (csrc/hip/decode_gemm.hip) over tile configs / LDS stages / split-K for the linear
layers of a model at decode batch sizes, against hipBLASLt (F.linear) on the same data.

    python tools/dgemm_probe.py --model qwen2-1.5b --m 64,128,256,384,512 [--full]

Weights rotate over enough copies (> 320 MB in total) that every launch streams them from HBM, as inside a
decode step, instead of from the 256 MB Infinity Cache. Each candidate is checked against an fp32 reference.
Prints, per (layer, M): the library time, the three fastest candidates and the time of the row kernel that
consumes the split-K slabs.
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from areal_vllm_amd.models.config import preset  # noqa: E402
from areal_vllm_amd.ops import decode_gemm as dg  # noqa: E402


def timeit(fn, n_copies, iters):
    """us per call of fn(c), c cycling over the copies; the calls are replayed from a captured graph so that
    the host's launch rate (10-20 us per call through Python here) does not hide kernels shorter than that."""
    for c in range(min(2, n_copies)):
        fn(c)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for c in range(n_copies):
                fn(c)
    torch.cuda.current_stream().wait_stream(side)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (iters * n_copies)  # us


def probe_qkv_rope(cfg, ms, iters, dev, g):
    """The qkv projection + bias + RoPE + KV-cache write: split-K GEMM followed by the slab row kernel (two
    launches) against the GEMM with that work in its epilogue (one launch), per tile configuration."""
    from areal_vllm_amd.ops import attention as A
    from areal_vllm_amd.ops.norm_act import RotaryTable

    H, D, hq, hkv = cfg.hidden_size, cfg.head_dim, cfg.n_q_heads, cfg.n_kv_heads
    nq = (hq + 2 * hkv) * D
    n_copies = max(2, int(320e6 // (nq * H * 2)) + 1)
    W = (torch.randn(n_copies, nq, H, device=dev, generator=g) * 0.03).to(torch.bfloat16)
    bias = torch.randn(nq, device=dev, generator=g).to(torch.bfloat16)
    table = RotaryTable(D, 4096, 1e6, dev)
    for M in ms:
        X = torch.randn(n_copies, M, H, device=dev, generator=g).to(torch.bfloat16)
        pages = (M * 4 + 31) // 32 + 1
        kc, vc = A.alloc_kv_cache(pages, hkv, D, torch.bfloat16, dev)
        pos = torch.randint(0, 4000, (M,), device=dev, dtype=torch.int32)
        slots = (torch.arange(M, device=dev) * 4).long()
        q = torch.empty(M, hq, D, dtype=torch.bfloat16, device=dev)
        c0, s0, k0, lw0, nt0 = dg.pick_config(M, nq, H, dg.EPI_SLAB)
        slab = torch.empty(k0, M, nq, dtype=torch.float32, device=dev)

        def two(c):
            dg.dgemm(X[c], W[c], epilogue=dg.EPI_SLAB, out=slab)
            dg.splitk_rope_cache_write(slab, k0, bias, pos, slots, table, kc, vc, hq, hkv, D, q_out=q)

        t2 = timeit(two, n_copies, iters)
        rows = []
        for cid in (9, 8, 3, 4):
            for stages in (3, 4, 6, 8):
                if stages * dg.stage_bytes(cid) > 160 * 1024:
                    continue
                for lw, nt in ((0, False), (0, True), (4, False), (4, True)):
                    bm, bn, _ = dg.DG_CONFIGS[cid]
                    if lw and ((bm + bn) // 8) % lw:
                        continue
                    fn = lambda c: dg.dgemm_qkv_rope(X[c], W[c], bias, pos, slots, table, kc, vc, hq, hkv, D, q_out=q,
                                                     cfg=cid, stages=stages, loader_waves=lw, nt=nt)
                    rows.append((timeit(fn, n_copies, iters), cid, stages, lw, int(nt)))
        rows.sort()
        td = timeit(lambda c: dg.dgemm_qkv_rope(X[c], W[c], bias, pos, slots, table, kc, vc, hq, hkv, D, q_out=q),
                    n_copies, iters)
        print(f"qkv_rope N={nq} K={H} M={M:4d}: split-K GEMM (cfg {c0}, ksplit {k0}) + row kernel {t2:6.1f} us; "
              f"fused default {td:6.1f} us; best fused:", flush=True)
        for t, cid, st, lw, nt in rows[:5]:
            bm, bn, thr = dg.DG_CONFIGS[cid]
            print(f"      {t:7.1f} us  cfg {cid} ({bm}x{bn}, {thr}+{64 * lw} thr) stages {st} nt {nt}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen2-1.5b")
    ap.add_argument("--m", default="64,128,256,384,512")
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--full", action="store_true", help="print every candidate, not only the best three")
    ap.add_argument("--layers", default="qkv,o,gate_up,down")
    ap.add_argument("--copies", type=int, default=0,
                    help="weight copies to rotate over (default: enough for > 320 MB, i.e. every launch streams "
                         "from HBM; 1 keeps the weights in the Infinity Cache)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = preset(args.model)
    H, D = cfg.hidden_size, cfg.head_dim
    nq = (cfg.n_q_heads + 2 * cfg.n_kv_heads) * D
    shapes = {"lm_head": (cfg.vocab_size // 32 * 32, H, dg.EPI_BF16),
              "qkv": (nq, H, dg.EPI_SLAB), "o": (H, cfg.n_q_heads * D, dg.EPI_SLAB),
              "gate_up": (2 * cfg.intermediate_size, H, dg.EPI_SWIGLU), "down": (H, cfg.intermediate_size, dg.EPI_SLAB)}
    g = torch.Generator(device=dev).manual_seed(0)
    if "qkv_rope" in args.layers.split(","):
        probe_qkv_rope(cfg, [int(x) for x in args.m.split(",")], args.iters, dev, g)
    for name in [n for n in args.layers.split(",") if n != "qkv_rope"]:
        N, K, epi = shapes[name]
        n_copies = args.copies or max(2, int(320e6 // (N * K * 2)) + 1)
        W = (torch.randn(n_copies, N, K, device=dev, generator=g) * 0.03).to(torch.bfloat16)
        for M in [int(x) for x in args.m.split(",")]:
            X = torch.randn(n_copies, M, K, device=dev, generator=g).to(torch.bfloat16)
            ref = X[0].float() @ W[0].float().t()
            if epi == dg.EPI_SWIGLU:
                f = N // 2
                refb = ref.to(torch.bfloat16).float()
                ref = F.silu(refb[:, :f]) * refb[:, f:]
            t_lib = timeit(lambda c: F.linear(X[c], W[c]), n_copies, args.iters)
            rows = []
            for cid, (bm, bn, thr) in dg.DG_CONFIGS.items():
                if bm > 2 * max(32, M) and bm > 64:
                    continue  # mostly padding
                for stages in range(2, dg.max_stages(cid) + 1):
                    ks_list = [1] if epi != dg.EPI_SLAB else sorted({dg.valid_ksplit(K, k) for k in
                                                                      (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20)})
                    for ks in ks_list:
                        cols = N // 2 if epi == dg.EPI_SWIGLU else N
                        per_tile = bn // 2 if epi == dg.EPI_SWIGLU else bn
                        blocks = ks * ((cols + per_tile - 1) // per_tile) * ((M + bm - 1) // bm)
                        if blocks < 64 or blocks > 2048:
                            continue
                        out = (torch.empty(n_copies, ks, M, N, dtype=torch.float32, device=dev) if epi == dg.EPI_SLAB
                               else torch.empty(n_copies, M, N // 2 if epi == dg.EPI_SWIGLU else N,
                                                dtype=torch.bfloat16, device=dev))
                        for lw, nt in ((0, False), (0, True), (4, False), (4, True)):
                            if lw and ((bm + bn) // 8) % lw:
                                continue
                            try:
                                fn = lambda c: dg.dgemm(X[c], W[c], epilogue=epi, cfg=cid, stages=stages, ksplit=ks,
                                                        out=out[c], loader_waves=lw, nt=nt)
                                out.zero_()
                                fn(0)
                                torch.cuda.synchronize()
                            except RuntimeError as e:
                                print(f"  {name} M={M} cfg={cid} st={stages} ks={ks} lw={lw}: {e}")
                                continue
                            got = out[0].sum(0) if epi == dg.EPI_SLAB else out[0].float()
                            err = (got - ref).abs().max().item()
                            scale = ref.abs().max().item()
                            t = timeit(fn, n_copies, args.iters)
                            rows.append((t, cid, stages, ks, blocks, err / max(scale, 1e-6), lw, int(nt)))
                        del out
            rows.sort()
            if not rows:
                print(f"{name:8s} N={N:6d} K={K:6d} M={M:4d}  lib {t_lib:7.1f} us   (no tile configuration within the "
                      f"64 .. 2048 workgroup range)", flush=True)
                continue
            print(f"{name:8s} N={N:6d} K={K:6d} M={M:4d}  lib {t_lib:7.1f} us   "
                  f"W {N * K * 2 / 1e6:6.1f} MB -> best {N * K * 2 / rows[0][0] / 1e6:6.2f} TB/s, "
                  f"{2.0 * M * N * K / rows[0][0] / 1e6:7.1f} TFLOP/s", flush=True)
            shown = rows if args.full else rows[:4]
            for lw, nt in ((0, 0), (0, 1), (4, 0), (4, 1)):  # best of every (loader waves, nt) arm
                arm = [r for r in rows if r[6] == lw and r[7] == nt]
                if arm and arm[0] not in shown:
                    shown.append(arm[0])
            for t, cid, st, ks, blocks, err, lw, nt in shown:
                bm, bn, thr = dg.DG_CONFIGS[cid]
                print(f"      {t:7.1f} us  cfg {cid} ({bm}x{bn}, {thr}+{64 * lw} thr) stages {st} ksplit {ks:2d} "
                      f"blocks {blocks:4d} nt {nt} rel.err {err:.1e}", flush=True)
            bad = [r for r in rows if r[5] > 2e-2]
            if bad:
                print(f"      !! {len(bad)} candidates off by more than 2e-2 relative: {bad[:4]}")
            if epi == dg.EPI_SLAB:
                t, cid, st, ks = rows[0][:4]
                slab = torch.randn(ks, M, N, dtype=torch.float32, device=dev)
                res = torch.randn(M, N, device=dev).to(torch.bfloat16)
                w = torch.ones(N, device=dev, dtype=torch.bfloat16)
                ro, y = torch.empty_like(res), torch.empty_like(res)
                tr = timeit(lambda c: dg.splitk_residual_rmsnorm(slab, ks, M, N, residual=res, weight=w, res_out=ro,
                                                                 out=y), 4, args.iters)
                print(f"      + slab sum + residual + rmsnorm row kernel at ksplit {ks}: {tr:6.1f} us", flush=True)
            del X
        del W


if __name__ == "__main__":
    main()
