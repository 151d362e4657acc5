"""Decode-batch GEMMs (M = 32 ... 512 rows) and the row kernels fused onto them (``csrc/hip/decode_gemm.hip``).

One decoder layer of a decode step becomes six launches instead of ten:

    x1 (normalised)  --dgemm_qkv_rope (bias + RoPE + KV-cache write in the epilogue)-->  q (rotated), paged K/V
    paged attention  --dgemm split-K-->  o slabs    --splitk_residual_rmsnorm-->  residual, x2
    x2  --dgemm + SwiGLU epilogue-->  act  --dgemm split-K-->  down slabs  --splitk_residual_rmsnorm--> residual, x1'

The split-K partial sums are plain fp32 slabs ``[split, M, N]`` summed in split order by the row kernel that
follows anyway (no atomics: replays of a captured graph are bit-identical). The reference gets this path from
SGLang's CUDA kernels (``areal/api/cli_args.py:398-454``).

CPU tensors take PyTorch paths with the same rounding points, so the engine's control flow is testable
without a GPU.
"""
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from . import _lib as L

EPI_BF16, EPI_SLAB, EPI_SWIGLU, EPI_GEGLU = 0, 1, 2, 3  # GEGLU: the SwiGLU epilogue with a tanh-GELU gate
GATED = (EPI_SWIGLU, EPI_GEGLU)


def gated_epilogue(act: str) -> int:
    """Epilogue id of the gated-MLP projection for a model's `hidden_act`."""
    return EPI_SWIGLU if act == "silu" else EPI_GEGLU


def _gate(g: torch.Tensor, epilogue: int) -> torch.Tensor:
    return F.silu(g) if epilogue == EPI_SWIGLU else F.gelu(g, approximate="tanh")

# cfg id -> (rows per block, weight columns per block, threads); must match areal_dgemm_block_shape
DG_CONFIGS = {0: (256, 64, 512), 1: (256, 128, 512), 2: (128, 128, 256), 3: (128, 64, 256), 4: (64, 128, 256),
              5: (256, 64, 256), 6: (256, 128, 512), 7: (256, 128, 256), 8: (64, 64, 128), 9: (32, 64, 64)}
N_CU = 256


def stage_bytes(cfg: int) -> int:
    bm, bn, _ = DG_CONFIGS[cfg]
    return (bm + bn) * 128


def max_stages(cfg: int) -> int:
    return min(4, (160 * 1024) // stage_bytes(cfg))


def valid_ksplit(K: int, ksplit: int) -> int:
    """Largest split count <= ksplit that leaves no empty split (splits are multiples of 64 deep)."""
    steps = K // 64
    ksplit = max(1, min(ksplit, steps))
    while ksplit > 1:
        per = (steps + ksplit - 1) // ksplit
        if (steps + per - 1) // per == ksplit:
            break
        ksplit -= 1
    return ksplit


def pick_config(M: int, N: int, K: int, epilogue: int) -> Tuple[int, int, int, int, bool]:
    """-> (cfg, stages, ksplit, loader_waves, nt), from the sweep on MI355X (``tools/dgemm_probe.py``, kept in
    ``profiles/dgemm_probe_qwen2-1.5b_s2.log``): what a block waits for is its CU's load path (~40 GB/s per CU
    measured), so small-N layers take small tiles and as many K splits as fill the chip once, wide layers take
    tiles as tall as the batch; four LDS stages where they fit; dedicated loader waves and non-temporal weight
    loads are worth 3-8 % except on the widest tiles."""
    override = _OVERRIDES.get((M, N, K, epilogue))
    if override is not None:
        return override
    wide = N >= 8192
    lw, nt = 4, True
    if wide:
        if M <= 64:
            cfg = 4
        elif M <= 128:
            cfg = 2
        elif M <= 256:
            cfg, lw = 1, 0
        else:
            cfg, lw, nt = 2, 0, False
    elif K >= 4096:
        cfg = 8 if M <= 64 else (3 if M <= 128 else 2)
    else:
        cfg = 8 if M <= 256 else 3
    bm, bn, _ = DG_CONFIGS[cfg]
    stages = max_stages(cfg)
    if wide and M > 256:
        stages = 2
    cols = N // 2 if epilogue in GATED else N
    per_tile = bn // 2 if epilogue in GATED else bn
    tiles = ((cols + per_tile - 1) // per_tile) * ((M + bm - 1) // bm)
    ksplit = 1
    if epilogue == EPI_SLAB:
        ksplit = valid_ksplit(K, max(1, min(N_CU // max(1, tiles), K // 192)))
    if lw and ((bm + bn) // 8) % lw:
        lw = 0
    return cfg, stages, ksplit, lw, nt


_OVERRIDES = {}


def dgemm(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, epilogue: int = EPI_BF16,
          cfg: Optional[int] = None, stages: Optional[int] = None, ksplit: Optional[int] = None,
          out: Optional[torch.Tensor] = None, loader_waves: Optional[int] = None,
          nt: Optional[bool] = None) -> torch.Tensor:
    """``x [M, K] @ w[N, K].T``. EPI_BF16 -> bf16 [M, N] (+bias); EPI_SLAB -> fp32 [ksplit, M, N] partial sums
    (bias NOT applied: the consuming row kernel adds it); EPI_SWIGLU (w = [gate; up]) -> bf16 [M, N/2]."""
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K
    if cfg is None:
        c, s, k, lw_, nt_ = pick_config(M, N, K, epilogue)
        cfg = c
        stages = s if stages is None else stages
        ksplit = k if ksplit is None else ksplit
        loader_waves = lw_ if loader_waves is None else loader_waves
        nt = nt_ if nt is None else nt
    stages = 3 if stages is None else min(stages, max_stages(cfg))
    ksplit = 1 if ksplit is None else ksplit
    loader_waves = 0 if loader_waves is None else loader_waves
    nt = bool(nt)
    if epilogue != EPI_SLAB:
        ksplit = 1
    ksplit = valid_ksplit(K, ksplit)
    if out is None:
        if epilogue == EPI_SLAB:
            out = torch.empty(ksplit, M, N, dtype=torch.float32, device=x.device)
        else:
            out = torch.empty(M, N // 2 if epilogue in GATED else N, dtype=x.dtype, device=x.device)
    if not L.use_hip(x):
        if epilogue == EPI_SLAB:
            steps = K // 64
            per = ((steps + ksplit - 1) // ksplit) * 64
            for s_ in range(ksplit):
                lo, hi = s_ * per, min(K, (s_ + 1) * per)
                out[s_, :M] = x[:, lo:hi].float() @ w[:, lo:hi].float().t()
            return out
        y = F.linear(x, w, bias)
        if epilogue in GATED:
            f = N // 2
            out.copy_((_gate(y[:, :f].float(), epilogue) * y[:, f:].float()).to(x.dtype))
        else:
            out.copy_(y)
        return out
    assert x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.stride(1) == 1 and w.is_contiguous()
    assert K % 64 == 0 and N % 32 == 0, f"decode GEMM needs K % 64 == 0 and N % 32 == 0, got N={N} K={K}"
    if epilogue == EPI_SLAB:
        assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() >= ksplit * M * N
        ys = N
    else:
        assert out.stride(1) == 1
        ys = out.stride(0)
    L.check(L.lib().areal_dgemm(L.ptr(x), L.c_int64(x.stride(0)), L.ptr(w), L.ptr(bias), L.ptr(out), L.c_int64(ys),
                                L.c_int(M), L.c_int(N), L.c_int(K), L.c_int(cfg), L.c_int(stages), L.c_int(epilogue),
                                L.c_int(ksplit), L.c_int(loader_waves), L.c_int(int(nt)), L.stream_ptr()), "dgemm")
    return out


def swiglu_plan(M: int, F: int) -> Optional[Tuple[int, int]]:
    """-> (columns per workgroup, row blocks) of the balanced gate|up kernel for M rows and F gate columns, or None
    where it has no instantiation. A workgroup stages (rows per block + 2 x columns) rows of K elements and the launch
    takes about 7 us + 0.065 us per staged row at K = 1536 (``profiles/swiglu_probe_s1.log``), so with row blocks x
    column ranges fixed at the number of CUs the best split has about twice as many block rows as columns: two row
    blocks from 129 rows on, ranges as narrow as fill the chip (a multiple of 4, at most 80 columns)."""
    if M > 512 or F % 4 != 0:
        return None
    for mblocks in ((2, 1) if 128 < M <= 256 else ((1,) if M <= 128 else (2,))):
        target = max(1, N_CU // mblocks)
        w = (-(-F // target) + 3) // 4 * 4
        if w <= 80:
            return w, mblocks
    return None


def dgemm_swiglu_balanced(x: torch.Tensor, w_gate_up: torch.Tensor, out: Optional[torch.Tensor] = None, *,
                          cols: Optional[int] = None, mblocks: Optional[int] = None, stages: int = 4,
                          nt: bool = True, act: str = "silu") -> torch.Tensor:
    """``silu(x @ gate.T) * (x @ up.T)`` with ``w_gate_up = [gate; up]`` ``[2F, K]`` -> bf16 ``[M, F]``, one workgroup
    per CU: column ranges of ``cols`` columns each (``csrc/hip/decode_swiglu.hip``). Same rounding points as
    ``dgemm(..., epilogue=EPI_SWIGLU)``."""
    M, K = x.shape
    F2 = w_gate_up.shape[0]
    Fc = F2 // 2
    if out is None:
        out = torch.empty(M, Fc, dtype=x.dtype, device=x.device)
    if not L.use_hip(x):
        y = F.linear(x, w_gate_up)
        y = y.to(x.dtype)
        out.copy_((_gate(y[:, :Fc].float(), gated_epilogue(act)) * y[:, Fc:].float()).to(x.dtype))
        return out
    plan = swiglu_plan(M, Fc)
    if cols is None or mblocks is None:
        assert plan is not None, f"no balanced gate|up decomposition for M={M}, F={Fc}"
        cols = plan[0] if cols is None else cols
        mblocks = plan[1] if mblocks is None else mblocks
    assert x.dtype == torch.bfloat16 and w_gate_up.dtype == torch.bfloat16 and x.stride(1) == 1
    assert w_gate_up.is_contiguous() and out.stride(1) == 1 and K % 64 == 0
    L.check(L.lib().areal_dswiglu(L.ptr(x), L.c_int64(x.stride(0)), L.ptr(w_gate_up), L.ptr(out),
                                  L.c_int64(out.stride(0)), L.c_int(M), L.c_int(Fc), L.c_int(K), L.c_int(int(cols)),
                                  L.c_int(int(mblocks)), L.c_int(int(stages)), L.c_int(int(bool(nt))),
                                  L.c_int(0 if act == "silu" else 1), L.stream_ptr()),
            "dswiglu")
    return out


def splitk_residual_rmsnorm(slab: torch.Tensor, n_split: int, M: int, N: int, *, bias: Optional[torch.Tensor] = None,
                            residual: Optional[torch.Tensor] = None, weight: Optional[torch.Tensor] = None,
                            eps: float = 1e-6, gemma: bool = False, res_out: Optional[torch.Tensor] = None,
                            out: Optional[torch.Tensor] = None, dtype=torch.bfloat16):
    """s = round(sum_k slab[k] + bias) (+ residual, rounded again); returns (rmsnorm(s) * weight, s).
    With ``weight=None`` the first value is s itself."""
    dev = slab.device
    if res_out is None:
        res_out = torch.empty(M, N, dtype=dtype, device=dev)
    if out is None:
        out = torch.empty(M, N, dtype=dtype, device=dev) if weight is not None else res_out
    if not L.use_hip(slab):
        s = slab.view(-1)[: n_split * M * N].view(n_split, M, N).sum(0)
        if bias is not None:
            s = s + bias.float()
        s = s.to(dtype)
        if residual is not None:
            s = (s.float() + residual[:M].float()).to(dtype)
        res_out[:M].copy_(s)
        if weight is not None:
            sf = s.float()
            rstd = torch.rsqrt(sf.pow(2).mean(-1, keepdim=True) + eps)
            out[:M].copy_((sf * rstd * (weight.float() + (1.0 if gemma else 0.0))).to(dtype))
        return out, res_out
    L.check(L.lib().areal_splitk_residual_rmsnorm(
        L.ptr(slab), L.c_int(n_split), L.ptr(bias), L.ptr(residual), L.ptr(weight), L.ptr(res_out), L.ptr(out),
        L.c_int(M), L.c_int(N), L.c_float(eps), L.c_int(int(bool(gemma))), L.stream_ptr()), "splitk_residual_rmsnorm")
    return out, res_out


def splitk_rope_cache_write(slab: torch.Tensor, n_split: int, bias: Optional[torch.Tensor], positions, slots, table,
                            kcache, vcache, n_q_heads: int, n_kv_heads: int, head_dim: int,
                            q_out: Optional[torch.Tensor] = None, do_rope: bool = True, dtype=torch.bfloat16,
                            q_norm: Optional[torch.Tensor] = None, k_norm: Optional[torch.Tensor] = None,
                            norm_eps: float = 1e-6):
    """Finish a split-K qkv projection: sum the slabs (+bias), [RMS-normalise every q / k head with ``q_norm`` /
    ``k_norm`` ``[D]`` (Qwen3),] rotate q / k, write k / v into the paged caches at ``slots`` and return the rotated
    q ``[T, Hq, D]``."""
    T = positions.numel()
    Nq = (n_q_heads + 2 * n_kv_heads) * head_dim
    if q_out is None:
        q_out = torch.empty(T, n_q_heads, head_dim, dtype=dtype, device=slab.device)
    if not L.use_hip(slab):
        from .attention import rope_and_cache_write_

        s = slab.view(-1)[: n_split * T * Nq].view(n_split, T, Nq).sum(0)
        if bias is not None:
            s = s + bias.float()
        qkv = s.to(dtype).view(T, n_q_heads + 2 * n_kv_heads, head_dim).contiguous()
        if q_norm is not None:
            def _n(x, w):
                xf = x.float()
                return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + norm_eps) * w.float()).to(dtype)

            qkv[:, :n_q_heads] = _n(qkv[:, :n_q_heads], q_norm)
            qkv[:, n_q_heads:n_q_heads + n_kv_heads] = _n(qkv[:, n_q_heads:n_q_heads + n_kv_heads], k_norm)
        rope_and_cache_write_(qkv, positions, slots, table, kcache, vcache, n_q_heads, n_kv_heads, do_rope)
        q_out.copy_(qkv[:, :n_q_heads])
        return q_out
    assert positions.dtype == torch.int32 and slots.dtype == torch.int64 and q_out.is_contiguous()
    L.check(L.lib().areal_splitk_rope_cache_write(
        L.ptr(slab), L.c_int(n_split), L.ptr(bias), L.ptr(positions), L.ptr(slots), L.ptr(table.cos), L.ptr(table.sin),
        L.ptr(q_out), L.ptr(kcache), L.ptr(vcache), L.c_int(T), L.c_int(n_q_heads), L.c_int(n_kv_heads),
        L.c_int(head_dim), L.c_int(table.max_pos), L.c_int(int(do_rope)), L.ptr(q_norm), L.ptr(k_norm),
        L.c_float(float(norm_eps)), L.c_int(int(kcache.dtype == torch.float8_e4m3fn)), L.stream_ptr()),
        "splitk_rope_cache_write")
    return q_out


# (cfg, stages, loader_waves, nt) of the fused qkv projection by batch rows, from ``tools/dgemm_probe.py --layers qkv_rope``
# (profiles/dgemm_probe_qkv_rope_s12.log): 32-row tiles up to 256 rows, 64-row tiles above; four dedicated loader waves
QKV_ROPE_CONFIG = {"small": (9, 4, 4, True), "large": (8, 4, 4, True)}
# Below this many rows the split-K GEMM + row kernel pair is as fast or faster: a block of the fused kernel walks
# all of K (24 steps of ~0.4 us at K = 1536), split-K blocks an eighth of it. Stand-alone 12.1 vs 14.0 us at 256
# rows and 13.5 vs 17.1 us at 384; inside the captured decode step the difference is within noise at 256 rows
# and 1 % of the step at 384 (profiles/decode_probe_fused_qkv_s12.log).
QKV_ROPE_MIN_ROWS = 320


def dgemm_qkv_rope(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], positions, slots, table, kcache,
                   vcache, n_q_heads: int, n_kv_heads: int, head_dim: int, q_out: Optional[torch.Tensor] = None,
                   do_rope: bool = True, cfg: Optional[int] = None, stages: Optional[int] = None,
                   loader_waves: Optional[int] = None, nt: Optional[bool] = None) -> torch.Tensor:
    """The packed q|k|v projection of a decode step in ONE launch: ``x @ w.T + bias`` rounded to bf16, q / k
    rotated, k / v written into the paged caches at ``slots`` (rows with slot < 0 write nothing); returns the
    rotated q ``[T, Hq, D]``. Replaces a split-K GEMM plus ``splitk_rope_cache_write`` (one launch and the fp32
    slab round trip less per layer)."""
    T, K = x.shape
    Nq = (n_q_heads + 2 * n_kv_heads) * head_dim
    assert w.shape == (Nq, K)
    if q_out is None:
        q_out = torch.empty(T, n_q_heads, head_dim, dtype=x.dtype, device=x.device)
    if not L.use_hip(x) or head_dim % 64 != 0 or kcache.dtype == torch.float8_e4m3fn:
        # (the GEMM's epilogue writes bf16 caches only: an fp8 cache takes the library GEMM + the row kernel)
        from .attention import rope_and_cache_write_

        qkv = F.linear(x, w, bias).view(T, n_q_heads + 2 * n_kv_heads, head_dim).contiguous()
        rope_and_cache_write_(qkv, positions, slots, table, kcache, vcache, n_q_heads, n_kv_heads, do_rope)
        q_out.copy_(qkv[:, :n_q_heads])
        return q_out
    if cfg is None:
        cfg, st_, lw_, nt_ = QKV_ROPE_CONFIG["small" if T <= 256 else "large"]
        stages = st_ if stages is None else stages
        loader_waves = lw_ if loader_waves is None else loader_waves
        nt = nt_ if nt is None else nt
    stages = min(4 if stages is None else stages, (160 * 1024) // stage_bytes(cfg), 8)
    if stages in (5, 7):
        stages -= 1
    assert x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.stride(1) == 1 and w.is_contiguous()
    assert positions.dtype == torch.int32 and slots.dtype == torch.int64 and q_out.is_contiguous()
    assert positions.numel() >= T and slots.numel() >= T
    L.check(L.lib().areal_dgemm_qkv_rope(
        L.ptr(x), L.c_int64(x.stride(0)), L.ptr(w), L.ptr(bias), L.ptr(positions), L.ptr(slots), L.ptr(table.cos),
        L.ptr(table.sin), L.ptr(q_out), L.ptr(kcache), L.ptr(vcache), L.c_int(T), L.c_int(K), L.c_int(n_q_heads),
        L.c_int(n_kv_heads), L.c_int(head_dim), L.c_int(table.max_pos), L.c_int(int(do_rope)), L.c_int(cfg),
        L.c_int(stages), L.c_int(int(loader_waves or 0)), L.c_int(int(bool(nt))), L.stream_ptr()), "dgemm_qkv_rope")
    return q_out
