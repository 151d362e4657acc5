"""Variable-length causal attention (training / prefill) and paged-KV decode attention.

GPU: hand-written MFMA kernels (``csrc/hip/flash_attn.hip``, ``csrc/hip/paged_attn.hip``).
CPU: PyTorch reference used by unit tests and the gloo plumbing configuration.

Counterparts in the reference: ``flash_attn_varlen_func`` / ``flash_attn_with_kvcache`` call sites at
``realhf/impl/model/modules/attn.py:272,289`` and the HF flash-attention-2 path driven by
``areal/engine/base_hf_engine.py:291-406``.
"""
import math
from typing import Optional, Tuple

import torch

from . import _lib as L

PAGE_SIZE = 32  # tokens per KV page; fixed by the decode kernel's MFMA operand layout


# ----------------------------------------------------------------------- reference (CPU) paths
def attention_varlen_ref(q, k, v, cu_q, cu_k, scale, causal=True):
    """q [Tq, Hq, D], k/v [Tk, Hkv, D]; fp32 math; returns (out [Tq, Hq, D], lse [Hq, Tq])."""
    Tq, Hq, D = q.shape
    Hkv = k.shape[1]
    G = Hq // Hkv
    out = torch.zeros(Tq, Hq, D, dtype=torch.float32, device=q.device)
    lse = torch.full((Hq, Tq), float("-inf"), dtype=torch.float32, device=q.device)
    cu_q = cu_q.tolist()
    cu_k = cu_k.tolist()
    for b in range(len(cu_q) - 1):
        qs, qe, ks, ke = cu_q[b], cu_q[b + 1], cu_k[b], cu_k[b + 1]
        if qe == qs:
            continue
        qq = q[qs:qe].float().transpose(0, 1)  # [Hq, Lq, D]
        kk = k[ks:ke].float().transpose(0, 1).repeat_interleave(G, 0)
        vv = v[ks:ke].float().transpose(0, 1).repeat_interleave(G, 0)
        s = torch.matmul(qq, kk.transpose(1, 2)) * scale
        Lq, Lk = qe - qs, ke - ks
        if causal:
            i = torch.arange(Lq, device=q.device).unsqueeze(1)
            j = torch.arange(Lk, device=q.device).unsqueeze(0)
            s = s.masked_fill(j > i + (Lk - Lq), float("-inf"))
        l = torch.logsumexp(s, -1)
        p = torch.exp(s - l.unsqueeze(-1))
        p = torch.nan_to_num(p, nan=0.0)
        out[qs:qe] = torch.matmul(p, vv).transpose(0, 1)
        lse[:, qs:qe] = l
    return out.to(q.dtype), lse


def attention_varlen_sdpa(q, k, v, cu_q, cu_k, scale, causal=True, dropout_p: float = 0.0):
    """Head sizes the MFMA kernels do not take on the GPU (they are built for D = 128, the head size of the
    Qwen2 / Llama-3 8B+ families, and D = 256 of Gemma; other sizes that are multiples of 8 run on them
    zero-padded, see `_pad_heads`; what is left - above 256, or no multiple of 8 - comes here): PyTorch's fused
    attention, one call per packed sequence, the causal mask aligned to the bottom-right corner. Differentiable
    through autograd; returns out only."""
    import torch.nn.functional as F

    Hq, Hkv = q.shape[1], k.shape[1]
    cq, ck = cu_q.tolist(), cu_k.tolist()
    outs = []
    for b in range(len(cq) - 1):
        Lq, Lk = cq[b + 1] - cq[b], ck[b + 1] - ck[b]
        if Lq == 0:
            continue
        qs = q[cq[b]:cq[b + 1]].transpose(0, 1).unsqueeze(0)
        ks = k[ck[b]:ck[b + 1]].transpose(0, 1).unsqueeze(0)
        vs = v[ck[b]:ck[b + 1]].transpose(0, 1).unsqueeze(0)
        mask = None
        if causal and Lq != Lk:
            i = torch.arange(Lq, device=q.device).unsqueeze(1)
            j = torch.arange(Lk, device=q.device).unsqueeze(0)
            mask = j <= i + (Lk - Lq)
        o = F.scaled_dot_product_attention(qs, ks, vs, attn_mask=mask, is_causal=causal and mask is None, scale=scale,
                                           enable_gqa=Hq != Hkv, dropout_p=float(dropout_p))
        outs.append(o[0].transpose(0, 1))
    return torch.cat(outs, 0)


HIP_HEAD_DIM = 128  # the head size flash_attn.hip is built for
HIP_HEAD_DIMS = (128, 256)  # ... and flash_attn_d256.hip


def _pad_heads(t: torch.Tensor) -> torch.Tensor:
    """[.., D] -> [.., 128] (or [.., 256] for 128 < D < 256) with zeros. Zero columns change neither q.k nor the
    first D columns of p.v, so a smaller head (64 of Qwen2-0.5B, 80 / 96 of the Phi and small Llama families)
    runs on the same MFMA kernels, packed sequences and all, at the price of the idle part of every K-slab; the
    caller passes the softmax scale of the true head size and drops the padded output columns. Autograd slices
    the gradients back."""
    d = t.shape[-1]
    return torch.nn.functional.pad(t, (0, next(x for x in HIP_HEAD_DIMS if x >= d) - d))


def _paddable(d: int) -> bool:
    return d < HIP_HEAD_DIMS[-1] and d % 8 == 0 and d not in HIP_HEAD_DIMS


# ------------------------------------------------------------------------------- autograd op
class _FlashVarlenFn(torch.autograd.Function):
    """Attention over the fused projection qkv [T, Hq + 2*Hkv, D] (q heads | k heads | v heads)."""

    @staticmethod
    def forward(ctx, qkv, cu_seqlens, max_seqlen, n_q_heads, n_kv_heads, scale, causal):
        T, Htot, D = qkv.shape
        assert Htot == n_q_heads + 2 * n_kv_heads and qkv.is_contiguous()
        q = qkv[:, :n_q_heads]
        k = qkv[:, n_q_heads:n_q_heads + n_kv_heads]
        v = qkv[:, n_q_heads + n_kv_heads:]
        if L.use_hip(qkv):
            out, lse = _flash_fwd_hip(q, k, v, cu_seqlens, cu_seqlens, max_seqlen, scale, causal)
        else:
            out, lse = attention_varlen_ref(q, k, v, cu_seqlens, cu_seqlens, scale, causal)
        ctx.save_for_backward(qkv, out, lse, cu_seqlens)
        ctx.meta = (max_seqlen, n_q_heads, n_kv_heads, scale, causal)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse, cu = ctx.saved_tensors
        max_seqlen, Hq, Hkv, scale, causal = ctx.meta
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        q = qkv[:, :Hq]
        k = qkv[:, Hq:Hq + Hkv]
        v = qkv[:, Hq + Hkv:]
        dq = dqkv[:, :Hq]
        dk = dqkv[:, Hq:Hq + Hkv]
        dv = dqkv[:, Hq + Hkv:]
        if L.use_hip(qkv):
            _flash_bwd_hip(q, k, v, out, dout, lse, dq, dk, dv, cu, cu, max_seqlen, scale, causal)
        else:
            with torch.enable_grad():
                q_ = q.detach().float().requires_grad_(True)
                k_ = k.detach().float().requires_grad_(True)
                v_ = v.detach().float().requires_grad_(True)
                o_, _ = attention_varlen_ref(q_, k_, v_, cu, cu, scale, causal)
                gq, gk, gv = torch.autograd.grad(o_, (q_, k_, v_), dout.float())
            dq.copy_(gq)
            dk.copy_(gk)
            dv.copy_(gv)
        return dqkv, None, None, None, None, None, None


def flash_attn_varlen_qkv(qkv: torch.Tensor, cu_seqlens: torch.Tensor, max_seqlen: int, n_q_heads: int,
                          n_kv_heads: int, scale: Optional[float] = None, causal: bool = True) -> torch.Tensor:
    """Differentiable packed self-attention on a fused qkv tensor; returns [T, Hq, D]."""
    if scale is None:
        scale = 1.0 / math.sqrt(qkv.shape[-1])
    if L.use_hip(qkv) and qkv.shape[-1] not in HIP_HEAD_DIMS:
        d = qkv.shape[-1]
        if _paddable(d):
            return _FlashVarlenFn.apply(_pad_heads(qkv), cu_seqlens, int(max_seqlen), n_q_heads, n_kv_heads,
                                        float(scale), causal)[..., :d]
        q, k, v = qkv.split([n_q_heads, n_kv_heads, n_kv_heads], 1)
        return attention_varlen_sdpa(q, k, v, cu_seqlens, cu_seqlens, float(scale), causal)
    return _FlashVarlenFn.apply(qkv, cu_seqlens, int(max_seqlen), n_q_heads, n_kv_heads, float(scale), causal)


def _strides(t: torch.Tensor) -> int:
    assert t.stride(-1) == 1 and t.stride(-2) == t.shape[-1], "heads must be contiguous rows of D"
    return t.stride(0)


def _flash_fwd_hip(q, k, v, cu_q, cu_k, max_q, scale, causal) -> Tuple[torch.Tensor, torch.Tensor]:
    Tq, Hq, D = q.shape
    Hkv = k.shape[1]
    assert q.dtype == torch.bfloat16 and cu_q.dtype == torch.int32 and cu_k.dtype == torch.int32
    B = cu_q.numel() - 1
    out = torch.empty(Tq, Hq, D, dtype=q.dtype, device=q.device)
    lse = torch.empty(Hq, Tq, dtype=torch.float32, device=q.device)
    L.check(
        L.lib().areal_flash_fwd(
            L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out), L.ptr(lse), L.ptr(cu_q), L.ptr(cu_k), L.c_int(B),
            L.c_int(int(max_q)), L.c_int(Tq), L.c_int(Hq), L.c_int(Hkv), L.c_int(D), L.c_int64(_strides(q)),
            L.c_int64(_strides(k)), L.c_int64(_strides(v)), L.c_int64(_strides(out)), L.c_float(scale),
            L.c_int(int(causal)), L.stream_ptr(),
        ),
        "flash_fwd",
    )
    return out, lse


def _flash_bwd_hip(q, k, v, out, dout, lse, dq, dk, dv, cu_q, cu_k, max_k, scale, causal):
    Tq, Hq, D = q.shape
    Hkv = k.shape[1]
    B = cu_q.numel() - 1
    delta = torch.empty(Hq, Tq, dtype=torch.float32, device=q.device)
    # the fp32 accumulator of the atomic dQ path; long sequences compute dQ in a kernel of its own and need none
    split = D == 256 or bool(L.lib().areal_flash_bwd_split(L.c_int(int(max_k))))  # head size 256: always two kernels
    dq_acc = None if split else torch.zeros(Tq, Hq, D, dtype=torch.float32, device=q.device)
    L.check(
        L.lib().areal_flash_bwd(
            L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out), L.ptr(dout), L.ptr(lse), L.ptr(delta),
            L.ptr(dq_acc) if dq_acc is not None else None,
            L.ptr(dq), L.ptr(dk), L.ptr(dv), L.ptr(cu_q), L.ptr(cu_k), L.c_int(B), L.c_int(int(max_k)),
            L.c_int(Tq), L.c_int(Hq), L.c_int(Hkv), L.c_int(D), L.c_int64(_strides(q)), L.c_int64(_strides(k)),
            L.c_int64(_strides(v)), L.c_int64(_strides(out)), L.c_int64(_strides(dout)), L.c_int64(_strides(dq)),
            L.c_int64(_strides(dk)), L.c_int64(_strides(dv)), L.c_float(scale), L.c_int(int(causal)),
            L.stream_ptr(),
        ),
        "flash_bwd",
    )


class _FlashVarlenSepFn(torch.autograd.Function):
    """Attention with separate q [Tq, Hq, D] and k / v [Tk, Hkv, D] (views with contiguous heads), each
    batch entry attending q rows cu_q[b]:cu_q[b+1] to keys cu_k[b]:cu_k[b+1]; the causal mask is aligned
    to the bottom-right corner, i.e. the q rows are the LAST rows of their key range."""

    @staticmethod
    def forward(ctx, q, k, v, cu_q, cu_k, max_q, max_k, scale, causal):
        if L.use_hip(q):
            out, lse = _flash_fwd_hip(q, k, v, cu_q, cu_k, max_q, scale, causal)
        else:
            out, lse = attention_varlen_ref(q, k, v, cu_q, cu_k, scale, causal)
        ctx.save_for_backward(q, k, v, out, lse, cu_q, cu_k)
        ctx.meta = (max_k, scale, causal)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse, cu_q, cu_k = ctx.saved_tensors
        max_k, scale, causal = ctx.meta
        dout = dout.contiguous()
        if L.use_hip(q):
            dq = torch.empty(q.shape, dtype=q.dtype, device=q.device)
            dk = torch.zeros(k.shape, dtype=k.dtype, device=k.device)  # keys outside every range get 0
            dv = torch.zeros(v.shape, dtype=v.dtype, device=v.device)
            _flash_bwd_hip(q, k, v, out, dout, lse, dq, dk, dv, cu_q, cu_k, max_k, scale, causal)
        else:
            with torch.enable_grad():
                q_ = q.detach().float().requires_grad_(True)
                k_ = k.detach().float().requires_grad_(True)
                v_ = v.detach().float().requires_grad_(True)
                o_, _ = attention_varlen_ref(q_, k_, v_, cu_q, cu_k, scale, causal)
                dq, dk, dv = torch.autograd.grad(o_, (q_, k_, v_), dout.float())
            dq, dk, dv = dq.to(q.dtype), dk.to(k.dtype), dv.to(v.dtype)
        return dq, dk, dv, None, None, None, None, None, None


def flash_attn_varlen_sep(q, k, v, cu_q, cu_k, max_q: int, max_k: int, scale: Optional[float] = None,
                          causal: bool = True) -> torch.Tensor:
    """Differentiable varlen attention with separate q and k / v streams (context parallelism: a chunk
    of queries against the keys of its sequences up to the chunk's end)."""
    if scale is None:
        scale = 1.0 / math.sqrt(q.shape[-1])
    if L.use_hip(q) and q.shape[-1] not in HIP_HEAD_DIMS:
        d = q.shape[-1]
        if _paddable(d):
            return _FlashVarlenSepFn.apply(_pad_heads(q), _pad_heads(k), _pad_heads(v), cu_q, cu_k, int(max_q),
                                           int(max_k), float(scale), causal)[..., :d]
        return attention_varlen_sdpa(q, k, v, cu_q, cu_k, float(scale), causal)
    return _FlashVarlenSepFn.apply(q, k, v, cu_q, cu_k, int(max_q), int(max_k), float(scale), causal)


def flash_attn_varlen(q, k, v, cu_q, cu_k, max_q, scale=None, causal=True):
    """Forward-only varlen attention with separate q / k / v (prefill with a cached prefix:
    k/v longer than q, causal mask aligned bottom-right). Returns (out, lse)."""
    if scale is None:
        scale = 1.0 / math.sqrt(q.shape[-1])
    if L.use_hip(q):
        d = q.shape[-1]
        if _paddable(d):
            out, lse = _flash_fwd_hip(_pad_heads(q), _pad_heads(k), _pad_heads(v), cu_q, cu_k, max_q, scale, causal)
            return out[..., :d], lse
        if d not in HIP_HEAD_DIMS:  # no log-sum-exp on this path (the prefill does not need it)
            return attention_varlen_sdpa(q, k, v, cu_q, cu_k, float(scale), causal), None
        return _flash_fwd_hip(q, k, v, cu_q, cu_k, max_q, scale, causal)
    return attention_varlen_ref(q, k, v, cu_q, cu_k, scale, causal)


# ------------------------------------------------------------------------- paged KV cache ops
FP8 = torch.float8_e4m3fn  # element type of an fp8 KV cache (``kv_cache_dtype=fp8_e4m3``)
FP8_MAX = 448.0


def is_fp8(cache: torch.Tensor) -> bool:
    return cache.dtype == FP8


def to_fp8(x: torch.Tensor) -> torch.Tensor:
    """bf16 / fp32 -> e4m3 the way the kernels store it: round to nearest even, saturate at +-448."""
    return x.float().clamp(-FP8_MAX, FP8_MAX).to(FP8)


def alloc_kv_cache(num_pages: int, n_kv_heads: int, head_dim: int, dtype, device):
    """K pages [P, Hkv, 32, D]; V pages [P, Hkv, D, 32] (transposed, see paged_attn.hip). `dtype`: the model's
    compute dtype, or ``torch.float8_e4m3fn`` for a cache of e4m3 bytes (scale 1: post-rotary keys and values of
    the supported families stay well inside +-448; half the bytes to stream per decoded token, twice the tokens
    per GB; values are widened to bf16 inside the attention kernels, all arithmetic stays bf16 / fp32)."""
    k = torch.zeros(num_pages, n_kv_heads, PAGE_SIZE, head_dim, dtype=dtype, device=device)
    v = torch.zeros(num_pages, n_kv_heads, head_dim, PAGE_SIZE, dtype=dtype, device=device)
    return k, v


def rope_and_cache_write_(qkv, positions, slots, table, kcache, vcache, n_q_heads, n_kv_heads, do_rope=True):
    """In place: rotate q/k heads of qkv [T, Hq+2Hkv, D] and scatter k, v into the paged caches at
    `slots` (int64 [T], page*32+offset, negative = no write)."""
    T, Htot, D = qkv.shape
    assert qkv.is_contiguous() and Htot == n_q_heads + 2 * n_kv_heads
    if L.use_hip(qkv):
        assert positions.dtype == torch.int32 and slots.dtype == torch.int64
        L.check(
            L.lib().areal_rope_cache_write(
                L.ptr(qkv), L.ptr(positions), L.ptr(slots), L.ptr(table.cos), L.ptr(table.sin), L.ptr(kcache),
                L.ptr(vcache), L.c_int(T), L.c_int(n_q_heads), L.c_int(n_kv_heads), L.c_int(D),
                L.c_int(table.max_pos), L.c_int(int(do_rope)), L.c_int(int(is_fp8(kcache))), L.stream_ptr(),
            ),
            "rope_cache_write",
        )
        return qkv
    from .norm_act import rope_inplace_

    if do_rope:
        rope_inplace_(qkv, positions, table, n_q_heads + n_kv_heads)
    ok = slots >= 0
    if ok.any():
        pg = (slots[ok] // PAGE_SIZE).long()
        off = (slots[ok] % PAGE_SIZE).long()
        kk = qkv[ok, n_q_heads:n_q_heads + n_kv_heads]  # [n, Hkv, D]
        vv = qkv[ok, n_q_heads + n_kv_heads:]
        if is_fp8(kcache):  # index through the bytes: indexed assignment has no fp8 implementation
            kcache.view(torch.uint8)[pg, :, off, :] = to_fp8(kk).view(torch.uint8)
            vcache.view(torch.uint8)[pg, :, :, off] = to_fp8(vv).view(torch.uint8)
        else:
            kcache[pg, :, off, :] = kk
            vcache[pg, :, :, off] = vv
    return qkv


def cache_gather(kcache, vcache, slots):
    """Read K, V of `slots` (int64 [N]) back as contiguous [N, Hkv, D] tensors (bf16 from an fp8 cache)."""
    N = slots.numel()
    Hkv, D = kcache.shape[1], kcache.shape[3]
    fp8 = is_fp8(kcache)
    if L.use_hip(kcache):
        k = torch.empty(N, Hkv, D, dtype=torch.bfloat16 if fp8 else kcache.dtype, device=kcache.device)
        v = torch.empty_like(k)
        L.check(
            L.lib().areal_cache_gather(L.ptr(kcache), L.ptr(vcache), L.ptr(slots), L.ptr(k), L.ptr(v), L.c_int(N),
                                       L.c_int(Hkv), L.c_int(D), L.c_int64(Hkv * D), L.c_int(int(fp8)),
                                       L.stream_ptr()),
            "cache_gather",
        )
        return k, v
    pg = (slots // PAGE_SIZE).long()
    off = (slots % PAGE_SIZE).long()
    if fp8:
        return (kcache.view(torch.uint8)[pg, :, off, :].contiguous().view(FP8).to(torch.bfloat16),
                vcache.view(torch.uint8)[pg, :, :, off].contiguous().view(FP8).to(torch.bfloat16))
    return kcache[pg, :, off, :].contiguous(), vcache[pg, :, :, off].contiguous()


def context_slots(block_tables: torch.Tensor, cu_k: torch.Tensor) -> torch.Tensor:
    """Cache slot of every context token of every request, request after request (int64 [cu_k[-1]]): token t of
    request b lives in page ``block_tables[b, t // 32]``."""
    lens = (cu_k[1:] - cu_k[:-1]).long()
    seq = torch.repeat_interleave(torch.arange(lens.numel(), device=cu_k.device), lens)
    t = torch.arange(int(lens.sum()), device=cu_k.device) - cu_k[:-1].long()[seq]
    return block_tables.long()[seq, t // PAGE_SIZE] * PAGE_SIZE + t % PAGE_SIZE


def flash_attn_paged_prefill(q, kcache, vcache, block_tables, cu_q, cu_k, max_q: int, scale: Optional[float] = None,
                             causal: bool = True) -> torch.Tensor:
    """Attention of prefill queries ``q [T, Hq, D]`` (request b owns rows cu_q[b]:cu_q[b+1]) over their requests'
    whole contexts in the paged cache (``cu_k``: cumulative context lengths, query rows included - their K/V must
    have been written already; ``block_tables [B, max_pages]`` int32), causal mask aligned bottom-right. Head size
    128 runs ``csrc/hip/flash_attn_paged.hip`` straight on the pages; other head sizes and CPU tensors gather the
    context into contiguous tensors first. Role in the reference: ``flash_attn_with_kvcache``
    (``realhf/impl/model/modules/attn.py:272``) / the serving engine's radix-cache prefill."""
    T, Hq, D = q.shape
    Hkv = kcache.shape[1]
    scale = float(scale) if scale is not None else 1.0 / math.sqrt(D)
    if L.use_hip(q) and D == 128 and q.dtype == torch.bfloat16 and q.stride(2) == 1 and q.stride(1) == D:
        assert block_tables.dtype == torch.int32 and block_tables.is_contiguous()
        assert cu_q.dtype == torch.int32 and cu_k.dtype == torch.int32
        out = torch.empty(T, Hq, D, dtype=q.dtype, device=q.device)
        L.check(L.lib().areal_flash_fwd_paged(
            L.ptr(q), L.ptr(kcache), L.ptr(vcache), L.ptr(block_tables), L.c_int(block_tables.shape[1]), L.ptr(out),
            None, L.ptr(cu_q), L.ptr(cu_k), L.c_int(cu_q.numel() - 1), L.c_int(int(max_q)), L.c_int(T), L.c_int(Hq),
            L.c_int(Hkv), L.c_int(D), L.c_int64(q.stride(0)), L.c_int64(out.stride(0)), L.c_float(scale),
            L.c_int(int(causal)), L.c_int(int(is_fp8(kcache))), L.stream_ptr()), "flash_fwd_paged")
        return out
    k, v = cache_gather(kcache, vcache, context_slots(block_tables, cu_k))
    out, _ = flash_attn_varlen(q, k, v, cu_q, cu_k, max_q, scale=scale, causal=causal)
    return out


def paged_decode_ref(q, kcache, vcache, block_tables, context_lens, scale):
    B, Hq, D = q.shape
    Hkv = kcache.shape[1]
    G = Hq // Hkv
    out = torch.zeros(B, Hq, D, dtype=torch.float32, device=q.device)
    for b in range(B):
        n = int(context_lens[b])
        if n == 0:
            continue
        npg = (n + PAGE_SIZE - 1) // PAGE_SIZE
        pages = block_tables[b, :npg].long()
        if is_fp8(kcache):
            kp, vp = kcache.view(torch.uint8)[pages].view(FP8), vcache.view(torch.uint8)[pages].view(FP8)
        else:
            kp, vp = kcache[pages], vcache[pages]
        kk = kp.permute(1, 0, 2, 3).reshape(Hkv, npg * PAGE_SIZE, D)[:, :n].float()
        vv = vp.permute(1, 0, 3, 2).reshape(Hkv, npg * PAGE_SIZE, D)[:, :n].float()
        kk = kk.repeat_interleave(G, 0)
        vv = vv.repeat_interleave(G, 0)
        s = torch.einsum("hd,hnd->hn", q[b].float(), kk) * scale
        p = torch.softmax(s, -1)
        out[b] = torch.einsum("hn,hnd->hd", p, vv)
    return out.to(q.dtype)


# One workgroup per CU: measured on the decode graph (batch 32-384, context 640 and 1200) fewer, longer
# KV ranges beat more splits - every split adds a partial result to write and the reduce kernel to run.
DECODE_TARGET_WORKGROUPS = 256
# An fp8 cache halves the bytes a wave has in flight per page: with one workgroup per CU the kernel is then bound by
# load latency instead of bandwidth (128 rows x 8192 tokens: attention 3.6 ms at 256 workgroups, 2.8 at 512, 2.5 at
# 1024 against 4.7 ms for the bf16 cache, which gains nothing from more workgroups).
DECODE_TARGET_WORKGROUPS_FP8 = 1024


def decode_num_splits(batch: int, n_kv_heads: int, max_context: int, fp8: bool = False) -> int:
    """Split the KV range so that roughly >= DECODE_TARGET_WORKGROUPS workgroups are in flight."""
    base = max(1, batch * n_kv_heads)
    target = DECODE_TARGET_WORKGROUPS_FP8 if fp8 else DECODE_TARGET_WORKGROUPS
    want = max(1, target // base)
    max_useful = max(1, (max_context + 4 * PAGE_SIZE - 1) // (4 * PAGE_SIZE))
    return int(max(1, min(want, max_useful, 32)))


class DecodeWorkspace:
    """Preallocated split-KV partial buffers (stable addresses for hipGraph capture)."""

    def __init__(self, max_batch, n_q_heads, head_dim, max_splits, device):
        self.max_splits = max_splits
        self.part_o = torch.empty(max_batch, n_q_heads, max_splits, head_dim, dtype=torch.float32, device=device)
        self.part_ml = torch.empty(max_batch, n_q_heads, max_splits, 2, dtype=torch.float32, device=device)


def paged_attention_decode(q, kcache, vcache, block_tables, context_lens, scale=None, nsplit=1,
                           workspace: Optional[DecodeWorkspace] = None, out=None, q_row_stride=None):
    """q [B, Hq, D] (may be the q slice of a fused qkv row, pass q_row_stride in elements).
    block_tables int32 [B, max_pages]; context_lens int32 [B] (including the current token)."""
    B, Hq, D = q.shape
    Hkv = kcache.shape[1]
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    if not L.use_hip(q):
        return paged_decode_ref(q, kcache, vcache, block_tables, context_lens, scale)
    assert q.dtype == torch.bfloat16 and block_tables.dtype == torch.int32 and context_lens.dtype == torch.int32
    if out is None:
        out = torch.empty(B, Hq, D, dtype=q.dtype, device=q.device)
    if q_row_stride is None:
        q_row_stride = q.stride(0)
    part_o = part_ml = None
    if nsplit > 1:
        if workspace is None:
            workspace = DecodeWorkspace(B, Hq, D, nsplit, q.device)
        assert nsplit <= workspace.max_splits
        part_o, part_ml = workspace.part_o, workspace.part_ml
    L.check(
        L.lib().areal_paged_decode(
            L.ptr(q), L.ptr(kcache), L.ptr(vcache), L.ptr(block_tables), L.ptr(context_lens), L.ptr(out),
            L.ptr(part_o), L.ptr(part_ml), L.c_int(B), L.c_int(Hq), L.c_int(Hkv), L.c_int(D),
            L.c_int(block_tables.shape[1]), L.c_int(nsplit), L.c_float(scale), L.c_int(q_row_stride),
            L.c_int(int(is_fp8(kcache))), L.stream_ptr(),
        ),
        "paged_decode",
    )
    return out
