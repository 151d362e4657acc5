"""Policy log-prob / entropy over the vocabulary, fused with the LM head.

``gather_logprobs`` / ``gather_logprobs_entropy`` have the semantics of the reference's
``areal/utils/functional.py:10-80`` (``log_softmax(logits.float() / T)[label]`` and ``-sum p log p``),
but on GPU one HIP pass per row produces (max, sum exp, sum exp*z, target) so fp32 [T, V] is never
materialised. ``lm_head_logprobs`` additionally fuses the LM-head GEMM in row chunks: bf16 logits of a
chunk are produced by hipBLASLt, reduced by the kernel and (for training) kept in bf16 for the backward,
whose gradient kernel overwrites them in place with d(logits).

Tensor-parallel vocab shards (reference: ``realhf/impl/model/parallelism/tensor_parallel/modules.py:
1060-1177``, 3 all-reduces) exchange one [T, 4] tensor instead: see ``vocab_parallel_logprobs``.
"""
from typing import Optional, Tuple

import torch
import torch.distributed as dist

from . import _lib as L


def _stats_ref(logits, labels, inv_temp, vocab_start=0):
    z = logits.float() * inv_temp
    m = z.max(-1).values
    e = torch.exp(z - m.unsqueeze(-1))
    s = e.sum(-1)
    w = (e * z).sum(-1)
    lab = labels - vocab_start
    ok = (lab >= 0) & (lab < logits.shape[-1])
    tgt = torch.where(ok, z.gather(-1, lab.clamp(0, logits.shape[-1] - 1).unsqueeze(-1)).squeeze(-1),
                      torch.zeros_like(m))
    return m, s, w, tgt


def logprob_stats(logits: torch.Tensor, labels: torch.Tensor, temperature: float = 1.0, vocab_start: int = 0):
    """Per-row (m, s, w, tgt) over this vocab shard; z = logits / temperature."""
    assert logits.dim() == 2
    T, V = logits.shape
    inv_temp = 1.0 / temperature
    if not L.use_hip(logits):
        return _stats_ref(logits, labels, inv_temp, vocab_start)
    assert logits.stride(1) == 1 and labels.dtype == torch.int64
    outs = [torch.empty(T, dtype=torch.float32, device=logits.device) for _ in range(4)]
    L.check(
        L.lib().areal_logprob_stats(
            L.ptr(logits), L.c_int(L.dtype_flag(logits)), L.c_int64(T), L.c_int(V), L.c_int64(logits.stride(0)),
            L.ptr(labels), L.c_int(vocab_start), L.c_float(inv_temp), L.ptr(outs[0]), L.ptr(outs[1]),
            L.ptr(outs[2]), L.ptr(outs[3]), L.ptr(None), L.ptr(None), L.ptr(None), L.ptr(None), L.c_int(0),
            L.stream_ptr(),
        ),
        "logprob_stats",
    )
    return tuple(outs)


def _finalize(m, s, w, tgt):
    lse = m + torch.log(s)
    ez = w / s
    return tgt - lse, lse - ez, lse, ez


def _fwd(logits, labels, temperature):
    """Returns logp, entropy, lse, ez for full-vocab logits [T, V]."""
    T, V = logits.shape
    inv_temp = 1.0 / temperature
    if not L.use_hip(logits):
        return _finalize(*_stats_ref(logits, labels, inv_temp))
    assert logits.stride(1) == 1 and labels.dtype == torch.int64
    outs = [torch.empty(T, dtype=torch.float32, device=logits.device) for _ in range(4)]
    L.check(
        L.lib().areal_logprob_stats(
            L.ptr(logits), L.c_int(L.dtype_flag(logits)), L.c_int64(T), L.c_int(V), L.c_int64(logits.stride(0)),
            L.ptr(labels), L.c_int(0), L.c_float(inv_temp), L.ptr(None), L.ptr(None), L.ptr(None), L.ptr(None),
            L.ptr(outs[0]), L.ptr(outs[1]), L.ptr(outs[2]), L.ptr(outs[3]), L.c_int(1), L.stream_ptr(),
        ),
        "logprob_stats",
    )
    return tuple(outs)


def _bwd_inplace(logits, labels, temperature, lse, ez, dlogp, dent, vocab_start=0):
    """Overwrite `logits` with d(loss)/d(logits)."""
    T, V = logits.shape
    inv_temp = 1.0 / temperature
    if L.use_hip(logits):
        L.check(
            L.lib().areal_logprob_bwd(
                L.ptr(logits), L.ptr(logits), L.c_int(L.dtype_flag(logits)), L.c_int64(T), L.c_int(V),
                L.c_int64(logits.stride(0)), L.ptr(labels), L.c_int(vocab_start), L.c_float(inv_temp), L.ptr(lse),
                L.ptr(ez), L.ptr(dlogp), L.ptr(dent), L.stream_ptr(),
            ),
            "logprob_bwd",
        )
        return logits
    z = logits.float() * inv_temp
    p = torch.exp(z - lse.unsqueeze(-1))
    gl = dlogp.unsqueeze(-1) if dlogp is not None else 0.0
    g = -gl * p
    if dent is not None:
        g = g - dent.unsqueeze(-1) * p * (z - ez.unsqueeze(-1))
    lab = labels - vocab_start
    ok = (lab >= 0) & (lab < V)
    if dlogp is not None:
        idx = lab.clamp(0, V - 1).unsqueeze(-1)
        g.scatter_add_(-1, idx, (dlogp * ok).unsqueeze(-1))
    logits.copy_((g * inv_temp).to(logits.dtype))
    return logits


def _aligned_copy(logits: torch.Tensor) -> torch.Tensor:
    """Copy of `logits` whose rows start on 16-byte boundaries (row stride padded to 8 elements)."""
    T, V = logits.shape
    Vp = (V + 7) // 8 * 8
    buf = torch.empty(T, Vp, dtype=logits.dtype, device=logits.device)[:, :V]
    buf.copy_(logits)
    return buf


def _is_aligned(logits: torch.Tensor) -> bool:
    eb = logits.element_size()
    return logits.stride(1) == 1 and (logits.stride(0) * eb) % 16 == 0 and logits.data_ptr() % 16 == 0


class _LogitsLogprobFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, temperature, want_entropy):
        if logits.is_cuda and not _is_aligned(logits):
            logits = _aligned_copy(logits)
        logp, ent, lse, ez = _fwd(logits, labels, temperature)
        ctx.save_for_backward(logits, labels, lse, ez)
        ctx.temperature = temperature
        return logp, ent

    @staticmethod
    def backward(ctx, dlogp, dent):
        logits, labels, lse, ez = ctx.saved_tensors
        g = _aligned_copy(logits)
        _bwd_inplace(g, labels, ctx.temperature, lse, ez,
                     dlogp.contiguous().float() if dlogp is not None else None,
                     dent.contiguous().float() if dent is not None else None)
        return g, None, None, None


def gather_logprobs_entropy(logits: torch.Tensor, labels: torch.Tensor, temperature: float = 1.0):
    """logits [..., V] -> (logp [...], entropy [...]), differentiable w.r.t. logits."""
    shape = labels.shape
    lg = logits.reshape(-1, logits.shape[-1])
    logp, ent = _LogitsLogprobFn.apply(lg, labels.reshape(-1).long(), float(temperature), True)
    return logp.view(shape), ent.view(shape)


def gather_logprobs(logits: torch.Tensor, labels: torch.Tensor, temperature: float = 1.0):
    return gather_logprobs_entropy(logits, labels, temperature)[0]


# ------------------------------------------------------------------------ fused LM head + logprob
class _LMHeadLogprobFn(torch.autograd.Function):
    """hidden [T, H] x weight [V, H]^T -> (logp [T], entropy [T]) in row chunks.

    The bf16 logits of every chunk are kept for the backward (288 GB of HBM makes keeping them cheaper
    than recomputing the LM-head GEMM); the backward turns them into d(logits) in place and feeds two
    GEMMs. With `keep_logits=False` they are recomputed instead.
    """

    @staticmethod
    def forward(ctx, hidden, weight, labels, temperature, chunk, keep_logits):
        T = hidden.shape[0]
        logp = torch.empty(T, dtype=torch.float32, device=hidden.device)
        ent = torch.empty_like(logp)
        lse = torch.empty_like(logp)
        ez = torch.empty_like(logp)
        need_grad = hidden.requires_grad or weight.requires_grad
        kept = []
        for s in range(0, T, chunk):
            e = min(T, s + chunk)
            lg = torch.matmul(hidden[s:e], weight.t())
            a, b, c, d = _fwd(lg, labels[s:e], temperature)
            logp[s:e], ent[s:e], lse[s:e], ez[s:e] = a, b, c, d
            if need_grad and keep_logits:
                kept.append(lg)
        ctx.save_for_backward(hidden, weight, labels, lse, ez, *kept)
        ctx.meta = (temperature, chunk, keep_logits)
        return logp, ent

    @staticmethod
    def backward(ctx, dlogp, dent):
        hidden, weight, labels, lse, ez, *kept = ctx.saved_tensors
        temperature, chunk, keep_logits = ctx.meta
        T = hidden.shape[0]
        dlogp = dlogp.contiguous().float() if dlogp is not None else None
        dent = dent.contiguous().float() if dent is not None else None
        dh = torch.empty_like(hidden)
        dw = torch.zeros_like(weight) if ctx.needs_input_grad[1] else None
        for i, s in enumerate(range(0, T, chunk)):
            e = min(T, s + chunk)
            lg = kept[i] if keep_logits else torch.matmul(hidden[s:e], weight.t())
            _bwd_inplace(lg, labels[s:e], temperature, lse[s:e], ez[s:e],
                         dlogp[s:e] if dlogp is not None else None, dent[s:e] if dent is not None else None)
            torch.matmul(lg, weight, out=dh[s:e])
            if dw is not None:
                dw.addmm_(lg.t(), hidden[s:e])
        return dh, dw, None, None, None, None


def lm_head_logprobs(hidden: torch.Tensor, weight: torch.Tensor, labels: torch.Tensor, temperature: float = 1.0,
                     chunk: int = 8192, keep_logits: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """Fused LM head + log-prob + entropy. hidden [T, H] (bf16), weight [V, H]; labels int64 [T]."""
    return _LMHeadLogprobFn.apply(hidden, weight, labels.long(), float(temperature), int(chunk), keep_logits)


def vocab_parallel_logprobs(local_logits: torch.Tensor, labels: torch.Tensor, vocab_start: int,
                            group: Optional["dist.ProcessGroup"], temperature: float = 1.0):
    """Forward-only log-prob / entropy from a vocab shard [T, V/tp]: one all-gather of [T, 4] stats."""
    m, s, w, tgt = logprob_stats(local_logits, labels, temperature, vocab_start)
    st = torch.stack([m, s, w, tgt], -1)
    if group is not None and dist.get_world_size(group) > 1:
        allst = [torch.empty_like(st) for _ in range(dist.get_world_size(group))]
        dist.all_gather(allst, st, group=group)
        allst = torch.stack(allst, 0)  # [tp, T, 4]
        mm = allst[..., 0].max(0).values
        f = torch.exp(allst[..., 0] - mm)
        s = (allst[..., 1] * f).sum(0)
        w = (allst[..., 2] * f).sum(0)
        tgt = allst[..., 3].sum(0)
        m = mm
    return _finalize(m, s, w, tgt)
