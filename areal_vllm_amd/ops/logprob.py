"""Policy log-prob / entropy over the vocabulary, fused with the LM head.

``gather_logprobs`` / ``gather_logprobs_entropy`` have the semantics of the reference's
``areal/utils/functional.py:10-80`` (``log_softmax(logits.float() / T)[label]`` and ``-sum p log p``),
but on GPU one HIP pass per row produces (max, sum exp, sum exp*z, target) so fp32 [T, V] is never
materialised. ``lm_head_logprobs`` additionally fuses the LM-head GEMM in row chunks: bf16 logits of a
chunk are produced by hipBLASLt, reduced by the kernel and (for training) kept in bf16 for the backward,
whose gradient kernel overwrites them in place with d(logits).

Tensor-parallel vocab shards (reference: ``realhf/impl/model/parallelism/tensor_parallel/modules.py:
1060-1177``, 3 all-reduces) exchange one [4, T] tensor instead: ``lm_head_logprobs(..., vocab_start, group)``.
"""
from typing import Optional, Tuple

import torch
import torch.distributed as dist

from . import _lib as L
from . import train_gemm


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


def logprob_stats(logits: torch.Tensor, labels: torch.Tensor, temperature: float = 1.0, vocab_start: int = 0,
                  out: Optional[torch.Tensor] = None):
    """Per-row (m, s, w, tgt) over this vocab shard; z = logits / temperature. `labels` are global token
    ids: rows whose label lies outside [vocab_start, vocab_start + V) get tgt = 0. `out`: fp32 [4, T]
    (rows may be column slices of a larger buffer) that receives the four vectors."""
    assert logits.dim() == 2
    T, V = logits.shape
    inv_temp = 1.0 / temperature
    if not L.use_hip(logits):
        res = _stats_ref(logits, labels, inv_temp, vocab_start)
        if out is not None:
            for i in range(4):
                out[i].copy_(res[i])
            return tuple(out[i] for i in range(4))
        return res
    assert logits.stride(1) == 1 and labels.dtype == torch.int64
    if out is not None:
        assert out.dtype == torch.float32 and out.shape == (4, T) and out.stride(1) == 1
        outs = [out[i] for i in range(4)]
    else:
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
def vocab_shard_range(vocab: int, tp: int, rank: int) -> Tuple[int, int]:
    """Rows [lo, hi) of the vocabulary held by tensor-parallel rank `rank`: equal shards of a multiple of 8
    rows (16-byte aligned bf16 logit rows), the last one takes what is left."""
    if tp <= 1:
        return 0, vocab
    per = (-(-vocab // tp) + 7) // 8 * 8
    lo = min(vocab, rank * per)
    hi = min(vocab, lo + per)
    assert hi > lo, f"vocabulary of {vocab} rows leaves tensor-parallel rank {rank} of {tp} without a shard"
    return lo, hi


def _group_size(group) -> int:
    return dist.get_world_size(group) if group is not None and dist.is_initialized() else 1


def _chunk_logits(hidden: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    lg = torch.matmul(hidden, weight.t())
    if lg.is_cuda and not _is_aligned(lg):  # odd vocabulary (shard): rows must start on 16-byte boundaries
        lg = _aligned_copy(lg)
    return lg


def merge_shard_stats(stats: torch.Tensor, group) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """stats [4, T] = this shard's (max, sum exp, sum exp*z, target) rows -> the same four over the whole
    vocabulary. ONE all-gather of 16 bytes per token replaces the reference's three all-reduces
    (``realhf/impl/model/parallelism/tensor_parallel/modules.py:1060-1177``)."""
    n = _group_size(group)
    if n == 1:
        return stats[0], stats[1], stats[2], stats[3]
    allst = torch.empty((n,) + tuple(stats.shape), dtype=stats.dtype, device=stats.device)
    if stats.is_cuda:
        dist.all_gather_into_tensor(allst, stats.contiguous(), group=group)
    else:
        dist.all_gather(list(allst.unbind(0)), stats.contiguous(), group=group)
    m = allst[:, 0].max(0).values
    f = torch.exp(allst[:, 0] - m)
    return m, (allst[:, 1] * f).sum(0), (allst[:, 2] * f).sum(0), allst[:, 3].sum(0)


class _LMHeadLogprobFn(torch.autograd.Function):
    """hidden [T, H] x weight [V_local, H]^T -> (logp [T], entropy [T]) in row chunks.

    `weight` is the whole LM head or, with a tensor-parallel `group`, this rank's vocabulary rows starting
    at `vocab_start`: every rank reduces its own logits shard to four numbers per token, one all-gather
    merges them, and the backward forms d(logits) of the shard from the global log-sum-exp. d(hidden) is
    then a partial sum over shards: it is all-reduced here when `reduce_dh`, or left to the caller's
    reduce-scatter (sequence parallelism).

    The bf16 logits of every chunk are kept for the backward (288 GB of HBM makes keeping them cheaper
    than recomputing the LM-head GEMM); the backward turns them into d(logits) in place and feeds two
    GEMMs. With `keep_logits=False` they are recomputed instead.
    """

    @staticmethod
    def forward(ctx, hidden, weight, labels, temperature, chunk, keep_logits, vocab_start, group, reduce_dh):
        T = hidden.shape[0]
        stats = torch.empty(4, T, dtype=torch.float32, device=hidden.device)
        need_grad = hidden.requires_grad or weight.requires_grad
        kept = []
        for s in range(0, T, chunk):
            e = min(T, s + chunk)
            lg = _chunk_logits(hidden[s:e], weight)
            logprob_stats(lg, labels[s:e], temperature, vocab_start, out=stats[:, s:e])
            if need_grad and keep_logits:
                kept.append(lg)
        logp, ent, lse, ez = _finalize(*merge_shard_stats(stats, group))
        ctx.save_for_backward(hidden, weight, labels, lse, ez, *kept)
        ctx.meta = (temperature, chunk, keep_logits, vocab_start, group, reduce_dh)
        return logp, ent

    @staticmethod
    def backward(ctx, dlogp, dent):
        hidden, weight, labels, lse, ez, *kept = ctx.saved_tensors
        temperature, chunk, keep_logits, vocab_start, group, reduce_dh = ctx.meta
        T = hidden.shape[0]
        dlogp = dlogp.contiguous().float() if dlogp is not None else None
        dent = dent.contiguous().float() if dent is not None else None
        dh = torch.empty_like(hidden)
        dw = torch.zeros_like(weight) if ctx.needs_input_grad[1] else None
        # With an fp32 gradient buffer behind the weight, every chunk's d(logits)^T . hidden is summed straight into it
        # (fp32 across chunks, no [V, H] bf16 temporary, no separate accumulation pass).
        mg = getattr(weight, "main_grad", None) if dw is not None else None
        direct = mg is not None and hidden.is_cuda and mg.dtype == torch.float32 and mg.shape == weight.shape
        if direct:
            dw = None
        for i, s in enumerate(range(0, T, chunk)):
            e = min(T, s + chunk)
            lg = kept[i] if keep_logits else _chunk_logits(hidden[s:e], weight)
            _bwd_inplace(lg, labels[s:e], temperature, lse[s:e], ez[s:e],
                         dlogp[s:e] if dlogp is not None else None, dent[s:e] if dent is not None else None,
                         vocab_start)
            torch.matmul(lg, weight, out=dh[s:e])
            if direct:
                if train_gemm.wgrad_supported(lg, hidden[s:e], mg):
                    train_gemm.tgemm_wgrad_(mg, lg, hidden[s:e])
                else:
                    torch.addmm(mg, lg.t(), hidden[s:e], out_dtype=torch.float32, out=mg)
            elif dw is not None:
                dw.addmm_(lg.t(), hidden[s:e])
        if direct:
            done = getattr(weight, "_after_fold", None)
            if done is not None:  # this gradient never passes the accumulation hook
                done(weight)
        if reduce_dh and _group_size(group) > 1:
            dist.all_reduce(dh, group=group)
        return dh, dw, None, None, None, None, None, None, None


def lm_head_logprobs(hidden: torch.Tensor, weight: torch.Tensor, labels: torch.Tensor, temperature: float = 1.0,
                     chunk: int = 8192, keep_logits: bool = True, vocab_start: int = 0, group=None,
                     reduce_dh: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """Fused LM head + log-prob + entropy. hidden [T, H] (bf16), labels int64 [T] (global token ids);
    weight [V, H], or the vocabulary rows [vocab_start, vocab_start + V_local) of a tensor-parallel shard
    whose peers are the ranks of `group`."""
    return _LMHeadLogprobFn.apply(hidden, weight, labels.long(), float(temperature), int(chunk), keep_logits,
                                  int(vocab_start), group, bool(reduce_dh))
