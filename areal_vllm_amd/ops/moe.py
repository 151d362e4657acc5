"""Token routing glue of mixture-of-experts layers: rows into expert order and the weighted recombination of the
expert outputs, forward and backward as gathers (``csrc/hip/moe_ops.hip``; the token dispatchers of the reference:
``realhf/impl/model/modules/moe/token_dispatcher.py``).

`order` [T*k] is the expert-sorted order of the (token, choice) pairs (position p holds pair order[p] = t*k + j) and
`inv` [T*k] its inverse (pair t*k + j sits at position inv[t*k + j]). Every token occurs exactly k times, so the
scatter-adds of the textbook formulation (``index_add_`` for the combine, the indexing backward of ``x[tok]``) become
k-row gathers: each output row is written once, in a fixed order."""
import torch

from . import _lib as L


def _hip_rows(t: torch.Tensor) -> bool:
    return L.use_hip(t) and t.dtype == torch.bfloat16 and t.shape[-1] % 8 == 0


def _gather_rows(src: torch.Tensor, idx: torch.Tensor, scale=None) -> torch.Tensor:
    """out[p] = scale[p] * src[idx[p]]"""
    if not _hip_rows(src):
        out = src[idx]
        return out if scale is None else out * scale.unsqueeze(-1).to(out.dtype)
    src = src.contiguous()
    out = torch.empty(idx.numel(), src.shape[1], dtype=src.dtype, device=src.device)
    L.check(L.lib().areal_moe_gather_rows(L.ptr(src), L.ptr(idx), L.ptr(scale) if scale is not None else None,
                                          L.ptr(out), L.c_int64(idx.numel()), L.c_int(src.shape[1]), L.stream_ptr()),
            "moe_gather_rows")
    return out


def _combine_rows(src: torch.Tensor, inv: torch.Tensor, w, T: int, k: int) -> torch.Tensor:
    """out[t] = sum_j w[t, j] * src[inv[t*k + j]]   (w None: plain sum)"""
    if not _hip_rows(src) or k > 16:
        g = src[inv].view(T, k, -1)
        if w is not None:
            g = g * w.view(T, k, 1).to(g.dtype)
        return g.sum(1, dtype=torch.float32).to(src.dtype)
    src = src.contiguous()
    out = torch.empty(T, src.shape[1], dtype=src.dtype, device=src.device)
    L.check(L.lib().areal_moe_combine_rows(L.ptr(src), L.ptr(inv), L.ptr(w) if w is not None else None, L.ptr(out),
                                           L.c_int64(T), L.c_int(k), L.c_int(src.shape[1]), L.stream_ptr()),
            "moe_combine_rows")
    return out


class _PermuteFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, order, inv, k, src_row=None):
        ctx.save_for_backward(inv)
        ctx.k = k
        return _gather_rows(x, order // k if src_row is None else src_row)

    @staticmethod
    def backward(ctx, dxs):
        (inv,) = ctx.saved_tensors
        k = ctx.k
        return _combine_rows(dxs.contiguous(), inv, None, inv.numel() // k, k), None, None, None, None


class _CombineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ys, w, order, inv):
        T, k = w.shape
        ctx.w_dtype = w.dtype
        w = w.to(ys.dtype).contiguous()
        ctx.save_for_backward(ys, w, order, inv)
        return _combine_rows(ys, inv, w, T, k)

    @staticmethod
    def backward(ctx, dout):
        ys, w, order, inv = ctx.saved_tensors
        T, k = w.shape
        dout = dout.contiguous()
        dys = dw = None
        if ctx.needs_input_grad[0]:  # dys[p] = w[pair at p] * dout[token at p]
            dys = _gather_rows(dout, order // k, w.reshape(-1)[order].contiguous())
        if ctx.needs_input_grad[1]:  # dw[t, j] = <dout[t], ys[inv[t*k + j]]>
            if _hip_rows(ys):
                dw = torch.empty(T, k, dtype=ys.dtype, device=ys.device)
                L.check(L.lib().areal_moe_combine_dots(L.ptr(dout), L.ptr(ys), L.ptr(inv), L.ptr(dw), L.c_int64(T),
                                                       L.c_int(k), L.c_int(ys.shape[1]), L.stream_ptr()),
                        "moe_combine_dots")
            else:
                dw = (ys[inv].view(T, k, -1).float() * dout.float().unsqueeze(1)).sum(-1).to(w.dtype)
        return dys, (dw.to(ctx.w_dtype) if dw is not None else None), None, None


def moe_permute(x: torch.Tensor, order: torch.Tensor, inv: torch.Tensor, k: int, src_row=None) -> torch.Tensor:
    """x [T, H] -> rows in expert order [T*k, H] (row p = token order[p] // k = src_row[p] when the sort handed
    it out)."""
    return _PermuteFn.apply(x, order, inv, k, src_row)


def moe_topk(logits: torch.Tensor, k: int, norm: bool, out_dtype=torch.float32):
    """Softmax over the experts and its k largest entries per token, without autograd (serving, scoring passes):
    logits [T, E] fp32 or bf16 (the arithmetic is fp32 either way) -> (weights [T, k] in `out_dtype` - softmax
    probabilities, re-normalised over the k choices when `norm` -, expert ids int64 [T, k]), choices in descending
    order of probability. One launch on the GPU."""
    T, E = logits.shape
    if not (L.use_hip(logits) and logits.dtype in (torch.float32, torch.bfloat16) and E <= 512 and k <= min(64, E)
            and out_dtype in (torch.float32, torch.bfloat16)):
        topv, topi = torch.topk(torch.softmax(logits.float(), -1), k, -1)
        if norm:
            topv = topv / topv.sum(-1, keepdim=True)
        return topv.to(out_dtype), topi
    logits = logits.contiguous()
    topi = torch.empty(T, k, dtype=torch.int64, device=logits.device)
    topv = torch.empty(T, k, dtype=out_dtype, device=logits.device)
    f32 = out_dtype == torch.float32
    L.check(L.lib().areal_moe_topk(L.ptr(logits), L.c_int(L.dtype_flag(logits)), L.c_int64(T), L.c_int(E), L.c_int(k),
                                   L.c_int(int(bool(norm))), L.ptr(topi), L.ptr(topv) if f32 else None,
                                   None if f32 else L.ptr(topv), L.stream_ptr()), "moe_topk")
    return topv, topi


def moe_sort(flat_e: torch.Tensor, n_experts: int, k: int):
    """Expert-sorted order of the (token, choice) pairs: flat_e int64 [T*k] with ids 0 .. n_experts (id n_experts =
    dropped pair, sorted last) -> (order [T*k], inv [T*k], src_row [T*k] = order // k, all int64; offsets int32
    [n_experts + 2]: expert e owns positions offsets[e] .. offsets[e + 1], the dropped pairs sit from
    offsets[n_experts] on). The order is the stable one (ties in pair order), what ``torch.argsort(stable=True)``
    gives; on the GPU a counting sort in one launch (lists of a few thousand pairs: a decode batch) or three."""
    n = flat_e.numel()
    bins = n_experts + 1
    if not (L.use_hip(flat_e) and flat_e.dtype == torch.int64 and bins <= 513):
        order = torch.argsort(flat_e, stable=True)
        inv = torch.empty_like(order)
        inv[order] = torch.arange(n, device=flat_e.device)
        counts = torch.bincount(flat_e, minlength=bins)[:bins]
        offsets = torch.zeros(bins + 1, dtype=torch.int32, device=flat_e.device)
        offsets[1:] = torch.cumsum(counts, 0)
        return order, inv, order // k, offsets
    flat_e = flat_e.contiguous()
    lib = L.lib()
    order = torch.empty(n, dtype=torch.int64, device=flat_e.device)
    inv = torch.empty_like(order)
    src_row = torch.empty_like(order)
    offsets = torch.empty(bins + 1, dtype=torch.int32, device=flat_e.device)
    items, ws = 1, None
    if not int(lib.areal_moe_sort_single(L.c_int64(n), L.c_int(bins))):
        items = max(1, -(-n // (256 * 128)))  # at most 128 chunks
        chunks = -(-n // (256 * items))
        ws = torch.empty(chunks * bins, dtype=torch.int32, device=flat_e.device)
    L.check(lib.areal_moe_sort(L.ptr(flat_e), L.c_int64(n), L.c_int(bins), L.c_int(k), L.ptr(order), L.ptr(inv),
                               L.ptr(src_row), L.ptr(offsets), L.ptr(ws), L.c_int(items), L.stream_ptr()),
            "moe_sort")
    return order, inv, src_row, offsets


def moe_combine(ys: torch.Tensor, w: torch.Tensor, order: torch.Tensor, inv: torch.Tensor) -> torch.Tensor:
    """Expert outputs in expert order ys [T*k, H], routing weights w [T, k] -> [T, H]."""
    return _CombineFn.apply(ys, w, order, inv)
