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
    def forward(ctx, x, order, inv, k):
        ctx.save_for_backward(inv)
        ctx.k = k
        return _gather_rows(x, order // k)

    @staticmethod
    def backward(ctx, dxs):
        (inv,) = ctx.saved_tensors
        k = ctx.k
        return _combine_rows(dxs.contiguous(), inv, None, inv.numel() // k, k), None, None, None


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


def moe_permute(x: torch.Tensor, order: torch.Tensor, inv: torch.Tensor, k: int) -> torch.Tensor:
    """x [T, H] -> rows in expert order [T*k, H] (row p = token order[p] // k)."""
    return _PermuteFn.apply(x, order, inv, k)


def moe_combine(ys: torch.Tensor, w: torch.Tensor, order: torch.Tensor, inv: torch.Tensor) -> torch.Tensor:
    """Expert outputs in expert order ys [T*k, H], routing weights w [T, k] -> [T, H]."""
    return _CombineFn.apply(ys, w, order, inv)
