"""Grouped GEMM for expert-sorted tokens (MoE): ``out[t] = x[t] @ W[expert(t)]^T``.

Forward runs the MFMA kernel of ``csrc/hip/grouped_gemm.hip`` (one launch for all experts, tiles never
straddle experts). The backward pass computes ``dx`` with the same kernel on the transposed weights and
``dW`` per expert with library GEMMs. Counterpart of ``grouped_gemm.ops.gmm`` in the reference
(``realhf/impl/model/modules/moe/experts.py:195-210``).
"""
from typing import List

import torch

from . import _lib as L


def grouped_gemm_ref(x: torch.Tensor, w: torch.Tensor, counts: List[int]) -> torch.Tensor:
    outs, s = [], 0
    for e, n in enumerate(counts):
        if n:
            outs.append(x[s:s + n] @ w[e].t())
        s += n
    return torch.cat(outs, 0) if outs else x.new_zeros(0, w.shape[1])


def _hip_ok(x, w) -> bool:
    return (x.is_cuda and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and w.shape[1] % 64 == 0
            and w.shape[2] % 64 == 0)


def _launch(x: torch.Tensor, w: torch.Tensor, offsets: torch.Tensor) -> torch.Tensor:
    T, K = x.shape
    E, N, K2 = w.shape
    assert K == K2 and offsets.numel() == E + 1 and offsets.dtype == torch.int32 and offsets.is_cuda
    assert x.is_contiguous() and w.stride(2) == 1 and w.stride(1) == K, "operands must be K-contiguous"
    out = torch.empty(T, N, dtype=x.dtype, device=x.device)
    L.check(L.lib().areal_grouped_gemm_nt(L.ptr(x), L.ptr(w), L.ptr(out), L.ptr(offsets), L.c_int(T), L.c_int(E),
                                          L.c_int(N), L.c_int(K), L.c_int64(w.stride(0)), L.c_int(T // 64 + E),
                                          L.stream_ptr()), "grouped_gemm_nt")
    return out


class _GroupedGemmFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, offsets, counts):
        ctx.save_for_backward(x, w, offsets)
        ctx.counts = counts
        return _launch(x, w, offsets)

    @staticmethod
    def backward(ctx, dy):
        x, w, offsets = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            # dx[t] = dy[t] @ W[e]  =  dy[t] @ (W[e]^T)^T : the same kernel on K-contiguous transposed weights
            dx = _launch(dy, w.transpose(1, 2).contiguous(), offsets)
        if ctx.needs_input_grad[1]:
            dw = torch.zeros_like(w)
            s = 0
            for e, n in enumerate(ctx.counts):
                if n:
                    torch.mm(dy[s:s + n].t(), x[s:s + n], out=dw[e])
                s += n
        return dx, dw, None, None


def grouped_gemm(x: torch.Tensor, w: torch.Tensor, counts: List[int]) -> torch.Tensor:
    """x [T, K] rows sorted by expert, w [E, N, K], counts[e] rows per expert (host ints, sum = T)."""
    if not _hip_ok(x, w) or x.shape[0] == 0:
        if x.is_cuda and x.dtype == torch.bfloat16:
            L.lib()  # on a GPU box the library must exist even when a shape falls back
        return grouped_gemm_ref(x, w, counts)
    offsets = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32).to(x.device, non_blocking=True)
    return _GroupedGemmFn.apply(x.contiguous(), w, offsets, list(counts))
