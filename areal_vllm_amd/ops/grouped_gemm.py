"""Grouped GEMM for expert-sorted tokens (MoE): ``out[t] = x[t] @ W[expert(t)]^T``.

Forward runs the MFMA kernel of ``csrc/hip/grouped_gemm.hip`` (one launch for all experts, tiles never
straddle experts). The backward pass computes ``dx`` with a sibling kernel that reads the weights untransposed and
``dW`` of all experts with one launch of the grouped weight-gradient kernel (expert shapes that are no multiples of
128: per expert with library GEMMs). Counterpart of ``grouped_gemm.ops.gmm`` in the reference
(``realhf/impl/model/modules/moe/experts.py:195-210``).
"""
import os
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
    tile_m = int(os.environ.get("AREAL_GROUPED_GEMM_TILE_M", "0"))  # 0: chosen from the rows per expert
    tm = L.lib().areal_grouped_gemm_tile_m(L.c_int(T), L.c_int(E), L.c_int(N), L.c_int(tile_m))
    L.check(L.lib().areal_grouped_gemm_nt(L.ptr(x), L.ptr(w), L.ptr(out), L.ptr(offsets), L.c_int(T), L.c_int(E),
                                          L.c_int(N), L.c_int(K), L.c_int64(w.stride(0)), L.c_int(T // tm + E),
                                          L.c_int(tile_m), L.stream_ptr()), "grouped_gemm_nt")
    return out


def _fold_target(w: torch.Tensor):
    """The fp32 `main_grad` view of a weight, when its gradient can be accumulated there directly."""
    mg = getattr(w, "main_grad", None)
    return mg if mg is not None and mg.dtype == torch.float32 and mg.is_contiguous() else None


def _after_fold(w: torch.Tensor):
    done = getattr(w, "_after_fold", None)
    if done is not None:  # the gradient went into the fp32 buffer without passing the accumulation hook
        done(w)


class _GroupedGemmFn(torch.autograd.Function):
    """Grouped MFMA kernels: forward, dx on the untransposed weights, dW of all experts in one launch."""

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
        E, N, K = w.shape
        if ctx.needs_input_grad[0]:
            if K % 128 == 0:
                # dx[t] = dy[t] @ W[e]: the weights are read as they are (transposed LDS reads in the kernel)
                T = dy.shape[0]
                dx = torch.empty(T, K, dtype=dy.dtype, device=dy.device)
                tile_m = int(os.environ.get("AREAL_GROUPED_GEMM_TILE_M", "0"))
                tm = L.lib().areal_grouped_gemm_tile_m(L.c_int(T), L.c_int(E), L.c_int(K), L.c_int(tile_m))
                L.check(L.lib().areal_grouped_gemm_nn(L.ptr(dy), L.ptr(w), L.ptr(dx), L.ptr(offsets), L.c_int(T),
                                                      L.c_int(E), L.c_int(N), L.c_int(K), L.c_int64(w.stride(0)),
                                                      L.c_int(T // tm + E), L.c_int(tile_m), L.stream_ptr()),
                        "grouped_gemm_nn")
            else:  # the forward kernel on a K-contiguous transposed copy of the weights
                dx = _launch(dy, w.transpose(1, 2).contiguous(), offsets)
        if ctx.needs_input_grad[1]:
            if N % 128 == 0 and K % 128 == 0 and x.shape[0] > 0:
                # dW[e] = dy_e^T . x_e for all experts in one launch (csrc/hip/grouped_gemm.hip); when the weight
                # carries an fp32 `main_grad` view the kernel adds into it (no bf16 dW, no separate add kernel)
                mg = _fold_target(w)
                dst = mg if mg is not None else torch.empty_like(w)
                L.check(L.lib().areal_grouped_gemm_wgrad(L.ptr(dy), L.ptr(x), L.ptr(dst), L.ptr(offsets),
                                                         L.c_int(x.shape[0]), L.c_int(E), L.c_int(N), L.c_int(K),
                                                         L.c_int64(dst.stride(0)), L.c_int(int(mg is not None)),
                                                         L.stream_ptr()), "grouped_gemm_wgrad")
                if mg is not None:
                    _after_fold(w)
                else:
                    dw = dst
            else:
                dw = torch.zeros_like(w)
                s = 0
                for e, n in enumerate(ctx.counts):
                    if n:
                        torch.mm(dy[s:s + n].t(), x[s:s + n], out=dw[e])
                    s += n
        return dx, dw, None, None


class _LibraryLoopFn(torch.autograd.Function):
    """One library GEMM per expert, forward and backward, written straight into slices of the outputs (no
    concatenation, no per-expert autograd nodes); dW is accumulated by the GEMM into the fp32 `main_grad` view of the
    weight when it has one (``ops/linear.py`` does the same for the dense layers)."""

    @staticmethod
    def forward(ctx, x, w, counts):
        ctx.save_for_backward(x, w)
        ctx.counts = counts
        out = torch.empty(x.shape[0], w.shape[1], dtype=x.dtype, device=x.device)
        s = 0
        for e, n in enumerate(counts):
            if n:
                torch.mm(x[s:s + n], w[e].t(), out=out[s:s + n])
            s += n
        return out

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            s = 0
            for e, n in enumerate(ctx.counts):
                if n:
                    torch.mm(dy[s:s + n], w[e], out=dx[s:s + n])
                s += n
        if ctx.needs_input_grad[1]:
            mg = _fold_target(w)
            fold = mg is not None and dy.is_cuda and dy.dtype in (torch.bfloat16, torch.float16)
            if not fold:
                dw = torch.zeros_like(w)
            s = 0
            for e, n in enumerate(ctx.counts):
                if n:
                    if fold:
                        torch.addmm(mg[e], dy[s:s + n].t(), x[s:s + n], out_dtype=torch.float32, out=mg[e])
                    else:
                        torch.mm(dy[s:s + n].t(), x[s:s + n], out=dw[e])
                s += n
            if fold:
                _after_fold(w)
        return dx, dw, None


# Few, large experts (Mixtral: 8 experts, thousands of rows each): one library GEMM per expert runs at 0.9-1.1 PFLOP/s,
# the grouped kernel at about half of that; many small experts (Qwen3-MoE: 128 experts, ~1000 rows each): the loop is
# launch-bound and the grouped kernel is 2-3x faster (tools/grouped_gemm_probe.py). AREAL_GROUPED_GEMM=hip|library
# overrides the choice.
LIBRARY_MAX_EXPERTS = 16
LIBRARY_MIN_ROWS_PER_EXPERT = 1024


def _prefer_library(n_rows: int, n_experts: int) -> bool:
    forced = os.environ.get("AREAL_GROUPED_GEMM", "auto")
    if forced in ("hip", "library"):
        return forced == "library"
    return n_experts <= LIBRARY_MAX_EXPERTS and n_rows >= LIBRARY_MIN_ROWS_PER_EXPERT * n_experts


def grouped_gemm_offsets_ok(x: torch.Tensor, w: torch.Tensor) -> bool:
    """Whether `grouped_gemm_offsets` serves these operands."""
    return _hip_ok(x, w) and x.shape[0] > 0


def grouped_gemm_offsets(x: torch.Tensor, w: torch.Tensor, offsets: torch.Tensor) -> torch.Tensor:
    """Forward-only grouped GEMM whose expert row ranges live on the device: x [T, K] rows sorted by expert,
    w [E, N, K], offsets int32 [E + 1] (offsets[e]..offsets[e + 1] = rows of expert e). The launch depends on shapes
    only, so no host synchronisation is needed and the call can be captured into a graph."""
    assert not torch.is_grad_enabled() or not (x.requires_grad or w.requires_grad)
    return _launch(x.contiguous(), w, offsets)


def grouped_gemm(x: torch.Tensor, w: torch.Tensor, counts: List[int]) -> torch.Tensor:
    """x [T, K] rows sorted by expert, w [E, N, K], counts[e] rows per expert (host ints, sum = T)."""
    if _hip_ok(x, w) and x.shape[0] > 0 and _prefer_library(x.shape[0], w.shape[0]):
        L.lib()  # on a GPU box the kernel library must exist even when a shape goes to the GEMM library
        return _LibraryLoopFn.apply(x.contiguous(), w, list(counts))
    if not _hip_ok(x, w) or x.shape[0] == 0:
        if x.is_cuda and x.dtype == torch.bfloat16:
            L.lib()  # on a GPU box the library must exist even when a shape falls back
        return grouped_gemm_ref(x, w, counts)
    offsets = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32).to(x.device, non_blocking=True)
    return _GroupedGemmFn.apply(x.contiguous(), w, offsets, list(counts))
