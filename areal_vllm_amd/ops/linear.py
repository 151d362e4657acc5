"""Linear layer whose weight gradient is accumulated by the GEMM itself into the fp32 main-grad buffer.

``y = x @ W^T (+ b)``. In the backward pass ``dW`` is not materialised in bf16 and added afterwards:
when the weight carries a ``main_grad`` view (see ``models/transformer.py::FlatParams.alloc_main_grad``)
the weight-gradient GEMM runs with bf16 operands, fp32 output and beta = 1 directly on that view
(hipBLASLt through ``torch.addmm(..., out_dtype=float32, out=main_grad)``), so micro-batch gradients are
summed in fp32 without a rounding step and without a separate add kernel. This is the role of apex's
``fused_weight_gradient_mlp_cuda.wgrad_gemm_accum_fp32`` in the reference
(``realhf/impl/model/parallelism/tensor_parallel/modules.py:242-402``).
"""
from typing import Optional

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import train_gemm


def _accumulate_wgrad(weight: torch.Tensor, dy2: torch.Tensor, x2: torch.Tensor) -> None:
    mg = weight.main_grad
    if dy2.is_cuda and train_gemm.wgrad_supported(dy2, x2, mg):
        # token-major operands reduced over tokens: the hand-written kernel (transposed LDS reads, split over tokens
        # when the gradient has few tiles) is 1.2-1.6x the library's kernel for this layout
        train_gemm.tgemm_wgrad_(mg, dy2, x2)
    elif dy2.is_cuda and mg.dtype == torch.float32 and dy2.dtype in (torch.bfloat16, torch.float16):
        torch.addmm(mg, dy2.t(), x2, out_dtype=torch.float32, out=mg)
    else:
        mg.add_(torch.mm(dy2.t(), x2))


class _LinearMainGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, reduce_dx_group):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.group = reduce_dx_group
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        dx, work = None, None
        if ctx.needs_input_grad[0]:
            dx = torch.mm(dy2, weight).view(x.shape)
            if ctx.group is not None:
                # column-parallel layer: the input gradient is a partial sum over the tensor-parallel
                # ranks. Its all-reduce is started here and runs on the communicator's stream while the
                # weight-gradient GEMM below keeps the compute units busy.
                work = dist.all_reduce(dx, group=ctx.group, async_op=True)
        main = getattr(weight, "main_grad", None) is not None
        dw = None
        if ctx.needs_input_grad[1]:
            if main:
                _accumulate_wgrad(weight, dy2, x.reshape(-1, x.shape[-1]))
                done = getattr(weight, "_after_fold", None)
                if done is not None:  # the gradient went into the fp32 buffer without passing the accumulation hook
                    done(weight)
            else:
                dw = torch.mm(dy2.t(), x.reshape(-1, x.shape[-1]))
        db = dy2.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        if work is not None:
            work.wait()
        return dx, dw, db, None


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
           reduce_dx_group=None) -> torch.Tensor:
    """`reduce_dx_group`: tensor-parallel group of a column-parallel layer whose input is replicated; the
    backward all-reduces the input gradient over it, overlapped with the weight-gradient GEMM (the
    reference's async-communication linear, ``realhf/impl/model/parallelism/tensor_parallel/modules.py:
    242-402``, without its dependence on a single hardware queue)."""
    if reduce_dx_group is not None and (not dist.is_initialized() or dist.get_world_size(reduce_dx_group) == 1):
        reduce_dx_group = None
    if torch.is_grad_enabled() and (reduce_dx_group is not None or (
            weight.requires_grad and getattr(weight, "main_grad", None) is not None)):
        return _LinearMainGrad.apply(x, weight, bias, reduce_dx_group)
    return F.linear(x, weight, bias)


class _EmbeddingMainGrad(torch.autograd.Function):
    """Embedding lookup whose backward adds the rows' gradients straight into the fp32 ``main_grad`` of the
    table. The stock backward materialises a dense [V, H] gradient (a zero fill plus a scatter) that the
    accumulation hook then reads once more to fold it into ``main_grad``: for a 152k x 1536 table that is
    ~1.4 GB of traffic per micro-batch to deliver T rows."""

    @staticmethod
    def forward(ctx, ids, weight):
        ctx.save_for_backward(ids)
        ctx.weight = weight
        return F.embedding(ids, weight)

    @staticmethod
    def backward(ctx, dy):
        (ids,) = ctx.saved_tensors
        w = ctx.weight
        w.main_grad.index_add_(0, ids.reshape(-1), dy.reshape(-1, dy.shape[-1]).to(w.main_grad.dtype))
        done = getattr(w, "_after_fold", None)
        if done is not None:  # this contribution never passes the accumulation hook: report it ourselves
            done(w)
        return None, None


def embedding(ids: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    if torch.is_grad_enabled() and weight.requires_grad and getattr(weight, "main_grad", None) is not None:
        return _EmbeddingMainGrad.apply(ids, weight)
    return F.embedding(ids, weight)
