"""Sequence-parallel linears whose collective travels while the GEMM runs.

With sequence parallelism a column-parallel linear is ``all-gather(rows) -> GEMM`` and a row-parallel one is
``GEMM -> reduce-scatter(rows)``; issued one after the other the links idle during the GEMM and the matrix cores
idle during the collective. Here the token rows of every rank are cut into a few chunks and the two are pipelined:

* ``gather_linear``: chunk c of every rank's rows is pushed to all peers (one peer-memory exchange kernel on a side
  stream) while the GEMM of chunk c - 1 runs; only the first chunk's transfer is exposed.
* ``linear_scatter``: the GEMM of chunk c writes partial sums that the side stream reduce-scatters (rank-ordered bf16
  sums) while chunk c + 1 is computed; only the last chunk's transfer is exposed.

The backward passes are the mirror images (the input gradient of one is the other's forward), the weight gradients
are accumulated into the fp32 ``main_grad`` from the gathered operand. The exchange object belongs to these
functions alone: its calls all sit on one side stream, in program order, as the kernel's slot reuse requires.

Role in the reference: the asynchronous all-gather / reduce-scatter linears of Megatron-style tensor parallelism
(``realhf/impl/model/parallelism/tensor_parallel/modules.py:242-402``). Bandwidth over xGMI is unmeasured (the ranks
of the tests share one GPU).
"""
import os
from typing import List, Optional, Tuple

import torch

from ..ops.linear import _accumulate_wgrad
from ..ops.peer_collectives import SCATTER, SUM_BF16, SUM_F32, PeerCollectives

CHUNKS = int(os.environ.get("AREAL_SP_OVERLAP_CHUNKS", "4"))
# rows of one rank per chunk below which a chunk's GEMMs stop filling the device
MIN_ROWS = int(os.environ.get("AREAL_SP_OVERLAP_MIN_ROWS", "256"))


class SPOverlap:
    def __init__(self, pc: PeerCollectives, chunks: Optional[int] = None, min_rows: Optional[int] = None):
        self.pc, self.n, self.rank = pc, pc.n, pc.rank
        self.stream = torch.cuda.Stream(device=pc.device)
        self.chunks = int(chunks or CHUNKS)
        self.min_rows = int(min_rows or MIN_ROWS)

    def plan(self, t: int, row_bytes: int) -> List[Tuple[int, int]]:
        """Row ranges [r0, r1) of one rank's `t` rows, one per pipeline step."""
        k = max(1, min(self.chunks, t // self.min_rows))
        k = max(k, -(-self.n * t * row_bytes // self.pc.slot_bytes))  # a chunk from every rank must fit a slot
        step = -(-t // k)
        return [(r0, min(t, r0 + step)) for r0 in range(0, t, step)]

    def usable(self, x: torch.Tensor) -> bool:
        return (x.is_cuda and x.dim() == 2 and x.dtype in (torch.bfloat16, torch.float32)
                and x.shape[1] * x.element_size() % 16 == 0)

    # ------------------------------------------------------------------ building blocks
    def gather(self, x: torch.Tensor, events: Optional[list] = None) -> torch.Tensor:
        """All-gather the rows of `x` on the side stream, chunk by chunk; `events[i]` fires when chunk i of EVERY
        rank has landed. The caller orders its stream after the events (or after the side stream) before reading."""
        t, width = x.shape
        rb = width * x.element_size()
        full = x.new_empty((self.n * t, width))
        full.record_stream(self.stream)
        x.record_stream(self.stream)
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            for r0, r1 in self.plan(t, rb):
                nb = (r1 - r0) * rb
                self.pc.exchange(x, full, nb, nb, 0, nb, t * rb, nb, SCATTER, in_offset=r0 * rb, out_offset=r0 * rb)
                if events is not None:
                    events.append(self.stream.record_event())
        return full

    def gather_matmul(self, x: torch.Tensor, w_right: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """(all-gather(x) @ w_right, all-gather(x)); `w_right` is [in, out] (any strides)."""
        t, width = x.shape
        n = self.n
        events: list = []
        full = self.gather(x.contiguous(), events)
        y = x.new_empty((n * t, w_right.shape[1]))
        cur = torch.cuda.current_stream()
        f3, y3 = full.view(n, t, width), y.view(n, t, -1)
        for (r0, r1), ev in zip(self.plan(t, width * x.element_size()), events):
            cur.wait_event(ev)
            for s in range(n):  # rows r0..r1 of every rank's block
                torch.mm(f3[s, r0:r1], w_right, out=y3[s, r0:r1])
        return y, full

    def matmul_scatter(self, h: torch.Tensor, w_right: torch.Tensor) -> torch.Tensor:
        """This rank's rows of sum over ranks of (h @ w_right): [n t, in] -> [t, out]."""
        T, _ = h.shape
        n = self.n
        assert T % n == 0
        t, width = T // n, w_right.shape[1]
        rb = width * h.element_size()
        partial = h.new_empty((T, width))
        out = h.new_empty((t, width))
        partial.record_stream(self.stream)
        out.record_stream(self.stream)
        cur = torch.cuda.current_stream()
        h3, p3 = h.view(n, t, -1), partial.view(n, t, width)
        mode = SUM_BF16 if h.dtype == torch.bfloat16 else SUM_F32
        for r0, r1 in self.plan(t, rb):
            for s in range(n):
                torch.mm(h3[s, r0:r1], w_right, out=p3[s, r0:r1])
            ev = cur.record_event()
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                nb = (r1 - r0) * rb
                self.pc.exchange(partial, out, nb, nb, t * rb, nb, 0, nb, mode, in_offset=r0 * rb, out_offset=r0 * rb)
        cur.wait_stream(self.stream)
        return out


def _fold_wgrad(weight: torch.Tensor, dy: torch.Tensor, x: torch.Tensor):
    """dW = dy^T x: into the fp32 main_grad when the weight has one (returns None), else returned."""
    if getattr(weight, "main_grad", None) is not None:
        _accumulate_wgrad(weight, dy, x)
        done = getattr(weight, "_after_fold", None)
        if done is not None:
            done(weight)
        return None
    return torch.mm(dy.t(), x)


class _GatherLinear(torch.autograd.Function):
    """y = all-gather(x) @ W^T for a column-parallel W [out_shard, in]."""

    @staticmethod
    def forward(ctx, x, weight, ov: SPOverlap):
        ctx.ov = ov
        ctx.save_for_backward(x, weight)
        y, _ = ov.gather_matmul(x, weight.t())
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        ov: SPOverlap = ctx.ov
        dy = dy.contiguous()
        # the gathered input is needed again for dW: it travels first, under the input-gradient GEMMs
        full = ov.gather(x.contiguous()) if ctx.needs_input_grad[1] else None
        dx = ov.matmul_scatter(dy, weight) if ctx.needs_input_grad[0] else None
        dw = None
        if full is not None:
            torch.cuda.current_stream().wait_stream(ov.stream)
            dw = _fold_wgrad(weight, dy, full)
        return dx, dw, None


class _LinearScatter(torch.autograd.Function):
    """y = reduce-scatter(h @ W^T) for a row-parallel W [out, in_shard]."""

    @staticmethod
    def forward(ctx, h, weight, ov: SPOverlap):
        ctx.ov = ov
        ctx.save_for_backward(h, weight)
        return ov.matmul_scatter(h.contiguous(), weight.t())

    @staticmethod
    def backward(ctx, dy):
        h, weight = ctx.saved_tensors
        ov: SPOverlap = ctx.ov
        dh, dy_full = ov.gather_matmul(dy.contiguous(), weight)  # [T, out] @ [out, in_shard]
        dw = _fold_wgrad(weight, dy_full, h.contiguous()) if ctx.needs_input_grad[1] else None
        return (dh if ctx.needs_input_grad[0] else None), dw, None


def gather_linear(x: torch.Tensor, weight: torch.Tensor, ov: SPOverlap, bias: Optional[torch.Tensor] = None):
    y = _GatherLinear.apply(x, weight, ov)
    return y if bias is None else y + bias


def linear_scatter(h: torch.Tensor, weight: torch.Tensor, ov: SPOverlap):
    return _LinearScatter.apply(h, weight, ov)
