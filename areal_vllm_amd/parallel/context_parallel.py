"""Zig-zag context parallelism for packed sequences with all-gathered K / V.

Role of Megatron's context parallelism in the reference (``areal/experimental/utils/mcore/
packed_context_parallel.py:1-149``: every sequence cut into 2*cp chunks, rank i takes chunks i and
2cp-1-i so that causal attention work is balanced, ring exchange of K/V inside mcore). MI355X-first
differences:

* The *packed token stream* (already padded to a multiple of 2*cp) is cut into 2*cp equal chunks and
  rank i owns chunks i and 2cp-1-i. A long sequence is balanced exactly like Megatron's per-sequence
  zig-zag, many short sequences balance statistically, and no per-sequence padding is needed.
* K and V are all-gathered once per layer instead of travelling a ring: with GQA they are a small
  fraction of the activations (2 of 14 heads for Qwen2-1.5B), and on a fully connected xGMI node an
  all-gather drives every link at once while a ring step is bound by one link.
* Each local query chunk then runs ONE varlen flash call against the key range
  [start of its first sequence, end of the chunk): for the first sequence the queries are the tail of
  their key range (bottom-right aligned causal mask), the other sequences start inside the chunk.
  The backward reduce-scatters dK / dV to the owners of the key chunks.

All position-wise work (norms, projections, MLP, LM head, loss) runs on the local 1/cp of the tokens;
per-token results are gathered back into stream order for the loss functions.
"""
import bisect
from typing import Dict, List, Tuple

import torch
import torch.distributed as dist

from .. import ops
from . import comm


def _chunk_order(cp: int) -> List[int]:
    """Stream chunk held at position j of the rank-major gathered layout (rank r holds r, 2cp-1-r)."""
    out = []
    for r in range(cp):
        out += [r, 2 * cp - 1 - r]
    return out


class _GatherChunks(torch.autograd.Function):
    """[2c, ...] local (chunks r and 2cp-1-r) -> [2cp*c, ...] in stream order. Backward: `reduce` sums the
    upstream gradient over ranks and returns the local chunks (K/V: every rank contributes a gradient);
    otherwise the local chunks of a replicated gradient are picked (per-token loss terms)."""

    @staticmethod
    def forward(ctx, x, group, cp, rank, reduce):
        ctx.group, ctx.cp, ctx.rank, ctx.reduce = group, cp, rank, reduce
        c = x.shape[0] // 2
        full = comm.all_gather_cat(x.contiguous(), group)  # rank-major: [r0a, r0b, r1a, r1b, ...]
        parts = full.view(2 * cp, c, *x.shape[1:])
        inv = torch.empty(2 * cp, dtype=torch.long)
        inv[torch.tensor(_chunk_order(cp))] = torch.arange(2 * cp)
        return parts[inv.to(x.device)].reshape(2 * cp * c, *x.shape[1:])

    @staticmethod
    def backward(ctx, g):
        cp, r = ctx.cp, ctx.rank
        c = g.shape[0] // (2 * cp)
        parts = g.reshape(2 * cp, c, *g.shape[1:])
        if not ctx.reduce:
            return torch.cat([parts[r], parts[2 * cp - 1 - r]], 0), None, None, None, None
        order = torch.tensor(_chunk_order(cp), device=g.device)
        rank_major = parts[order].reshape(2 * cp * c, *g.shape[1:]).contiguous()
        mine = comm.reduce_scatter_rows(rank_major.float(), ctx.group)  # fp32 sum over ranks
        return mine.to(g.dtype), None, None, None, None


class ZigZagContext:
    """Drop-in alternative to ``UlyssesContext`` (same interface towards the model and the engine)."""

    def __init__(self, group, size: int, rank: int):
        self.group, self.size, self.rank = group, size, rank
        self.align = 2 * size
        self._plans: Dict[Tuple[int, int], list] = {}

    # ------------------------------------------------------------------ token partition
    def local_slice(self, t: torch.Tensor) -> torch.Tensor:
        n = t.shape[0]
        assert n % (2 * self.size) == 0, f"packed length {n} must be padded to a multiple of 2*cp={2 * self.size}"
        c = n // (2 * self.size)
        a, b = self.rank, 2 * self.size - 1 - self.rank
        return torch.cat([t[a * c:(a + 1) * c], t[b * c:(b + 1) * c]], 0)

    def gather_tokens(self, t: torch.Tensor) -> torch.Tensor:
        return _GatherChunks.apply(t, self.group, self.size, self.rank, False)

    # ----------------------------------------------------------------------- attention
    def _plan(self, cu_seqlens: torch.Tensor, T: int):
        """Per local chunk: (q_lo, k_lo, k_hi, cu_q, cu_k, max_q, max_k) from the host copy of cu_seqlens."""
        key = (cu_seqlens.data_ptr(), T)
        cached = self._plans.get(key)
        if cached is not None and cached[0] is cu_seqlens:
            return cached[1]
        cu = [int(x) for x in cu_seqlens.tolist()]
        if cu[-1] < T:  # trailing padding behaves like one more sequence
            cu.append(T)
        c = T // (2 * self.size)
        plan = []
        for ci in (self.rank, 2 * self.size - 1 - self.rank):
            a0, a1 = ci * c, (ci + 1) * c
            first = bisect.bisect_right(cu, a0) - 1  # sequence containing token a0
            k_lo = cu[first]
            cq, ck = [0], [0]
            s = first
            while s < len(cu) - 1 and cu[s] < a1:
                q_end = min(cu[s + 1], a1)
                cq.append(q_end - a0)
                ck.append(q_end - k_lo)
                s += 1
            lens_q = [y - x for x, y in zip(cq[:-1], cq[1:])]
            lens_k = [y - x for x, y in zip(ck[:-1], ck[1:])]
            dev = cu_seqlens.device
            plan.append((a0, k_lo, a1, torch.tensor(cq, dtype=torch.int32, device=dev),
                         torch.tensor(ck, dtype=torch.int32, device=dev), max(lens_q), max(lens_k)))
        if len(self._plans) > 64:
            self._plans.clear()
        self._plans[key] = (cu_seqlens, plan)
        return plan

    def attention(self, qkv: torch.Tensor, hq: int, hkv: int, cu_seqlens: torch.Tensor, max_seqlen: int):
        """`qkv` [T/cp, hq + 2 hkv, D] (local chunks, RoPE applied) -> attention output [T/cp, hq, D]."""
        cp = self.size
        tl = qkv.shape[0]
        c = tl // 2
        T = tl * cp
        kv_full = _GatherChunks.apply(qkv[:, hq:], self.group, cp, self.rank, True)  # [T, 2 hkv, D]
        outs = []
        for i, (a0, k_lo, k_hi, cu_q, cu_k, max_q, max_k) in enumerate(self._plan(cu_seqlens, T)):
            q = qkv[i * c:(i + 1) * c, :hq]
            k = kv_full[k_lo:k_hi, :hkv]
            v = kv_full[k_lo:k_hi, hkv:]
            outs.append(ops.flash_attn_varlen_sep(q, k, v, cu_q, cu_k, max_q, max_k))
        return torch.cat(outs, 0)
