"""Tensor-parallel building blocks over the xGMI ``tp`` group.

The transformer's default TP path is the all-reduce form (identity/all-reduce pairs inside
``models/transformer.py``). This module adds the sequence-parallel form - activations between blocks are
sharded along tokens, the column-parallel GEMM is preceded by an all-gather and the row-parallel GEMM is
followed by a reduce-scatter - plus the vocab-parallel embedding and helpers to shard / merge
checkpoint tensors. Semantics follow the reference's Megatron-style modules (``realhf/impl/model/
parallelism/tensor_parallel/mappings.py:12-304``, ``.../modules.py:57-402, 726-1215``); the shapes are
chosen for xGMI rings: one collective per GEMM pair on the full [T, H] activation, never per head.
"""
from typing import List, Optional

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import comm


def _size(group) -> int:
    return dist.get_world_size(group) if group is not None and dist.is_initialized() else 1


class CopyToTP(torch.autograd.Function):
    """Identity forward, all-reduce backward (input of a column-parallel GEMM)."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return x

    @staticmethod
    def backward(ctx, g):
        if _size(ctx.group) > 1:
            g = g.contiguous()
            dist.all_reduce(g, group=ctx.group)
        return g, None


class ReduceFromTP(torch.autograd.Function):
    """All-reduce forward, identity backward (output of a row-parallel GEMM)."""

    @staticmethod
    def forward(ctx, x, group):
        if _size(group) > 1:
            x = x.contiguous()
            dist.all_reduce(x, group=group)
        return x

    @staticmethod
    def backward(ctx, g):
        return g, None


class GatherFromSP(torch.autograd.Function):
    """Token-sharded [T/tp, H] -> [T, H]; backward reduce-scatters the gradient."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        n = _size(group)
        if n == 1:
            return x
        return comm.all_gather_cat(x, group)

    @staticmethod
    def backward(ctx, g):
        n = _size(ctx.group)
        if n == 1:
            return g, None
        return comm.reduce_scatter_rows(g, ctx.group), None


class GatherForReplicatedUse(torch.autograd.Function):
    """Token-sharded [T/tp, H] -> [T, H] for a consumer that every rank evaluates identically (replicated
    LM head and loss): the upstream gradient is the same on all ranks, so the backward keeps this rank's
    slice instead of summing tp copies."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        n = _size(group)
        if n == 1:
            return x
        ctx.rank, ctx.rows = dist.get_rank(group), x.shape[0]
        return comm.all_gather_cat(x, group)

    @staticmethod
    def backward(ctx, g):
        if _size(ctx.group) == 1:
            return g, None
        return g[ctx.rank * ctx.rows:(ctx.rank + 1) * ctx.rows].contiguous(), None


class ReduceScatterToSP(torch.autograd.Function):
    """Partial sums [T, H] -> token-sharded reduced [T/tp, H]; backward all-gathers."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        n = _size(group)
        if n == 1:
            return x
        return comm.reduce_scatter_rows(x, group)

    @staticmethod
    def backward(ctx, g):
        n = _size(ctx.group)
        if n == 1:
            return g, None
        return comm.all_gather_cat(g, ctx.group), None


class ScatterToSP(torch.autograd.Function):
    """Replicated [T, H] -> this rank's token slice; backward all-gathers."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        n = _size(group)
        if n == 1:
            return x
        r = dist.get_rank(group)
        c = x.shape[0] // n
        return x[r * c:(r + 1) * c].contiguous()

    @staticmethod
    def backward(ctx, g):
        n = _size(ctx.group)
        if n == 1:
            return g, None
        return comm.all_gather_cat(g, ctx.group), None


class GatherVocab(torch.autograd.Function):
    """Vocabulary-sharded logits [T, V_r] -> [T, V]: shards may differ in width (the last one takes what an
    even split leaves), so they travel padded to the widest. Backward keeps this rank's columns."""

    @staticmethod
    def forward(ctx, x, group, vocab):
        n = _size(group)
        if n == 1:
            return x
        from ..ops import vocab_shard_range

        bounds = [vocab_shard_range(vocab, n, i) for i in range(n)]
        widths = [hi - lo for lo, hi in bounds]
        assert x.shape[1] == widths[dist.get_rank(group)]
        wmax = max(widths)
        padded = x if x.shape[1] == wmax else F.pad(x, (0, wmax - x.shape[1]))
        parts = comm.all_gather_cat(padded.contiguous().unsqueeze(0), group)
        ctx.lo, ctx.hi = bounds[dist.get_rank(group)]
        return torch.cat([parts[i, :, :w] for i, w in enumerate(widths)], 1)

    @staticmethod
    def backward(ctx, g):
        if not hasattr(ctx, "lo"):
            return g, None, None
        return g[:, ctx.lo:ctx.hi].contiguous(), None, None


def column_parallel_linear(x, weight, bias=None, group=None, sequence_parallel: bool = False):
    """y_shard = x @ W_shard^T. `x` is replicated (or token-sharded when sequence_parallel)."""
    x = GatherFromSP.apply(x, group) if sequence_parallel else CopyToTP.apply(x, group)
    return F.linear(x, weight, bias)


def row_parallel_linear(x_shard, weight, bias=None, group=None, sequence_parallel: bool = False):
    """y = sum_r x_r @ W_r^T. Output replicated (or token-sharded when sequence_parallel)."""
    y = F.linear(x_shard, weight)
    y = ReduceScatterToSP.apply(y, group) if sequence_parallel else ReduceFromTP.apply(y, group)
    return y if bias is None else y + bias


def vocab_parallel_embedding(ids: torch.Tensor, weight_shard: torch.Tensor, group=None,
                             vocab_start: Optional[int] = None) -> torch.Tensor:
    """Embedding with the vocabulary split over the tp group: out-of-shard ids contribute zero and the
    partial results are summed (reference ``modules.py:57-139``). `vocab_start`: first row of this shard
    (default: equal shards)."""
    n = _size(group)
    if n == 1:
        return F.embedding(ids, weight_shard)
    r = dist.get_rank(group)
    per = weight_shard.shape[0]
    lo = r * per if vocab_start is None else vocab_start
    mask = (ids < lo) | (ids >= lo + per)
    local = (ids - lo).masked_fill(mask, 0)
    out = F.embedding(local, weight_shard)
    out = out.masked_fill(mask.unsqueeze(-1), 0.0)
    return ReduceFromTP.apply(out, group)


# ------------------------------------------------------------------------------ sharding helpers
def shard_rows(t: torch.Tensor, tp: int, rank: int) -> torch.Tensor:
    assert t.shape[0] % tp == 0
    c = t.shape[0] // tp
    return t[rank * c:(rank + 1) * c]


def shard_cols(t: torch.Tensor, tp: int, rank: int) -> torch.Tensor:
    assert t.shape[1] % tp == 0
    c = t.shape[1] // tp
    return t[:, rank * c:(rank + 1) * c]


def merge_rows(shards: List[torch.Tensor]) -> torch.Tensor:
    return torch.cat(shards, 0)


def merge_cols(shards: List[torch.Tensor]) -> torch.Tensor:
    return torch.cat(shards, 1)
