"""Ulysses sequence (context) parallelism for packed sequences.

Every rank of the ``cp`` group holds a contiguous 1/cp slice of the packed token stream for all
position-wise work (norms, projections, MLP, LM head). Attention needs whole sequences, so one
all-to-all swaps the partition from tokens to heads ([T/cp, H] -> [T, H/cp]), the flash kernel runs on
full sequences with a head subset, and a second all-to-all swaps back. The reference does the same
around HF attention modules (``areal/utils/ulysses.py:1-318``, ``areal/models/transformers/ulyssess_patch.py:
19-140``); here q, k and v travel in ONE all-to-all per direction (they are packed per destination
rank), which matters on xGMI where every collective launch costs a full ring step.

When there are fewer KV heads than cp ranks the KV heads are repeated up to cp (their gradient sums
back through the repeat).
"""
from typing import Optional

import torch
import torch.distributed as dist

from .. import ops
from . import comm


class _SeqToHead(torch.autograd.Function):
    """[T/cp, cp, h, D] (dest-major head groups) -> [T, h, D]."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return _seq_to_head(x, group)

    @staticmethod
    def backward(ctx, g):
        return _head_to_seq(g.contiguous(), ctx.group), None


class _HeadToSeq(torch.autograd.Function):
    """[T, h, D] -> [T/cp, cp, h, D]."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return _head_to_seq(x, group)

    @staticmethod
    def backward(ctx, g):
        return _seq_to_head(g, ctx.group), None


def _peer_for(x, group):
    """The group's peer-memory exchange when it can carry `x` (rows of whole 16-byte vectors), else None."""
    pc = comm.peer(group)
    if pc is None or not x.is_cuda or (x.shape[-1] * x.shape[-2] * x.element_size()) % 16 != 0:
        return None
    return pc


def _seq_to_head(x, group):
    tl, cp, h, D = x.shape
    pc = _peer_for(x, group)
    if pc is not None:  # the transpose is the kernel's addressing
        return pc.seq_to_head(x.contiguous())
    send = x.permute(1, 0, 2, 3).contiguous()
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    return recv.view(cp * tl, h, D)


def _head_to_seq(x, group):
    cp = dist.get_world_size(group)
    T, h, D = x.shape
    pc = _peer_for(x, group)
    if pc is not None:
        return pc.head_to_seq(x.contiguous())
    send = x.contiguous().view(cp, T // cp, h, D)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    return recv.permute(1, 0, 2, 3).contiguous()


class _GatherTokens(torch.autograd.Function):
    """All-gather per-token results of a loss that every rank evaluates identically: the backward of
    rank r is its own slice of the (replicated) upstream gradient."""

    @staticmethod
    def forward(ctx, x, group, rank):
        ctx.rank, ctx.n = rank, x.shape[0]
        return comm.all_gather_cat(x, group)

    @staticmethod
    def backward(ctx, g):
        return g[ctx.rank * ctx.n:(ctx.rank + 1) * ctx.n].contiguous(), None, None


class UlyssesContext:
    def __init__(self, group, size: int, rank: int):
        self.group, self.size, self.rank = group, size, rank

    # ------------------------------------------------------------------ token partition
    def local_slice(self, t: torch.Tensor) -> torch.Tensor:
        """This rank's contiguous 1/cp slice of a packed [T, ...] tensor (T divisible by cp)."""
        n = t.shape[0]
        assert n % self.size == 0, f"packed length {n} must be padded to a multiple of cp={self.size}"
        c = n // self.size
        return t[self.rank * c:(self.rank + 1) * c]

    def gather_tokens(self, t: torch.Tensor) -> torch.Tensor:
        return _GatherTokens.apply(t, self.group, self.rank)

    # ----------------------------------------------------------------------- attention
    def attention(self, qkv: torch.Tensor, hq: int, hkv: int, cu_seqlens: torch.Tensor, max_seqlen: int):
        """`qkv` [T/cp, hq + 2 hkv, D] with RoPE applied -> attention output [T/cp, hq, D]."""
        cp = self.size
        tl, _, D = qkv.shape
        assert hq % cp == 0, f"q heads ({hq}) must divide by cp ({cp})"
        q, k, v = qkv[:, :hq], qkv[:, hq:hq + hkv], qkv[:, hq + hkv:]
        if hkv % cp != 0:
            assert cp % hkv == 0, f"cp ({cp}) must be a multiple of the kv heads ({hkv})"
            rep = cp // hkv
            k = k.repeat_interleave(rep, dim=1)
            v = v.repeat_interleave(rep, dim=1)
            hkv = cp
        hq_l, hkv_l = hq // cp, hkv // cp
        packed = torch.cat([q.reshape(tl, cp, hq_l, D), k.reshape(tl, cp, hkv_l, D), v.reshape(tl, cp, hkv_l, D)], 2)
        full = _SeqToHead.apply(packed, self.group)  # [T, hq_l + 2 hkv_l, D]
        out = ops.flash_attn_varlen_qkv(full, cu_seqlens, max_seqlen, hq_l, hkv_l)  # [T, hq_l, D]
        back = _HeadToSeq.apply(out, self.group)  # [T/cp, cp, hq_l, D]
        return back.reshape(tl, hq, D)
