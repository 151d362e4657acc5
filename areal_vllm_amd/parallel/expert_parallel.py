"""Expert-parallel token dispatch for MoE layers.

Experts are split over the ``ep`` group (a sub-group of data parallel). Tokens, already sorted by global
expert id, travel to the owner rank with one count exchange + one all-to-all-v, run through the local
experts as contiguous slabs, and return by the mirrored all-to-all-v. The reference's MoE token
dispatcher keeps all experts on every rank (``realhf/impl/model/modules/moe/token_dispatcher.py:
16-194``) and gets expert parallelism only through Megatron (``areal/api/alloc_mode.py:35-214``);
on xGMI one large all-to-all-v per direction per layer is the cheapest shape for this exchange.
"""
from typing import List

import torch
import torch.distributed as dist


class _AllToAllV(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, send_splits: List[int], recv_splits: List[int], group):
        ctx.group, ctx.send_splits, ctx.recv_splits = group, send_splits, recv_splits
        out = x.new_empty((sum(recv_splits),) + tuple(x.shape[1:]))
        dist.all_to_all_single(out, x.contiguous(), output_split_sizes=recv_splits, input_split_sizes=send_splits,
                               group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        out = g.new_empty((sum(ctx.send_splits),) + tuple(g.shape[1:]))
        dist.all_to_all_single(out, g.contiguous(), output_split_sizes=ctx.send_splits,
                               input_split_sizes=ctx.recv_splits, group=ctx.group)
        return out, None, None, None


def ep_dispatch_combine(moe, xs: torch.Tensor, counts: torch.Tensor, ep_group) -> torch.Tensor:
    """`xs` [N, H] sorted by global expert id, `counts` [n_experts] tokens per expert on THIS rank.
    Returns expert outputs in the same order as `xs`."""
    ep = dist.get_world_size(ep_group)
    n_local = moe.n_local
    # exchange per-expert counts: recv_counts[src, e_local]
    send_counts = counts.view(ep, n_local).contiguous()
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts, group=ep_group)
    send_splits = send_counts.sum(1).tolist()
    rc = recv_counts.tolist()
    recv_splits = [sum(r) for r in rc]
    recv = _AllToAllV.apply(xs, send_splits, recv_splits, ep_group)  # grouped by source rank, then expert
    # regroup source-major -> expert-major so each local expert sees one contiguous slab
    offs, o = [], 0
    for s in range(ep):
        row = []
        for e in range(n_local):
            row.append(o)
            o += rc[s][e]
        offs.append(row)
    idx_chunks = []
    for e in range(n_local):
        for s in range(ep):
            if rc[s][e]:
                idx_chunks.append(torch.arange(offs[s][e], offs[s][e] + rc[s][e], device=xs.device))
    if idx_chunks:
        perm = torch.cat(idx_chunks)
        per_expert = [sum(rc[s][e] for s in range(ep)) for e in range(n_local)]
        ys = moe._experts(recv[perm], per_expert)[torch.argsort(perm)]
    else:
        ys = recv.new_zeros(0, xs.shape[1])
    return _AllToAllV.apply(ys, recv_splits, send_splits, ep_group)
