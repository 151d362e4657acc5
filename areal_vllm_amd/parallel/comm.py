"""Collective helpers that work on both RCCL (GPU) and gloo (CPU tests).

gloo lacks reduce_scatter_tensor, so the CPU path emulates it with an all-reduce and a slice; the
numerical result is the same, which is what the gloo fake-cluster tests rely on.
"""
import os
from typing import Optional

import torch
import torch.distributed as dist

# A group of ONE rank normally short-cuts every collective into a local copy. With this switch the data-parallel
# collectives (ZeRO-1 bucket reduce-scatter / all-gather, ZeRO-3 gathers and reductions, norm all-reduces) go
# through the backend anyway: on a single MI355X that runs the real RCCL enqueue path - stream ordering against
# the compute stream, asynchronous work handles, bucket order, graph capture - which the host-backend rehearsals
# of multi-rank layouts cannot show.
FORCE_COLLECTIVES = os.environ.get("AREAL_FORCE_COLLECTIVES", "0") == "1"


# peer-memory collectives (ops/peer_collectives.py) attached to a group for the ACTIVATION collectives the model
# issues on the compute stream; the sharded optimizer holds an object of its own (it works on a side stream)
_PEERS = {}


def attach_peer(group, pc) -> None:
    if pc is None:
        _PEERS.pop(group, None)
    else:
        _PEERS[group] = pc


def peer(group):
    return _PEERS.get(group) if _PEERS else None


# parallel/sp_overlap.SPOverlap of a tensor-parallel group: sequence-parallel linears pipelined with their collective
_OVERLAPS = {}


def attach_overlap(group, ov) -> None:
    if ov is None:
        _OVERLAPS.pop(group, None)
    else:
        _OVERLAPS[group] = ov


def overlap(group):
    return _OVERLAPS.get(group) if _OVERLAPS else None


def group_size(group) -> int:
    if not dist.is_initialized():
        return 1
    return dist.get_world_size(group)


def group_rank(group) -> int:
    if not dist.is_initialized():
        return 0
    return dist.get_rank(group)


def trivial(group) -> bool:
    """True when collectives over `group` may be replaced by local copies."""
    if not dist.is_initialized():
        return True
    return dist.get_world_size(group) == 1 and not FORCE_COLLECTIVES


def _is_nccl(t: torch.Tensor) -> bool:
    return t.is_cuda


def reduce_scatter_sum(out_shard: torch.Tensor, full: torch.Tensor, group) -> None:
    """out_shard = sum over ranks of full[rank_slice]."""
    n = group_size(group)
    if trivial(group):
        if out_shard.data_ptr() != full.data_ptr():
            out_shard.copy_(full[: out_shard.numel()])
        return
    if _is_nccl(full):
        dist.reduce_scatter_tensor(out_shard, full, op=dist.ReduceOp.SUM, group=group)
    else:
        tmp = full.clone()
        dist.all_reduce(tmp, group=group)
        r = group_rank(group)
        out_shard.copy_(tmp[r * out_shard.numel():(r + 1) * out_shard.numel()])


def all_gather_flat(full: torch.Tensor, shard: torch.Tensor, group) -> None:
    """full = concat over ranks of shard (shard may alias full's own slice)."""
    n = group_size(group)
    if trivial(group):
        if shard.data_ptr() != full.data_ptr():
            full[: shard.numel()].copy_(shard)
        return
    if _is_nccl(full):
        dist.all_gather_into_tensor(full, shard, group=group)
    else:
        parts = [torch.empty_like(shard) for _ in range(n)]
        dist.all_gather(parts, shard.clone(), group=group)
        for i, p in enumerate(parts):
            full[i * shard.numel():(i + 1) * shard.numel()].copy_(p)


def all_reduce_sum(t: torch.Tensor, group) -> torch.Tensor:
    if not trivial(group):
        dist.all_reduce(t, group=group)
    return t


def all_reduce_max(t: torch.Tensor, group) -> torch.Tensor:
    if not trivial(group):
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t


def all_gather_cat(x: torch.Tensor, group) -> torch.Tensor:
    """Concatenate `x` from every rank along dim 0."""
    n = group_size(group)
    if n == 1:
        return x
    x = x.contiguous()
    pc = peer(group)
    if pc is not None and pc.serves_rows(x) and x.numel() * x.element_size() % 16 == 0:
        return pc.all_gather_rows(x)
    out = x.new_empty((n * x.shape[0],) + tuple(x.shape[1:]))
    if _is_nccl(x):
        dist.all_gather_into_tensor(out, x, group=group)
    else:
        dist.all_gather(list(out.chunk(n, 0)), x, group=group)
    return out


def reduce_scatter_rows(x: torch.Tensor, group) -> torch.Tensor:
    """Sum `x` over ranks and return this rank's 1/n slice along dim 0."""
    n = group_size(group)
    if n == 1:
        return x
    x = x.contiguous()
    c = x.shape[0] // n
    pc = peer(group)
    if pc is not None and x.shape[0] % n == 0 and pc.serves_rows(x):
        return pc.reduce_scatter_rows(x)
    if _is_nccl(x):
        out = x.new_empty((c,) + tuple(x.shape[1:]))
        dist.reduce_scatter_tensor(out, x, group=group)
        return out
    tmp = x.clone()
    dist.all_reduce(tmp, group=group)
    r = group_rank(group)
    return tmp[r * c:(r + 1) * c].contiguous()
