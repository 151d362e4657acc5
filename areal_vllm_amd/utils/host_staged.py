"""Rehearse multi-rank layouts on fewer GPUs than ranks: collectives of a host-backend (gloo) group on DEVICE tensors
are staged through host memory.

RCCL refuses two ranks on one device, and the host backend's own device path (private streams, worker threads) was
seen to hang two ranks sharing a GPU. With `install()` every `torch.distributed` collective / point-to-point call whose
tensors live on a GPU while its group runs on gloo copies to host memory, runs the host collective and copies back -
slow, synchronous with the current stream, and exactly what is needed to run the tensor- / pipeline- / data-parallel
code paths with the real HIP kernels when only one GPU is at hand (`tests/test_engine_gpu.py`, the tensor-parallel
server's one-GPU rehearsal). Calls on CPU tensors or on RCCL groups pass through untouched. Not a production path.
"""
from typing import List, Optional

import torch
import torch.distributed as dist

_orig = {}


def _staged(group, *tensors) -> bool:
    ts = [t for t in tensors if isinstance(t, torch.Tensor)]
    return bool(ts) and any(t.is_cuda for t in ts) and dist.get_backend(group) == "gloo"


def _host(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to("cpu", copy=True).contiguous()


class _Done:
    def wait(self, *a, **k):
        return True

    def is_completed(self):
        return True


class _PendingSend:
    def __init__(self, work, keep):
        self.work, self.keep = work, keep

    def wait(self, *a, **k):
        return self.work.wait()

    def is_completed(self):
        return self.work.is_completed()


class _PendingRecv:
    def __init__(self, work, host, dev):
        self.work, self.host, self.dev, self.done = work, host, dev, False

    def wait(self, *a, **k):
        if not self.done:
            self.work.wait()
            self.dev.copy_(self.host)
            self.done = True
        return True

    def is_completed(self):
        return self.done or self.work.is_completed()


def _all_reduce(tensor, op=dist.ReduceOp.SUM, group=None, async_op=False):
    if not _staged(group, tensor):
        return _orig["all_reduce"](tensor, op=op, group=group, async_op=async_op)
    h = _host(tensor)
    _orig["all_reduce"](h, op=op, group=group)
    tensor.copy_(h)
    return _Done() if async_op else None


def _reduce(tensor, dst=None, op=dist.ReduceOp.SUM, group=None, async_op=False, group_dst=None):
    if not _staged(group, tensor):
        kw = dict(op=op, group=group, async_op=async_op)
        if group_dst is not None:
            kw["group_dst"] = group_dst
        return _orig["reduce"](tensor, dst, **kw)
    if dst is None:
        dst = dist.get_global_rank(group, group_dst) if group is not None else group_dst
    h = _host(tensor)
    _orig["reduce"](h, dst, op=op, group=group)
    if dist.get_rank() == dst:
        tensor.copy_(h)
    return _Done() if async_op else None


def _broadcast(tensor, src=None, group=None, async_op=False, group_src=None):
    if not _staged(group, tensor):
        kw = dict(group=group, async_op=async_op)
        if group_src is not None:
            kw["group_src"] = group_src
        return _orig["broadcast"](tensor, src, **kw)
    if src is None:
        src = dist.get_global_rank(group, group_src) if group is not None else group_src
    h = _host(tensor)
    _orig["broadcast"](h, src, group=group)
    if dist.get_rank() != src:
        tensor.copy_(h)
    return _Done() if async_op else None


def _all_gather(tensor_list, tensor, group=None, async_op=False):
    if not _staged(group, tensor, *tensor_list):
        return _orig["all_gather"](tensor_list, tensor, group=group, async_op=async_op)
    hosts = [torch.empty(t.shape, dtype=t.dtype) for t in tensor_list]
    _orig["all_gather"](hosts, _host(tensor), group=group)
    for dev, h in zip(tensor_list, hosts):
        dev.copy_(h)
    return _Done() if async_op else None


def _all_gather_into_tensor(output_tensor, input_tensor, group=None, async_op=False):
    if not _staged(group, output_tensor, input_tensor):
        return _orig["all_gather_into_tensor"](output_tensor, input_tensor, group=group, async_op=async_op)
    n = dist.get_world_size(group)
    hosts = [torch.empty(input_tensor.shape, dtype=input_tensor.dtype) for _ in range(n)]
    _orig["all_gather"](hosts, _host(input_tensor), group=group)
    output_tensor.copy_(torch.cat([h.reshape(-1) for h in hosts]).view(output_tensor.shape))
    return _Done() if async_op else None


def _reduce_scatter_tensor(output, input, op=dist.ReduceOp.SUM, group=None, async_op=False):
    if not _staged(group, output, input):
        return _orig["reduce_scatter_tensor"](output, input, op=op, group=group, async_op=async_op)
    h = _host(input)
    _orig["all_reduce"](h, op=op, group=group)
    r, numel = dist.get_rank(group), output.numel()
    output.copy_(h.reshape(-1)[r * numel:(r + 1) * numel].view(output.shape))
    return _Done() if async_op else None


def _all_to_all_single(output, input, output_split_sizes=None, input_split_sizes=None, group=None, async_op=False):
    if not _staged(group, output, input):
        return _orig["all_to_all_single"](output, input, output_split_sizes, input_split_sizes, group=group,
                                          async_op=async_op)
    h_out = torch.empty(output.shape, dtype=output.dtype)
    _orig["all_to_all_single"](h_out, _host(input), output_split_sizes, input_split_sizes, group=group)
    output.copy_(h_out)
    return _Done() if async_op else None


def _send(tensor, dst=None, group=None, tag=0, **kw):
    if not _staged(group, tensor):
        return _orig["send"](tensor, dst, group=group, tag=tag, **kw)
    return _orig["send"](_host(tensor), dst, group=group, tag=tag, **kw)


def _recv(tensor, src=None, group=None, tag=0, **kw):
    if not _staged(group, tensor):
        return _orig["recv"](tensor, src, group=group, tag=tag, **kw)
    h = torch.empty(tensor.shape, dtype=tensor.dtype)
    out = _orig["recv"](h, src, group=group, tag=tag, **kw)
    tensor.copy_(h)
    return out


def _isend(tensor, dst=None, group=None, tag=0, **kw):
    if not _staged(group, tensor):
        return _orig["isend"](tensor, dst, group=group, tag=tag, **kw)
    h = _host(tensor)
    return _PendingSend(_orig["isend"](h, dst, group=group, tag=tag, **kw), h)


def _irecv(tensor, src=None, group=None, tag=0, **kw):
    if not _staged(group, tensor):
        return _orig["irecv"](tensor, src, group=group, tag=tag, **kw)
    h = torch.empty(tensor.shape, dtype=tensor.dtype)
    return _PendingRecv(_orig["irecv"](h, src, group=group, tag=tag, **kw), h, tensor)


class _P2POp:
    """Stand-in for `dist.P2POp` that accepts the staged isend / irecv as its operation."""

    def __init__(self, op, tensor, peer=None, group=None, tag=0, group_peer=None):
        self.op, self.tensor, self.peer, self.group, self.tag = op, tensor, peer, group, tag
        if peer is None and group_peer is not None:
            self.peer = dist.get_global_rank(group, group_peer) if group is not None else group_peer


def _batch_isend_irecv(p2p_op_list: List) -> List:
    works = []
    for o in p2p_op_list:
        fn = _isend if o.op in (_isend, _orig["isend"]) else _irecv
        works.append(fn(o.tensor, o.peer, group=o.group, tag=o.tag))
    return works


_PATCHES = dict(all_reduce=_all_reduce, reduce=_reduce, broadcast=_broadcast, all_gather=_all_gather,
                all_gather_into_tensor=_all_gather_into_tensor, reduce_scatter_tensor=_reduce_scatter_tensor,
                all_to_all_single=_all_to_all_single, send=_send, recv=_recv, isend=_isend, irecv=_irecv,
                batch_isend_irecv=_batch_isend_irecv, P2POp=_P2POp)


def install() -> None:
    """Idempotent. Call after `torch.distributed` is importable and before the code under rehearsal issues its first
    collective."""
    if _orig:
        return
    for name, fn in _PATCHES.items():
        _orig[name] = getattr(dist, name)
        setattr(dist, name, fn)


def uninstall() -> None:
    for name, fn in _orig.items():
        setattr(dist, name, fn)
    _orig.clear()


def installed() -> bool:
    return bool(_orig)


def maybe_install_from_env(env: Optional[dict] = None) -> bool:
    """`AREAL_HOST_STAGED_COLLECTIVES=1` turns the staging on (launch scripts of one-GPU rehearsals set it)."""
    import os

    if (env or os.environ).get("AREAL_HOST_STAGED_COLLECTIVES", "0") == "1":
        install()
        return True
    return False
