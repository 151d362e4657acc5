"""Batch containers: padded <-> packed conversion, token-budget micro-batching, container collectives.

A *padded batch* is a plain ``dict`` of tensors shaped ``[B, S, ...]`` (per-token) or ``[B, ...]``
(per-sequence) with an ``attention_mask`` of shape ``[B, S]``. A *packed batch* holds the valid tokens
concatenated along dim 0 plus ``cu_seqlens`` (int32 ``[B+1]``) and ``max_seqlen``.

Behavioural counterpart of ``areal/utils/data.py`` in the reference (concat/pad :120-192, FFD
micro-batches :219-242 and :401-498, packing :245-294, 256-token page padding :501-733, container
broadcast/all-gather :815-985), re-implemented on plain dicts (tensordict is not a dependency) with one
packed byte-buffer collective per container instead of per-tensor calls.
"""
import dataclasses
import io
import pickle
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from ..api.cli_args import MicroBatchSpec
from . import datapack

N_TOKENS_PER_PAGE = 256
DEFAULT_MAX_TOKENS_PER_MB = int(1e12)

Batch = Dict[str, Any]


def batch_size_of(data: Batch) -> int:
    if "attention_mask" in data:
        return data["attention_mask"].shape[0]
    if "cu_seqlens" in data:
        return data["cu_seqlens"].numel() - 1
    for v in data.values():
        if torch.is_tensor(v):
            return v.shape[0]
    raise ValueError("cannot infer batch size")


def reorder_list(xs: List, indices: List[int]) -> List:
    return [xs[i] for i in indices]


def dict_map(x: Dict, fn: Callable) -> Dict:
    return {k: fn(v) for k, v in x.items()}


def list_of_dict2dict_of_list(xs: List[Dict]) -> Dict[str, List]:
    keys = xs[0].keys()
    assert all(set(x.keys()) == set(keys) for x in xs)
    return {k: [x[k] for x in xs] for k in keys}


def dict_of_list2list_of_dict(x: Dict[str, List]) -> List[Dict]:
    keys = list(x.keys())
    n = len(x[keys[0]])
    assert all(len(x[k]) == n for k in keys)
    return [{k: x[k][i] for k in keys} for i in range(n)]


def to_device(data: Batch, device) -> Batch:
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in data.items()}


def pad_sequences_to_tensors(seqs: List[Dict[str, Any]], pad_value: float = 0.0) -> Batch:
    """List of per-sequence dicts (1-D tensors / lists / scalars) -> padded batch with attention_mask."""
    if not seqs:
        return {}
    skip = {"multi_modal_input"}
    max_len = max(len(v) for s in seqs for k, v in s.items() if k not in skip and hasattr(v, "__len__")
                  and not isinstance(v, str))
    out: Batch = {}
    for k in seqs[0]:
        if k in skip:
            out[k] = [s[k] for s in seqs]
            continue
        vals = [torch.as_tensor(s[k]) for s in seqs]
        if vals[0].dim() == 0:
            out[k] = torch.stack(vals)
            continue
        out[k] = torch.stack([
            torch.nn.functional.pad(v, (0, max_len - v.shape[0]), value=pad_value) for v in vals
        ])
    lens = [len(next(v for k, v in s.items() if k not in skip and hasattr(v, "__len__"))) for s in seqs]
    out["attention_mask"] = torch.stack([
        torch.cat([torch.ones(n, dtype=torch.bool), torch.zeros(max_len - n, dtype=torch.bool)]) for n in lens
    ])
    return out


def concat_padded_tensors(dicts: Sequence[Batch], pad_value: float = 0.0) -> Batch:
    """Concatenate padded batches along dim 0, right-padding every [B, S] tensor to the longest S."""
    dicts = [d for d in dicts if d]
    if not dicts:
        return {}
    assert all("attention_mask" in d for d in dicts)
    max_len = max(d["attention_mask"].shape[1] for d in dicts)
    out: Batch = {}
    for k in dicts[0].keys():
        vals = []
        for d in dicts:
            v = d[k]
            if not torch.is_tensor(v):
                vals.append(v)
                continue
            S = d["attention_mask"].shape[1]
            if v.dim() >= 2 and v.shape[1] == S and S != max_len:
                pad = [0, 0] * (v.dim() - 2) + [0, max_len - S]
                pv = 0 if k == "attention_mask" else pad_value
                v = torch.nn.functional.pad(v, pad, value=pv)
            vals.append(v)
        if torch.is_tensor(vals[0]):
            out[k] = torch.cat(vals, 0)
        elif isinstance(vals[0], list):
            out[k] = [x for v in vals for x in v]
        else:
            out[k] = vals
    return out


def unpad_input(hidden: torch.Tensor, attention_mask: torch.Tensor):
    seqlens = attention_mask.sum(-1, dtype=torch.int32)
    indices = torch.nonzero(attention_mask.flatten(), as_tuple=False).flatten()
    cu = torch.nn.functional.pad(torch.cumsum(seqlens, 0, dtype=torch.int32), (1, 0))
    flat = hidden.reshape(-1, *hidden.shape[2:])
    return flat[indices], indices, cu, int(seqlens.max()) if seqlens.numel() else 0


def pad_input(packed: torch.Tensor, indices: torch.Tensor, batch: int, seqlen: int) -> torch.Tensor:
    out = torch.zeros(batch * seqlen, *packed.shape[1:], dtype=packed.dtype, device=packed.device)
    out[indices] = packed
    return out.view(batch, seqlen, *packed.shape[1:])


def amend_position_ids(data: Batch) -> Batch:
    """Add `position_ids` ([B, S], counting valid tokens from 0) if missing."""
    if "position_ids" not in data:
        am = data["attention_mask"].long()
        data["position_ids"] = ((am.cumsum(-1) - 1) * am).to(torch.int64)
    return data


def pack_tensor_dict(data: Batch) -> Batch:
    """Padded batch -> packed batch (valid tokens only)."""
    am = data["attention_mask"].bool()
    B, S = am.shape
    lens = am.sum(-1, dtype=torch.int32)
    cu = torch.nn.functional.pad(torch.cumsum(lens, 0, dtype=torch.int32), (1, 0))
    out: Batch = {}
    for k, v in data.items():
        if k == "attention_mask":
            continue
        if torch.is_tensor(v) and v.dim() >= 2 and v.shape[0] == B and v.shape[1] == S:
            out[k] = v[am]
        else:
            out[k] = v
    out["cu_seqlens"] = cu
    out["max_seqlen"] = int(lens.max()) if B else 0
    return out


def unpack_sequence(x: torch.Tensor, cu_seqlens: Optional[torch.Tensor] = None, lens: Optional[List[int]] = None):
    if lens is None:
        lens = (cu_seqlens[1:] - cu_seqlens[:-1]).tolist()
    return list(torch.split(x, lens, dim=0))


def allocate_balanced_mbs(mb_spec: MicroBatchSpec, lens: List[int]) -> List[List[int]]:
    groups = datapack.ffd_allocate(lens, mb_spec.max_tokens_per_mb, min_groups=mb_spec.n_mbs or 1)
    return sorted(sorted(g) for g in groups)


def allocate_balanced_mbs_synced(mb_spec: MicroBatchSpec, lens: List[int], group=None) -> List[List[int]]:
    """Same as `allocate_balanced_mbs` but every rank of `group` ends up with the same micro-batch count
    (one MAX all-reduce of an int instead of the reference's all_gather_object, data.py:219-242)."""
    groups = allocate_balanced_mbs(mb_spec, lens)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return groups
    dev = _group_device(group)
    for _ in range(8):
        # one MAX all-reduce of (count, -count) gives the group-wide maximum and minimum; the loop ends on
        # a condition every rank evaluates identically, so all ranks issue the same number of collectives
        n = torch.tensor([len(groups), -len(groups)], dtype=torch.int64, device=dev)
        dist.all_reduce(n, op=dist.ReduceOp.MAX, group=group)
        hi, lo = int(n[0].item()), -int(n[1].item())
        if hi == lo:
            return groups
        if len(groups) != hi:
            groups = allocate_balanced_mbs(MicroBatchSpec.new(mb_spec, n_mbs=hi), lens)
    raise RuntimeError("data-parallel ranks could not agree on a micro-batch count")


def _group_device(group=None):
    """Device for small host-side bookkeeping collectives: the CPU whenever the group has a gloo side
    (the default "cpu:gloo,cuda:nccl" groups do), so reading the result does not synchronise the GPU."""
    backend = str(dist.get_backend(group))
    if "gloo" in backend or "nccl" not in backend:
        return torch.device("cpu")
    return torch.device("cuda", torch.cuda.current_device())


@dataclasses.dataclass
class MicroBatchList:
    data: Batch
    mb_spec: MicroBatchSpec
    mbs: List[Batch]
    forward_indices: List[int]
    backward_indices: List[int]
    group_lens: List[int]
    padded_mbs: Optional[List[Batch]] = None
    padding_lengths: Optional[List[int]] = None
    padded_to_lengths: Optional[List[int]] = None
    align_to_lengths: Optional[List[int]] = None
    old_cu_seqlens_list: Optional[List[torch.Tensor]] = None

    def to(self, device):
        f = lambda d: to_device(d, device)  # noqa: E731
        return dataclasses.replace(
            self, data=f(self.data), mbs=[f(m) for m in self.mbs],
            padded_mbs=[f(m) for m in self.padded_mbs] if self.padded_mbs is not None else None,
            old_cu_seqlens_list=[t.to(device) for t in self.old_cu_seqlens_list]
            if self.old_cu_seqlens_list is not None else None,
        )


def split_padded_tensor_dict_into_mb_list(data: Batch, mb_spec: MicroBatchSpec, group=None) -> MicroBatchList:
    """Split a padded batch into token-budgeted micro-batches (FFD, synchronised count across `group`)."""
    am = data["attention_mask"]
    B, S = am.shape
    lens = am.sum(-1).long().tolist()
    spec = mb_spec
    if spec.max_tokens_per_mb is None:
        spec = MicroBatchSpec.new(mb_spec, max_tokens_per_mb=DEFAULT_MAX_TOKENS_PER_MB)
    n_min = min(spec.n_mbs or 1, B)
    spec = MicroBatchSpec.new(spec, n_mbs=n_min)
    groups = allocate_balanced_mbs_synced(spec, lens, group=group)
    forward_indices = datapack.flat2d(groups)
    backward_indices = np.zeros(B, dtype=np.int64)
    backward_indices[forward_indices] = np.arange(B)
    mbs = []
    for g in groups:
        idx = torch.as_tensor(g, dtype=torch.long, device=am.device)
        mb = {}
        for k, v in data.items():
            if torch.is_tensor(v) and v.shape[0] == B:
                mb[k] = v.index_select(0, idx)
            elif isinstance(v, list) and len(v) == B:
                mb[k] = [v[i] for i in g]
            else:
                mb[k] = v
        # trim right padding of the micro-batch
        ml = int(mb["attention_mask"].sum(-1).max())
        for k, v in list(mb.items()):
            if torch.is_tensor(v) and v.dim() >= 2 and v.shape[1] == S:
                mb[k] = v[:, :ml]
        mbs.append(mb)
    return MicroBatchList(data=data, mb_spec=mb_spec, mbs=mbs, forward_indices=forward_indices,
                          backward_indices=backward_indices.tolist(),
                          group_lens=[sum(lens[i] for i in g) for g in groups])


def pad_packed_tensor_dict(data: Batch, pad_to_length: int, pad_value: float = 0.0) -> Tuple[Batch, int]:
    """Append one dummy sequence so that the packed token count equals `pad_to_length`."""
    cu = data["cu_seqlens"]
    total = int(cu[-1])
    pad = pad_to_length - total
    assert pad >= 0
    if pad == 0:
        return data, 0
    out: Batch = {}
    for k, v in data.items():
        if k == "cu_seqlens":
            out[k] = torch.cat([cu, torch.tensor([pad_to_length], dtype=cu.dtype, device=cu.device)])
        elif k == "max_seqlen":
            out[k] = max(int(v), pad)
        elif k == "position_ids":
            out[k] = torch.cat([v, torch.arange(pad, dtype=v.dtype, device=v.device)])
        elif torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == total:
            filler = torch.full((pad, *v.shape[1:]), pad_value, dtype=v.dtype, device=v.device)
            out[k] = torch.cat([v, filler])
        else:
            out[k] = v
    return out, pad


def pad_mb_list(mb_list: MicroBatchList, pad_value: float = 0.0, pad_to_maximum: bool = False,
                align_to: int = 1) -> MicroBatchList:
    """Pack every micro-batch and pad its token count to a multiple of lcm(256, align_to)
    (or to `max_tokens_per_mb` when pad_to_maximum)."""
    import math

    unit = math.lcm(N_TOKENS_PER_PAGE, max(1, align_to))
    padded, pads, tos, olds = [], [], [], []
    for mb in mb_list.mbs:
        packed = pack_tensor_dict(amend_position_ids(dict(mb)))
        total = int(packed["cu_seqlens"][-1])
        if pad_to_maximum and mb_list.mb_spec.max_tokens_per_mb:
            to = int(mb_list.mb_spec.max_tokens_per_mb)
        else:
            to = (total + unit - 1) // unit * unit
        olds.append(packed["cu_seqlens"])
        p, n = pad_packed_tensor_dict(packed, to, pad_value)
        padded.append(p)
        pads.append(n)
        tos.append(to)
    return dataclasses.replace(mb_list, padded_mbs=padded, padding_lengths=pads, padded_to_lengths=tos,
                               old_cu_seqlens_list=olds)


def unpad_logits(x: torch.Tensor, padding_length: int) -> torch.Tensor:
    return x[: x.shape[0] - padding_length] if padding_length > 0 else x


def repad_packed(x: torch.Tensor, cu_seqlens: torch.Tensor, seqlen: int) -> torch.Tensor:
    """Packed per-token tensor [T, ...] -> padded [B, seqlen, ...]."""
    lens = (cu_seqlens[1:] - cu_seqlens[:-1]).tolist()
    out = x.new_zeros(len(lens), seqlen, *x.shape[1:])
    o = 0
    for i, n in enumerate(lens):
        out[i, :n] = x[o:o + n]
        o += n
    return out


# ------------------------------------------------------------------------ container collectives
def _serialize(data: Any) -> Tuple[bytes, List[torch.Tensor]]:
    tensors: List[torch.Tensor] = []

    def strip(x):
        if torch.is_tensor(x):
            tensors.append(x)
            return ("__tensor__", len(tensors) - 1, tuple(x.shape), str(x.dtype).replace("torch.", ""))
        if isinstance(x, dict):
            return {k: strip(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return type(x)(strip(v) for v in x)
        return x

    meta = strip(data)
    return pickle.dumps(meta), tensors


def _deserialize(meta_bytes: bytes, flat: torch.Tensor) -> Any:
    meta = pickle.loads(meta_bytes)
    off = [0]

    def build(x):
        if isinstance(x, tuple) and len(x) == 4 and x[0] == "__tensor__":
            dtype = getattr(torch, x[3])
            n = int(np.prod(x[2])) * torch.empty((), dtype=dtype).element_size()
            start = (off[0] + 15) // 16 * 16
            t = flat[start:start + n].view(dtype).view(x[2]).clone()
            off[0] = start + n
            return t
        if isinstance(x, dict):
            return {k: build(v) for k, v in x.items()}
        if isinstance(x, list):
            return [build(v) for v in x]
        if isinstance(x, tuple):
            return tuple(build(v) for v in x)
        return x

    return build(meta)


def _flatten_bytes(tensors: List[torch.Tensor], device) -> torch.Tensor:
    total = 0
    spans = []
    for t in tensors:
        start = (total + 15) // 16 * 16
        n = t.numel() * t.element_size()
        spans.append((start, n))
        total = start + n
    buf = torch.zeros(max(total, 1), dtype=torch.uint8, device=device)
    for t, (s, n) in zip(tensors, spans):
        if n:
            buf[s:s + n] = t.contiguous().view(-1).view(torch.uint8).to(device)
    return buf


def broadcast_tensor_container(data: Any, src_rank: int = 0, group=None) -> Any:
    """Broadcast a nested dict/list of tensors from `src_rank` (global rank) with ONE metadata object
    broadcast and ONE packed byte-buffer broadcast (reference: data.py:930-985 does it per tensor)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return data
    dev = _group_device(group)
    is_src = dist.get_rank() == src_rank
    if is_src:
        meta, tensors = _serialize(data)
        buf = _flatten_bytes(tensors, dev)
        hdr = [meta, buf.numel(), str(tensors[0].device) if tensors else "cpu"]
    else:
        hdr = [None, None, None]
    dist.broadcast_object_list(hdr, src=src_rank, group=group)
    if not is_src:
        buf = torch.empty(hdr[1], dtype=torch.uint8, device=dev)
    dist.broadcast(buf, src=src_rank, group=group)
    if is_src:
        return data
    out = _deserialize(hdr[0], buf)
    return out


def all_gather_tensor_container(data: Any, group=None) -> List[Any]:
    """Gather every rank's container; returns a list ordered by group rank."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [data]
    dev = _group_device(group)
    ws = dist.get_world_size(group)
    meta, tensors = _serialize(data)
    buf = _flatten_bytes(tensors, dev)
    metas = [None] * ws
    dist.all_gather_object(metas, (meta, buf.numel()), group=group)
    mx = max(m[1] for m in metas)
    padded = torch.zeros(mx, dtype=torch.uint8, device=dev)
    padded[: buf.numel()] = buf
    outs = [torch.empty(mx, dtype=torch.uint8, device=dev) for _ in range(ws)]
    dist.all_gather(outs, padded, group=group)
    return [_deserialize(m[0], o[: m[1]]) for m, o in zip(metas, outs)]


def bcast_mb_list(mb_list: Optional[MicroBatchList], src_rank: int = 0, group=None) -> MicroBatchList:
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return mb_list
    payload = dataclasses.asdict(mb_list) if (dist.get_rank() == src_rank and mb_list is not None) else None
    payload = broadcast_tensor_container(payload, src_rank, group)
    if dist.get_rank() == src_rank:
        return mb_list
    payload["mb_spec"] = MicroBatchSpec(**payload["mb_spec"])
    return MicroBatchList(**payload)
