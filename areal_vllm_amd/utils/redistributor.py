"""Balance a padded batch across a process group by token count.

Every rank contributes its local rows; rows are kept in consecutive segments of `granularity` (GRPO groups
stay together), segments are dealt to ranks by first-fit-decreasing on their token totals, and each rank
gets its share re-padded to its own longest row. Contract of ``areal/utils/redistributor.py:11-60``.
"""
import dataclasses
from typing import Any, Dict, List

import torch
import torch.distributed as dist

from . import datapack
from .data import all_gather_tensor_container, concat_padded_tensors


@dataclasses.dataclass
class RedistributedData:
    all_data: List[Dict[str, Any]]
    data: Dict[str, Any]
    rank: int
    group_indices: List[List[int]]


def _slice_rows(d: Dict[str, Any], lo: int, hi: int) -> Dict[str, Any]:
    B = d["attention_mask"].shape[0]
    out = {}
    for k, v in d.items():
        if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == B:
            out[k] = v[lo:hi]
        elif isinstance(v, list) and len(v) == B:
            out[k] = v[lo:hi]
        else:
            out[k] = v
    return out


def redistribute(data: Dict[str, Any], granularity: int = 1, group=None) -> RedistributedData:
    gathered = all_gather_tensor_container(data, group=group)
    segments: List[Dict[str, Any]] = []
    for d in gathered:
        bs = d["attention_mask"].shape[0]
        assert bs % granularity == 0, f"batch size {bs} is not a multiple of the granularity {granularity}"
        for i in range(0, bs, granularity):
            seg = _slice_rows(d, i, i + granularity)
            S = seg["attention_mask"].shape[1]
            longest = int(seg["attention_mask"].sum(-1).max())
            seg = {k: (v[:, :longest] if torch.is_tensor(v) and v.dim() >= 2 and v.shape[1] == S else v)
                   for k, v in seg.items()}
            segments.append(seg)
    tokens = [int(s["attention_mask"].sum()) for s in segments]
    n = dist.get_world_size(group) if dist.is_initialized() else 1
    r = dist.get_rank(group) if dist.is_initialized() else 0
    groups = datapack.ffd_allocate(tokens, int(1e12), n)
    mine = concat_padded_tensors([segments[i] for i in groups[r]])
    return RedistributedData(all_data=segments, data=mine, rank=r, group_indices=groups)
