"""Cartesian process topologies for the worker-based runtime.

``ProcessTopology(axes, dims)`` maps ranks <-> named coordinates and answers the queries the model
workers and the parameter-reallocation planner need: ranks matching a partial coordinate, the lists of
ranks that vary along one axis (communicator groups), axis-filtered sub-topologies. Two fixed layouts
exist: pipe-data-tensor (training; tensor ranks adjacent so TP stays on neighbouring xGMI links) and
data-pipe-tensor (generation). Capabilities of ``realhf/base/topology.py:86-369``; the implementation is
index arithmetic over strides rather than an enumerated coordinate table.
"""
import itertools
from typing import Dict, Iterable, List, NamedTuple, Optional, Sequence, Tuple

import torch.distributed as dist

_GROUP_CACHE: Dict[Tuple[int, ...], "dist.ProcessGroup"] = {}


def new_or_get_group(ranks: Sequence[int], backend: Optional[str] = None):
    """`dist.new_group` memoised on the sorted rank tuple (every rank must call it in the same order)."""
    key = tuple(sorted(ranks))
    if key not in _GROUP_CACHE:
        _GROUP_CACHE[key] = dist.new_group(list(key), backend=backend)
    return _GROUP_CACHE[key]


def destroy_all_comm_groups():
    for g in _GROUP_CACHE.values():
        try:
            dist.destroy_process_group(g)
        except Exception:
            pass
    _GROUP_CACHE.clear()


def decompose_to_three_factors(n: int) -> List[Tuple[int, int, int]]:
    """All ordered (a, b, c) with a*b*c == n."""
    out = []
    for a in range(1, n + 1):
        if n % a:
            continue
        for b in range(1, n // a + 1):
            if (n // a) % b == 0:
                out.append((a, b, n // a // b))
    return out


class PipeDataTensorProcessCoord(NamedTuple):
    pipe: int
    data: int
    tensor: int


class DataPipeTensorProcessCoord(NamedTuple):
    data: int
    pipe: int
    tensor: int


class ProcessTopology:
    def __init__(self, axes: Sequence[str], dims: Sequence[int]):
        assert len(axes) == len(dims) and all(d >= 1 for d in dims)
        self.axes, self.dims = list(axes), list(dims)
        self.strides = [1] * len(dims)
        for i in range(len(dims) - 2, -1, -1):
            self.strides[i] = self.strides[i + 1] * self.dims[i + 1]
        self._coord_cls = NamedTuple("ProcessCoord", [(a, int) for a in self.axes])

    def world_size(self) -> int:
        n = 1
        for d in self.dims:
            n *= d
        return n

    def get_dim(self, axis: str) -> int:
        return self.dims[self.axes.index(axis)] if axis in self.axes else 0

    def get_rank(self, **coord) -> int:
        assert set(coord) == set(self.axes), f"need every axis of {self.axes}"
        r = 0
        for a, s, d in zip(self.axes, self.strides, self.dims):
            assert 0 <= coord[a] < d, f"{a}={coord[a]} out of range {d}"
            r += coord[a] * s
        return r

    def get_coord(self, rank: int):
        assert 0 <= rank < self.world_size()
        return self._coord_cls(*[(rank // s) % d for s, d in zip(self.strides, self.dims)])

    def get_axis_names(self) -> List[str]:
        return list(self.axes)

    def filter_match(self, **partial) -> List[int]:
        """Ranks whose coordinates equal every given axis value."""
        free = [a for a in self.axes if a not in partial]
        out = []
        for vals in itertools.product(*[range(self.get_dim(a)) for a in free]):
            out.append(self.get_rank(**dict(partial, **dict(zip(free, vals)))))
        return sorted(out)

    def get_axis_list(self, axis: str, idx: int) -> List[int]:
        return self.filter_match(**{axis: idx})

    def get_axis_comm_lists(self, axis: str) -> List[List[int]]:
        """Groups of ranks that differ only along `axis` (one communicator each)."""
        if axis not in self.axes:
            return []
        others = [a for a in self.axes if a != axis]
        lists = []
        for vals in itertools.product(*[range(self.get_dim(a)) for a in others]):
            fixed = dict(zip(others, vals))
            lists.append([self.get_rank(**dict(fixed, **{axis: i})) for i in range(self.get_dim(axis))])
        return lists

    def get_rank_repr(self, rank: int, omit_axes: Iterable[str] = ("data", "pipe"), inner_sep="_", outer_sep="-"):
        c = self.get_coord(rank)._asdict()
        return outer_sep.join(f"{a}{inner_sep}{c[a]:02d}" for a in self.axes if a not in set(omit_axes))

    def __eq__(self, other):
        return isinstance(other, ProcessTopology) and self.axes == other.axes and self.dims == other.dims

    def __hash__(self):
        return hash((tuple(self.axes), tuple(self.dims)))

    def __repr__(self):
        return "x".join(f"{a}{d}" for a, d in zip(self.axes, self.dims))


class PipeDataTensorParallelTopology(ProcessTopology):
    """pipe-major layout used for training."""

    def __init__(self, num_pp: int, num_tp: int, num_dp: int, sequence_parallel: bool = False,
                 gradient_checkpointing: bool = False, gradient_accumulation_fusion: bool = False,
                 max_prompt_len: Optional[int] = None):
        super().__init__(["pipe", "data", "tensor"], [num_pp, num_dp, num_tp])
        self.sequence_parallel = sequence_parallel
        self.gradient_checkpointing = gradient_checkpointing
        self.gradient_accumulation_fusion = gradient_accumulation_fusion
        self.max_prompt_len = max_prompt_len


class DataPipeTensorParallelTopology(ProcessTopology):
    """data-major layout used by generation servers (one replica = contiguous ranks)."""

    def __init__(self, num_pp: int, num_tp: int, num_dp: int, sequence_parallel: bool = False,
                 max_prompt_len: Optional[int] = None):
        super().__init__(["data", "pipe", "tensor"], [num_dp, num_pp, num_tp])
        self.sequence_parallel = sequence_parallel
        self.max_prompt_len = max_prompt_len


class ParallelGrid:
    """Communicator groups of one model's topology embedded in the global world via `rank_mapping`
    (local topology rank -> global rank). With `build=False` only the rank lists are computed (planning
    on the master, tests without torch.distributed)."""

    def __init__(self, topology: ProcessTopology, rank_mapping: Optional[Dict[int, int]] = None, build: bool = True,
                 backend: Optional[str] = None):
        self.topo = topology
        n = topology.world_size()
        self.rank_mapping = rank_mapping or {i: i for i in range(n)}
        self.global_rank = dist.get_rank() if (build and dist.is_initialized()) else -1
        inv = {g: l for l, g in self.rank_mapping.items()}
        self.local_rank = inv.get(self.global_rank, -1)
        self.group_ranks: Dict[str, List[List[int]]] = {}
        self.groups: Dict[str, Optional[object]] = {}
        for axis in topology.get_axis_names():
            lists = [[self.rank_mapping[r] for r in l] for l in topology.get_axis_comm_lists(axis)]
            self.group_ranks[axis] = lists
            self.groups[axis] = None
            if build and dist.is_initialized():
                for l in lists:
                    g = new_or_get_group(l, backend)
                    if self.global_rank in l:
                        self.groups[axis] = g

    def coord(self):
        return self.topo.get_coord(self.local_rank) if self.local_rank >= 0 else None

    def get_data_parallel_group(self):
        return self.groups.get("data")

    def get_tensor_model_parallel_group(self):
        return self.groups.get("tensor")

    def get_pipe_parallel_group(self):
        return self.groups.get("pipe")

    def is_first_stage(self) -> bool:
        return self.coord().pipe == 0

    def is_last_stage(self) -> bool:
        return self.coord().pipe == self.topo.get_dim("pipe") - 1


class FakeGrid:
    """Grid stand-in for planning code that only needs sizes and one rank's coordinates."""

    def __init__(self, rank: int, topo: ProcessTopology):
        self.rank, self.topo = rank, topo
        self._c = topo.get_coord(rank)

    def get_data_parallel_rank(self):
        return self._c.data

    def get_tensor_model_parallel_rank(self):
        return self._c.tensor

    def get_pipe_parallel_rank(self):
        return self._c.pipe

    def get_data_parallel_world_size(self):
        return self.topo.get_dim("data")

    def get_tensor_model_parallel_world_size(self):
        return self.topo.get_dim("tensor")

    def get_pipe_parallel_world_size(self):
        return self.topo.get_dim("pipe")
