"""`SequenceSample`: the unit of data that flows between workers.

A sample batch holds, per key, ONE packed 1-D (or [T, ...]) tensor and, per key, the list of per-item
sequence-length lists (an item may own several sequences under one key, e.g. a group of answers). Ids
identify items across workers. Supports gather / split (by lengths or balanced over data-parallel
ranks) / select / update / JSON transport. Contract of ``realhf/api/core/data_api.py:129-745``
(validation by pydantic there; plain assertions here).
"""
import dataclasses
from typing import Any, Dict, Hashable, List, Optional, Sequence, Set

import numpy as np
import torch

from ..utils import datapack


@dataclasses.dataclass
class SequenceSplitSpec:
    partitions: Optional[List[tuple]] = None
    sizes: Optional[List[int]] = None

    def __post_init__(self):
        if self.partitions is None:
            assert self.sizes is not None
            b = np.concatenate([[0], np.cumsum(self.sizes)]).tolist()
            self.partitions = [(int(b[i]), int(b[i + 1])) for i in range(len(self.sizes))]
        else:
            self.sizes = [e - s for s, e in self.partitions]
        for (s0, e0), (s1, e1) in zip(self.partitions[:-1], self.partitions[1:]):
            assert e0 == s1, "partitions must be contiguous"


@dataclasses.dataclass
class SequenceSample:
    keys: Set[str]
    ids: List[Hashable]
    seqlens: Dict[str, List[List[int]]]
    data: Optional[Dict[str, Optional[torch.Tensor]]] = None
    trailing_shapes: Dict[str, Optional[tuple]] = dataclasses.field(default_factory=dict)
    dtypes: Dict[str, Optional[torch.dtype]] = dataclasses.field(default_factory=dict)
    metadata: Dict[str, List[Any]] = dataclasses.field(default_factory=dict)

    def __post_init__(self):
        self.keys = set(self.keys)
        self.ids = [str(i) for i in self.ids]
        assert len(set(self.ids)) == len(self.ids), f"duplicate ids: {self.ids}"
        assert set(self.seqlens) == self.keys, (set(self.seqlens), self.keys)
        for k in self.keys:
            assert len(self.seqlens[k]) == len(self.ids), f"key {k}: {len(self.seqlens[k])} != {len(self.ids)}"
            if self.data is not None and self.data.get(k) is not None:
                t = self.data[k]
                tot = sum(sum(l) for l in self.seqlens[k])
                assert t.shape[0] == tot, f"key {k}: data has {t.shape[0]} rows, seqlens sum to {tot}"
                self.trailing_shapes.setdefault(k, tuple(t.shape[1:]))
                self.dtypes.setdefault(k, t.dtype)
            else:
                self.trailing_shapes.setdefault(k, None)
                self.dtypes.setdefault(k, None)
        for k, v in self.metadata.items():
            assert len(v) == len(self.ids), f"metadata {k} must have one entry per item"

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_default(cls, ids: List[Hashable], seqlens: List[int], data: Dict[str, torch.Tensor],
                     metadata: Optional[Dict[str, Any]] = None) -> "SequenceSample":
        """One sequence per item; the length of each key is inferred from the key's conventional shape:
        per-sequence scalars (rewards, ...) have length 1, "short" arrays (logprobs, ...) length - 1."""
        sl = {}
        for k, v in data.items():
            if v is None:
                sl[k] = [[int(x)] for x in seqlens]
            elif v.shape[0] == len(seqlens) and v.shape[0] != sum(seqlens):
                sl[k] = [[1] for _ in seqlens]
            elif v.shape[0] == sum(seqlens) - len(seqlens):
                sl[k] = [[int(x) - 1] for x in seqlens]
            else:
                assert v.shape[0] == sum(seqlens), f"cannot infer sequence lengths of key {k}"
                sl[k] = [[int(x)] for x in seqlens]
        return cls(keys=set(data), ids=ids, seqlens=sl, data=data, metadata=metadata or {})

    @classmethod
    def gather(cls, samples: List["SequenceSample"], keys: Optional[Sequence[str]] = None) -> "SequenceSample":
        keys = set(keys) if keys is not None else set(samples[0].keys)
        seqlens = {k: [l for s in samples for l in s.seqlens[k]] for k in keys}
        data = None
        if samples[0].data is not None:
            data = {}
            for k in keys:
                parts = [s.data[k] for s in samples]
                data[k] = None if parts[0] is None else torch.cat(parts, 0)
        meta = {k: [x for s in samples for x in s.metadata.get(k, [None] * s.bs)] for k in samples[0].metadata}
        return cls(keys=keys, ids=[i for s in samples for i in s.ids], seqlens=seqlens, data=data,
                   trailing_shapes={k: samples[0].trailing_shapes.get(k) for k in keys},
                   dtypes={k: samples[0].dtypes.get(k) for k in keys}, metadata=meta)

    # ------------------------------------------------------------------------ views
    @property
    def bs(self) -> int:
        return len(self.ids)

    def _item_tokens(self, key: str) -> List[int]:
        return [sum(l) for l in self.seqlens[key]]

    def _split_key(self) -> str:
        return max(self.keys, key=lambda k: sum(self._item_tokens(k)))

    def split_with_spec(self, spec: SequenceSplitSpec) -> List["SequenceSample"]:
        out = []
        offs = {k: np.concatenate([[0], np.cumsum(self._item_tokens(k))]).astype(int) for k in self.keys}
        for s, e in spec.partitions:
            data = None
            if self.data is not None:
                data = {k: (None if self.data[k] is None else self.data[k][offs[k][s]:offs[k][e]]) for k in self.keys}
            out.append(SequenceSample(
                keys=set(self.keys), ids=self.ids[s:e], seqlens={k: self.seqlens[k][s:e] for k in self.keys},
                data=data, trailing_shapes=dict(self.trailing_shapes), dtypes=dict(self.dtypes),
                metadata={k: v[s:e] for k, v in self.metadata.items()}))
        return out

    def split_with_lengths(self, mb_spec, lens: List[int]):
        """Token-budgeted balanced split (FFD); returns (samples, forward_indices, backward_indices)."""
        groups = datapack.ffd_allocate(lens, int(mb_spec.max_tokens_per_mb or 1 << 40), int(mb_spec.n_mbs or 1))
        groups = sorted([sorted(g) for g in groups])
        fwd = datapack.flat2d(groups)
        bwd = np.zeros(self.bs, dtype=np.int64)
        bwd[fwd] = np.arange(self.bs)
        re = self.reorder(fwd)
        return re.split_with_spec(SequenceSplitSpec(sizes=[len(g) for g in groups])), fwd, bwd.tolist()

    def split(self, mb_spec):
        return self.split_with_lengths(mb_spec, self._item_tokens(self._split_key()))

    def reorder(self, indices: List[int]) -> "SequenceSample":
        parts = self.unpack()
        return SequenceSample.gather([parts[i] for i in indices])

    def unpack(self) -> List["SequenceSample"]:
        return self.split_with_spec(SequenceSplitSpec(sizes=[1] * self.bs))

    def select(self, keys: Sequence[str]) -> "SequenceSample":
        keys = set(keys)
        assert keys <= self.keys, keys - self.keys
        return SequenceSample(keys=keys, ids=list(self.ids), seqlens={k: self.seqlens[k] for k in keys},
                              data=None if self.data is None else {k: self.data[k] for k in keys},
                              trailing_shapes={k: self.trailing_shapes[k] for k in keys},
                              dtypes={k: self.dtypes[k] for k in keys}, metadata=dict(self.metadata))

    def meta(self) -> "SequenceSample":
        return SequenceSample(keys=set(self.keys), ids=list(self.ids), seqlens=dict(self.seqlens), data=None,
                              trailing_shapes=dict(self.trailing_shapes), dtypes=dict(self.dtypes),
                              metadata=dict(self.metadata))

    def update_(self, other: "SequenceSample") -> None:
        assert self.ids == other.ids, "update_ needs the same items in the same order"
        self.keys |= other.keys
        self.seqlens.update(other.seqlens)
        self.trailing_shapes.update(other.trailing_shapes)
        self.dtypes.update(other.dtypes)
        if other.data is not None:
            self.data = self.data or {}
            self.data.update(other.data)
        self.metadata.update(other.metadata)

    def remap_keys_(self, remap: Dict[str, str]) -> None:
        for old, new in remap.items():
            if old not in self.keys:
                continue
            self.keys.remove(old)
            self.keys.add(new)
            for d in (self.seqlens, self.trailing_shapes, self.dtypes):
                d[new] = d.pop(old)
            if self.data is not None:
                self.data[new] = self.data.pop(old)

    def to_device(self, device) -> "SequenceSample":
        if self.data is not None:
            self.data = {k: (v.to(device) if v is not None else None) for k, v in self.data.items()}
        return self

    def cpu(self):
        return self.to_device("cpu")

    # --------------------------------------------------------------------- transport
    def as_json_compatible(self) -> Dict:
        return dict(ids=self.ids, keys=sorted(self.keys), seqlens=self.seqlens,
                    trailing_shapes={k: (list(v) if v is not None else None) for k, v in self.trailing_shapes.items()},
                    dtypes={k: (str(v).replace("torch.", "") if v is not None else None) for k, v in self.dtypes.items()},
                    data=None if self.data is None else
                    {k: (v.cpu().float().numpy().tolist() if v is not None else None) for k, v in self.data.items()},
                    metadata=self.metadata)

    @classmethod
    def from_json_compatible(cls, d: Dict) -> "SequenceSample":
        dtypes = {k: (getattr(torch, v) if v is not None else None) for k, v in d["dtypes"].items()}
        data = None
        if d["data"] is not None:
            data = {k: (torch.tensor(v, dtype=dtypes[k]) if v is not None else None) for k, v in d["data"].items()}
        return cls(keys=set(d["keys"]), ids=d["ids"], seqlens=d["seqlens"], data=data,
                   trailing_shapes={k: (tuple(v) if v is not None else None) for k, v in d["trailing_shapes"].items()},
                   dtypes=dtypes, metadata=d.get("metadata", {}))


@dataclasses.dataclass
class DataBatchMeta:
    dp_rank: int
    meta_sample: Optional[SequenceSample]
    is_final_batch: bool = False
