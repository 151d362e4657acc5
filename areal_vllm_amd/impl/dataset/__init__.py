"""Datasets of the worker runtime. Every item is a one-item `SequenceSample`; a worker with index i of n
loads the i-th contiguous shard of a seed-shuffled index list, so data-parallel heads (model workers) or
rollout workers never see the same sample. Registry + sharding follow ``realhf/api/core/data_api.py``
(``register_dataset`` / ``make_dataset`` / ``load_shuffle_split_dataset``)."""
import json
from typing import Any, Callable, Dict, List, Optional

import numpy as np

ALL_DATASET_CLASSES: Dict[str, Any] = {}


def register_dataset(name: str, cls):
    assert name not in ALL_DATASET_CLASSES, name
    ALL_DATASET_CLASSES[name] = cls


def make_dataset(cfg, seed: int, dp_rank: int, world_size: int, tokenizer_or_path=None):
    from . import math_code_dataset, prompt_answer_dataset, prompt_dataset, rw_paired_dataset  # noqa: F401

    tok = tokenizer_or_path
    if isinstance(tok, str):
        from ...utils.hf_utils import load_hf_tokenizer

        tok = load_hf_tokenizer(tok)
    type_ = cfg.type_ if hasattr(cfg, "type_") else cfg["type_"]
    args = cfg.args if hasattr(cfg, "args") else cfg.get("args", {})
    return ALL_DATASET_CLASSES[type_](seed=seed, dp_rank=dp_rank, world_size=world_size, tokenizer=tok, **args)


def load_shuffle_split_dataset(seed: int, dp_rank: int, world_size: int, dataset_path: Optional[str] = None,
                               dataset_builder: Optional[Callable[[], List[Dict]]] = None) -> List[Dict]:
    if dataset_path is not None:
        if dataset_path.endswith(".jsonl"):
            with open(dataset_path) as f:
                data = [json.loads(l) for l in f if l.strip()]
        elif dataset_path.endswith(".json"):
            with open(dataset_path) as f:
                data = json.load(f)
        else:
            raise NotImplementedError(f"unknown dataset extension: {dataset_path}")
    else:
        assert dataset_builder is not None
        data = dataset_builder()
    if any("id" not in d and "query_id" not in d for d in data):
        for i, d in enumerate(data):
            d.setdefault("id", d.get("query_id", str(i)))
    n = len(data)
    order = np.random.RandomState(seed).permutation(n)
    base, extra = divmod(n, world_size)
    sizes = [base + (1 if r < extra else 0) for r in range(world_size)]
    start = sum(sizes[:dp_rank])
    return [data[i] for i in order[start:start + sizes[dp_rank]]]
