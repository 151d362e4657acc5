"""Dataset registry: ``get_custom_dataset(path, rank, world_size, type=sft|rl, split, ...)``.

Mirrors the dispatch of ``areal/dataset/__init__.py:8-99``: the dataset family is recognised from the
path (``gsm8k``, ``clevr_count_70k``, ``geometry3k``), the result is sharded by data-parallel rank.
Local ``.jsonl`` / ``.json`` / ``.parquet`` files and directories work without network access.
"""
from typing import Optional

VALID_DATASETS = ["gsm8k", "clevr_count_70k", "geometry3k", "countdown"]
_CUSTOM = {}


def register_dataset(key: str, fn) -> None:
    """Serve every path containing `key` with `fn(path=, split=, type=, rank=, world_size=, tokenizer=,
    processor=, max_length=, **kw)` - how example scripts plug their own data in."""
    _CUSTOM[key] = fn


def load_raw_dataset(path: str, split: Optional[str] = None, name: Optional[str] = None):
    """`datasets.load_dataset` that understands plain local files next to hub / saved datasets."""
    import os

    from datasets import load_dataset, load_from_disk

    if os.path.isfile(path):
        ext = os.path.splitext(path)[1].lstrip(".")
        kind = {"jsonl": "json", "json": "json", "parquet": "parquet", "csv": "csv"}[ext]
        return load_dataset(kind, data_files=path, split="train")
    if os.path.isdir(path):
        if os.path.exists(os.path.join(path, "dataset_info.json")) or os.path.exists(os.path.join(path, "state.json")):
            ds = load_from_disk(path)
            return ds[split] if split is not None and hasattr(ds, "keys") and split in ds else ds
        for ext, kind in (("jsonl", "json"), ("json", "json"), ("parquet", "parquet")):
            for cand in (f"{split}.{ext}", os.path.join(name or "main", f"{split}-00000-of-00001.{ext}")):
                f = os.path.join(path, cand)
                if os.path.exists(f):
                    return load_dataset(kind, data_files=f, split="train")
    return load_dataset(path=path, name=name, split=split)


def get_custom_dataset(path: str, rank: int, world_size: int, type: str = "sft", split: Optional[str] = None,
                       max_length: Optional[int] = None, tokenizer=None, processor=None, **kwargs):
    if path.startswith("synthetic:"):
        from .synthetic import get_synthetic_dataset

        return get_synthetic_dataset(path, rank, world_size, type=type, split=split, **kwargs)
    for key, fn in _CUSTOM.items():
        if key in path:
            return fn(path=path, split=split, type=type, rank=rank, world_size=world_size, tokenizer=tokenizer,
                      processor=processor, max_length=max_length, **kwargs)
    if "countdown" in path and type == "rl":
        from .countdown import get_countdown_rl_dataset

        return get_countdown_rl_dataset(path=path, split=split, tokenizer=tokenizer, rank=rank, world_size=world_size,
                                        max_length=max_length, **kwargs)
    if "gsm8k" in path:
        from . import gsm8k as m

        fn = {"sft": m.get_gsm8k_sft_dataset, "rl": m.get_gsm8k_rl_dataset}.get(type)
        if fn is not None:
            return fn(path=path, split=split, tokenizer=tokenizer, rank=rank, world_size=world_size,
                      max_length=max_length, **kwargs)
    elif "clevr_count_70k" in path:
        from . import clevr_count_70k as m

        fn = {"sft": m.get_clevr_count_70k_sft_dataset, "rl": m.get_clevr_count_70k_rl_dataset}.get(type)
        if fn is not None:
            return fn(path=path, split=split, processor=processor, rank=rank, world_size=world_size,
                      max_length=max_length, **kwargs)
    elif "geometry3k" in path:
        from . import geometry3k as m

        fn = {"sft": m.get_geometry3k_sft_dataset, "rl": m.get_geometry3k_rl_dataset}.get(type)
        if fn is not None:
            return fn(path=path, split=split, processor=processor, rank=rank, world_size=world_size,
                      max_length=max_length, **kwargs)
    raise ValueError(f"Dataset {path} with split {split} and training type {type} is not supported. "
                     f"Supported datasets are: {VALID_DATASETS}. ")
