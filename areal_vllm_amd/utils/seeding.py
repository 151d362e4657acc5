"""Process-wide random seeds.

Every process of a job derives its seed from the experiment's base seed and a key naming the process (for
example ``trainer3``), so that ranks draw different streams while a re-run reproduces each of them. Dataset
shuffles take `get_shuffle_seed()`, a counter on top of the base seed, so successive epochs differ but all
ranks that count the same number of epochs agree. Same contract as ``areal/utils/seeding.py``.
"""
import hashlib
import itertools
import os
import random
from typing import Optional

import numpy as np
import torch

_MOD = 2**31 - 1


class _Seeds:
    base: Optional[int] = None
    shuffles = itertools.count(1)


def derive_seed(base_seed: int, key: str) -> int:
    """Seed of the process called `key` in a job seeded with `base_seed` (stable across runs and machines)."""
    digest = hashlib.sha256(key.encode()).digest()
    return (int(base_seed) + int.from_bytes(digest, "big") % _MOD) % _MOD


def seed_everything(seed: int) -> None:
    os.environ["PYTHONHASHSEED"] = str(seed)
    for seeder in (random.seed, np.random.seed, torch.manual_seed):
        seeder(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def set_random_seed(base_seed: int, key: str) -> int:
    """Seed python, numpy and torch (all devices) for this process; -> the derived seed."""
    _Seeds.base, _Seeds.shuffles = int(base_seed), itertools.count(1)
    seed = derive_seed(base_seed, key)
    seed_everything(seed)
    return seed


def get_seed() -> int:
    if _Seeds.base is None:
        raise RuntimeError("set_random_seed has not been called in this process")
    return _Seeds.base


def get_shuffle_seed() -> int:
    return get_seed() + next(_Seeds.shuffles)
