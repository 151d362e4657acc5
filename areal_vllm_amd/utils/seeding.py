"""Deterministic seeding per (base seed, process key) (``areal/utils/seeding.py`` in the reference)."""
import hashlib
import os
import random

import numpy as np
import torch

_BASE_SEED = None
_SHUFFLER = None


def _seed_from_key(key: str) -> int:
    return int(hashlib.sha256(key.encode()).hexdigest(), 16) % (2**31 - 1)


def set_random_seed(base_seed: int, key: str) -> int:
    global _BASE_SEED
    _BASE_SEED = base_seed
    seed = (base_seed + _seed_from_key(key)) % (2**31 - 1)
    os.environ["PYTHONHASHSEED"] = str(seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    return seed


def get_seed() -> int:
    assert _BASE_SEED is not None, "set_random_seed has not been called"
    return _BASE_SEED


class _Shuffler:
    def __init__(self):
        self.cnt = 0

    def next_shuffle(self) -> int:
        self.cnt += 1
        return self.cnt


def get_shuffle_seed() -> int:
    global _SHUFFLER
    if _SHUFFLER is None:
        _SHUFFLER = _Shuffler()
    return get_seed() + _SHUFFLER.next_shuffle()
