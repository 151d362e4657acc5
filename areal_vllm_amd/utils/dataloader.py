"""A resumable data loader (the role of torchdata's StatefulDataLoader in the reference's scripts).

Yields lists of dataset items (``collate_fn`` defaults to identity, matching ``collate_fn=lambda x: x`` in
``examples/math/gsm8k_grpo.py:68-83``), reshuffles deterministically per epoch and can be saved / restored
in the middle of an epoch through ``state_dict`` / ``load_state_dict``.
"""
from typing import Any, Callable, Dict, Iterator, List, Optional, Sequence

import numpy as np


class StatefulDataLoader:
    def __init__(self, dataset: Sequence, batch_size: int = 1, shuffle: bool = False, drop_last: bool = True,
                 collate_fn: Optional[Callable[[List[Any]], Any]] = None, seed: int = 0, num_workers: int = 0,
                 pin_memory: bool = False):
        self.dataset = dataset
        self.batch_size = batch_size
        self.shuffle = shuffle
        self.drop_last = drop_last
        self.collate_fn = collate_fn or (lambda x: x)
        self.seed = seed
        self._epoch = 0
        self._pos = 0  # number of batches already yielded in the current epoch
        self._resume = False

    def __len__(self) -> int:
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _order(self, epoch: int) -> np.ndarray:
        n = len(self.dataset)
        if not self.shuffle:
            return np.arange(n)
        return np.random.RandomState(self.seed + epoch).permutation(n)

    def __iter__(self) -> Iterator:
        if not self._resume:
            self._pos = 0
        self._resume = False
        order = self._order(self._epoch)
        nb = len(self)
        while self._pos < nb:
            idx = order[self._pos * self.batch_size:(self._pos + 1) * self.batch_size]
            self._pos += 1
            yield self.collate_fn([self.dataset[int(i)] for i in idx])
        self._epoch += 1
        self._pos = 0

    def state_dict(self) -> Dict[str, int]:
        return dict(epoch=self._epoch, pos=self._pos, seed=self.seed)

    def load_state_dict(self, sd: Dict[str, int]):
        self._epoch = int(sd["epoch"])
        self._pos = int(sd["pos"])
        self.seed = int(sd.get("seed", self.seed))
        self._resume = True
