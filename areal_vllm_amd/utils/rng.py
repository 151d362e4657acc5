"""Keys for dropout masks under tensor / pipeline / data parallelism and activation recomputation.

A mask is a pure function of (base seed, training micro-batch counter, layer, site inside the layer, replica, and -
only where the masked activation is sharded over the tensor-parallel group - the tensor-parallel rank): nothing is
drawn from a stateful generator. That gives the three properties the reference gets from its RNG tracker
(``realhf/impl/model/utils/random.py``: a default generator shared by the tensor-parallel group plus a
"model-parallel" one seeded per rank, both saved and restored around recomputed segments):

* activations REPLICATED over the tensor-parallel group (the residual stream without sequence parallelism) get the
  SAME mask on every rank of the group, so the ranks stay replicas;
* activations SHARDED over the group (the token-sharded residual stream under sequence parallelism) get different
  masks per rank, so no pattern repeats across shards;
* a recomputed forward (gradient checkpointing) of the same micro-batch sees the same key, hence the same mask,
  without any generator state to carry.
"""
from typing import Optional

import torch

_MASK64 = (1 << 64) - 1


def _mix(*parts: int) -> int:
    """splitmix64-style hash of a tuple of integers -> 63-bit seed."""
    x = 0x9E3779B97F4A7C15
    for p in parts:
        x = (x ^ (int(p) & _MASK64)) & _MASK64
        x = (x + 0x9E3779B97F4A7C15) & _MASK64
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _MASK64
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _MASK64
        x ^= x >> 31
    return x & ((1 << 63) - 1)


class DropoutKeys:
    def __init__(self):
        self.seed, self.replica, self.tp_rank, self.counter = 0, 0, 0, 0

    def configure(self, seed: int, replica: int = 0, tp_rank: int = 0):
        """`replica`: index of this rank's data-parallel replica (different data, different masks); `tp_rank`: position
        in the tensor-parallel group."""
        self.seed, self.replica, self.tp_rank = int(seed), int(replica), int(tp_rank)

    def advance(self):
        """Once per training micro-batch forward (NOT per recomputation of it)."""
        self.counter += 1

    def seed_for(self, layer: int, site: int, sharded: bool) -> int:
        return _mix(self.seed, self.counter, layer + 1, site, self.replica, self.tp_rank + 1 if sharded else 0)


keys = DropoutKeys()


def dropout(x: torch.Tensor, p: float, layer: int, site: int, sharded: bool = False, training: bool = True,
            state: Optional[DropoutKeys] = None) -> torch.Tensor:
    """Inverted dropout with the mask keyed as described above. `site`: 0 embeddings, 1 attention output, 2 MLP
    output (any integer that tells the call sites of a layer apart)."""
    if not training or p <= 0.0:
        return x
    assert p < 1.0
    g = torch.Generator(device=x.device)
    g.manual_seed((state or keys).seed_for(layer, site, sharded))
    keep = torch.rand(x.shape, generator=g, device=x.device, dtype=torch.float32) >= p
    return x * keep.to(x.dtype) * (1.0 / (1.0 - p))
