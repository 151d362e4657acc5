"""Running mean / std estimators for value normalisation (PopArt-style critics).

Statistics are all-reduced over the data-parallel group in ONE packed collective (count, sum, sum of
squares). Behaviour of ``realhf/impl/model/modules/rms.py:19-192``: exponential (debiased) and
moving-average variants, variance floor 1e-2, normalised values clipped to +-5.
"""
from typing import Optional, Tuple

import torch
import torch.distributed as dist
from torch import nn


def _batch_moments(x: torch.Tensor, mask: Optional[torch.Tensor], dtype, group) -> Tuple[torch.Tensor, torch.Tensor]:
    x = x.to(dtype)
    if mask is None:
        cnt = torch.tensor(float(x.numel()), dtype=dtype, device=x.device)
    else:
        m = mask.to(dtype)
        x, cnt = x * m, m.sum()
    packed = torch.stack([cnt, x.sum(), x.square().sum()])
    if dist.is_initialized() and (group is not None or dist.get_world_size() > 1):
        dist.all_reduce(packed, group=group)
    return packed[1] / packed[0], packed[2] / packed[0]


class _RunningMeanStd(nn.Module):
    def __init__(self, epsilon: float, high_precision: bool, group=None):
        super().__init__()
        self.eps = epsilon
        self.dt = torch.float64 if high_precision else torch.float32
        self.group = group

    def forward(self, *a, **k):
        raise NotImplementedError("use update / normalize / denormalize")

    @torch.no_grad()
    def normalize(self, x):
        mean, std = self.mean_std()
        return ((x.to(self.dt) - mean) / std).clip(-5, 5).float()

    @torch.no_grad()
    def denormalize(self, x):
        mean, std = self.mean_std()
        return (x.to(self.dt) * std + mean).float()


class ExponentialRunningMeanStd(_RunningMeanStd):
    def __init__(self, beta: float = 0.999, epsilon: float = 1e-5, high_precision: bool = True, group=None):
        super().__init__(epsilon, high_precision, group)
        self.beta = beta
        self.register_buffer("mean", torch.zeros(1, dtype=self.dt))
        self.register_buffer("mean_sq", torch.zeros(1, dtype=self.dt))
        self.register_buffer("debias", torch.zeros(1, dtype=self.dt))

    @torch.no_grad()
    def update(self, x: torch.Tensor, mask: Optional[torch.Tensor] = None):
        m, m2 = _batch_moments(x, mask, self.dt, self.group)
        b = self.beta
        self.mean.mul_(b).add_(m.to(self.mean.device) * (1 - b))
        self.mean_sq.mul_(b).add_(m2.to(self.mean.device) * (1 - b))
        self.debias.mul_(b).add_(1 - b)

    @torch.no_grad()
    def mean_std(self):
        d = self.debias.clamp(min=self.eps)
        mean, msq = self.mean / d, self.mean_sq / d
        return mean, (msq - mean ** 2).clamp(min=1e-2).sqrt()


class MovingAverageRunningMeanStd(_RunningMeanStd):
    def __init__(self, epsilon: float = 1e-5, high_precision: bool = True, group=None):
        super().__init__(epsilon, high_precision, group)
        self.register_buffer("mean", torch.zeros(1, dtype=self.dt))
        self.register_buffer("mean_sq", torch.zeros(1, dtype=self.dt))
        self.register_buffer("count", torch.zeros(1, dtype=self.dt))

    @torch.no_grad()
    def update(self, x: torch.Tensor, mask: Optional[torch.Tensor] = None):
        m, m2 = _batch_moments(x, mask, self.dt, self.group)
        n = self.count + 1
        self.mean.add_((m.to(self.mean.device) - self.mean) / n)
        self.mean_sq.add_((m2.to(self.mean.device) - self.mean_sq) / n)
        self.count.copy_(n)

    @torch.no_grad()
    def mean_std(self):
        return self.mean, (self.mean_sq - self.mean ** 2).clamp(min=1e-2).sqrt()
