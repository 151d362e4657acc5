"""Logits warpers for host-side / reference sampling (temperature, top-k, top-p, epsilon cut-off) and
their composition. The rollout engine samples with the fused HIP kernel (``ops.sample``); these classes
are the composable PyTorch form used by evaluation scripts, tests and CPU runs. Interface of
``realhf/impl/model/utils/logits_warper.py:12-225``: a warper maps ``(input_ids, logits) -> logits`` with
filtered entries set to ``filter_value``.
"""
from typing import List, Optional

import torch


class LogitsWarper:
    def __call__(self, input_ids: torch.Tensor, logits: torch.Tensor, inplace: bool = False) -> torch.Tensor:
        raise NotImplementedError()


class TemperatureLogitsWarper(LogitsWarper):
    def __init__(self, temperature: float):
        if not temperature > 0:
            raise ValueError(f"`temperature` has to be a strictly positive float, but is {temperature}")
        self.temperature = float(temperature)

    def __call__(self, input_ids, logits, inplace: bool = False):
        return logits.div_(self.temperature) if inplace else logits / self.temperature


class TopKLogitsWarper(LogitsWarper):
    def __init__(self, top_k: int, filter_value: float = -float("inf"), min_tokens_to_keep: int = 1):
        if top_k <= 0:
            raise ValueError(f"`top_k` has to be a strictly positive integer, but is {top_k}")
        self.top_k = max(int(top_k), min_tokens_to_keep)
        self.filter_value = filter_value

    def __call__(self, input_ids, logits, inplace: bool = False):
        k = min(self.top_k, logits.size(-1))
        kth = torch.topk(logits, k, dim=-1).values[..., -1:]
        remove = logits < kth
        return logits.masked_fill_(remove, self.filter_value) if inplace else logits.masked_fill(remove,
                                                                                                 self.filter_value)


class TopPLogitsWarper(LogitsWarper):
    def __init__(self, top_p: float, filter_value: float = -float("inf"), min_tokens_to_keep: int = 1):
        if not 0 <= top_p <= 1.0:
            raise ValueError(f"`top_p` has to be a float >= 0 and <= 1, but is {top_p}")
        self.top_p, self.filter_value, self.min_keep = float(top_p), filter_value, min_tokens_to_keep

    def __call__(self, input_ids, logits, inplace: bool = False):
        sorted_logits, idx = torch.sort(logits, descending=True, dim=-1)
        cum = sorted_logits.softmax(-1).cumsum(-1)
        remove_sorted = (cum - sorted_logits.softmax(-1)) >= self.top_p  # keep the token that crosses top_p
        remove_sorted[..., : self.min_keep] = False
        remove = torch.zeros_like(remove_sorted).scatter(-1, idx, remove_sorted)
        return logits.masked_fill_(remove, self.filter_value) if inplace else logits.masked_fill(remove,
                                                                                                 self.filter_value)


class EpsilonLogitsWarper(LogitsWarper):
    """Drop tokens whose probability is below `epsilon` (truncation sampling)."""

    def __init__(self, epsilon: float, filter_value: float = -float("inf"), min_tokens_to_keep: int = 1):
        if not 0 < epsilon < 1:
            raise ValueError(f"`epsilon` has to be a float > 0 and < 1, but is {epsilon}")
        self.epsilon, self.filter_value, self.min_keep = float(epsilon), filter_value, min_tokens_to_keep

    def __call__(self, input_ids, logits, inplace: bool = False):
        probs = logits.softmax(-1)
        remove = probs < self.epsilon
        k = min(self.min_keep, logits.size(-1))
        kth = torch.topk(logits, k, dim=-1).values[..., -1:]
        remove = remove & (logits < kth)
        return logits.masked_fill_(remove, self.filter_value) if inplace else logits.masked_fill(remove,
                                                                                                 self.filter_value)


def chained_logits_wraper(xs: List[LogitsWarper], inplace: bool = False):
    """Apply the warpers one after the other."""
    def f(input_ids, logits):
        for x in xs:
            logits = x(input_ids, logits, inplace)
        return logits

    return f


def unioned_logits_wraper(xs: List[LogitsWarper], inplace: bool = False):
    """Filter a token if ANY warper filters it (every warper sees the original logits)."""
    def f(input_ids, logits):
        mask = torch.zeros_like(logits, dtype=torch.bool)
        for x in xs:
            mask |= ~torch.isfinite(x(input_ids, logits, False))
        return logits.masked_fill_(mask, -float("inf")) if inplace else logits.masked_fill(mask, -float("inf"))

    return f


def top_k_top_p_logits(logits: torch.Tensor, top_k: int = 0, top_p: float = 1.0, inplace: bool = False,
                       ordered: bool = False) -> torch.Tensor:
    """Top-k and/or top-p filtering; `ordered` applies top-k first and top-p on the survivors."""
    ws: List[LogitsWarper] = []
    if 0 < top_k < logits.size(-1):
        ws.append(TopKLogitsWarper(top_k))
    if top_p < 1.0:
        ws.append(TopPLogitsWarper(top_p))
    if not ws:
        return logits
    f = chained_logits_wraper(ws, inplace) if ordered else unioned_logits_wraper(ws, inplace)
    return f(None, logits)
