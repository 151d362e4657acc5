"""Direct preference optimisation on packed preference pairs.

Sequence log-probabilities arrive interleaved - chosen_0, rejected_0, chosen_1, rejected_1, ... - once under
the policy and once under the frozen reference. With the implicit reward r = beta * (log pi - log pi_ref) the
loss is -log sigmoid(r_chosen - r_rejected), averaged over pairs. Role of
``realhf/impl/model/utils/dpo_functional.py``.
"""
from typing import Tuple

import torch
import torch.nn.functional as F


def dpo_loss(pi_logps: torch.Tensor, ref_logps: torch.Tensor, beta: float
             ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (loss, sum of chosen rewards, sum of rejected rewards, -sum of log-ratios); only the loss carries
    gradients."""
    if pi_logps.dim() != 1 or pi_logps.numel() % 2 or pi_logps.shape != ref_logps.shape:
        raise AssertionError(f"expected two equal 1-D tensors of even length, got {tuple(pi_logps.shape)} and "
                             f"{tuple(ref_logps.shape)}")
    log_ratio = pi_logps - ref_logps  # implicit reward / beta, one entry per sequence
    chosen, rejected = log_ratio[0::2], log_ratio[1::2]
    loss = F.softplus(-beta * (chosen - rejected)).mean()  # == -logsigmoid(beta * margin)
    with torch.no_grad():
        stats = (beta * chosen.sum(), beta * rejected.sum(), -log_ratio.sum())
    return (loss, *stats)
