"""Direct preference optimisation loss on packed (chosen, rejected) pairs
(``realhf/impl/model/utils/dpo_functional.py``): sequence log-probs arrive interleaved
[chosen_0, rejected_0, chosen_1, ...]."""
from typing import Tuple

import torch
import torch.nn.functional as F


def dpo_loss(pi_logps: torch.Tensor, ref_logps: torch.Tensor, beta: float
             ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (loss, chosen score sum, rejected score sum, -sum(pi - ref)); scores are beta-scaled log-ratios."""
    assert pi_logps.dim() == 1 and pi_logps.numel() % 2 == 0 and pi_logps.shape == ref_logps.shape, (
        pi_logps.shape, ref_logps.shape)
    pi, ref = pi_logps.view(-1, 2), ref_logps.view(-1, 2)
    margin = (pi[:, 0] - pi[:, 1]) - (ref[:, 0] - ref[:, 1])
    loss = -F.logsigmoid(beta * margin).mean()
    pos = beta * (pi[:, 0] - ref[:, 0]).detach().sum()
    neg = beta * (pi[:, 1] - ref[:, 1]).detach().sum()
    kl = -(pi - ref).detach().sum()
    return loss, pos, neg, kl
