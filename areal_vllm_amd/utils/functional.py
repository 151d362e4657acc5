"""RL math: masked normalisation, decoupled PPO actor loss, critic loss, dynamic sampling, DAPO penalty.

Formulae follow the reference (``areal/utils/functional.py:83-259``;
``realhf/impl/model/utils/ppo_functional.py:51-226``); reductions that the reference runs as Python loops
over groups / samples are vectorised, and the three scalar all-reduces of the batch-level
normalisation are packed into one.
"""
import warnings
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

from ..ops.logprob import gather_logprobs, gather_logprobs_entropy  # noqa: F401  (public re-export)


def _packed_all_reduce(tensors, group):
    """One SUM all-reduce for several same-dtype tensors."""
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    out, o = [], 0
    for t in tensors:
        out.append(flat[o:o + t.numel()].view_as(t))
        o += t.numel()
    return out


@torch.no_grad()
def masked_normalization(x: torch.Tensor, mask: Optional[torch.Tensor] = None, dim=None, unbiased=False, eps=1e-5,
                         high_precision=True, all_reduce=True, reduce_group=None) -> torch.Tensor:
    dtype = torch.float64 if high_precision else torch.float32
    x = x.to(dtype)
    if dim is None:
        dim = tuple(range(x.dim()))
    if mask is None:
        factor = torch.tensor(float(np.prod([x.shape[d] for d in dim])), dtype=dtype, device=x.device).reshape(
            [1] * x.dim())
    else:
        mask = mask.to(dtype)
        x = x * mask
        factor = mask.sum(dim, keepdim=True)
    x_sum = x.sum(dim=dim, keepdim=True)
    x_sum_sq = x.square().sum(dim=dim, keepdim=True)
    if dist.is_initialized() and all_reduce and dist.get_world_size(reduce_group) > 1:
        factor, x_sum, x_sum_sq = _packed_all_reduce([factor, x_sum, x_sum_sq], reduce_group)
    mean = x_sum / factor
    var = x_sum_sq / factor - mean**2
    if unbiased:
        var = var * factor / (factor - 1)
    return ((x - mean) / (var.clamp_min(0).sqrt() + eps)).float()


def ppo_actor_loss_fn(logprobs: torch.Tensor, proximal_logprobs: torch.Tensor, old_logprobs: torch.Tensor,
                      advantages: torch.Tensor, eps_clip: float, loss_mask: torch.Tensor,
                      eps_clip_higher: Optional[float] = None, c_clip: Optional[float] = None,
                      behav_imp_weight_cap: Optional[float] = None) -> Tuple[torch.Tensor, Dict]:
    """Decoupled PPO objective: the trust region is centred on the proximal policy (recomputed log-probs),
    the behaviour policy (rollout log-probs) only enters through an importance weight; tokens whose
    behaviour weight exceeds the cap are dropped from the loss."""
    loss_mask = loss_mask.bool()
    cnt = loss_mask.count_nonzero().clamp_min(1)
    ratio = torch.where(loss_mask, torch.exp(logprobs - proximal_logprobs), 0)
    hi = eps_clip if eps_clip_higher is None else eps_clip_higher
    clipped_ratio = torch.clamp(ratio, 1.0 - eps_clip, 1.0 + hi)
    pg1 = -advantages * ratio
    pg2 = -advantages * clipped_ratio
    clip_mask = pg1.detach() < pg2.detach()
    pg = torch.max(pg1, pg2)
    if c_clip is not None:
        assert c_clip > 1.0, c_clip
        pg3 = torch.sign(advantages) * c_clip * advantages
        dual_clip_mask = pg3.detach() < pg.detach()
        pg = torch.min(pg, pg3)
    else:
        dual_clip_mask = torch.zeros_like(clip_mask)
    behav_kl = proximal_logprobs - old_logprobs
    behav_w = behav_kl.exp()
    behav_mask = (behav_w <= behav_imp_weight_cap).logical_and(loss_mask) if behav_imp_weight_cap is not None \
        else loss_mask
    behav_kl = torch.where(behav_mask, behav_kl, 0.0)
    behav_w = torch.where(behav_mask, behav_w, 0.0)
    pg = pg * behav_w
    logging_loss = pg.detach()
    loss = torch.where(loss_mask, pg, 0).sum() / cnt
    stat = dict(loss=logging_loss, importance_weight=ratio.detach(), approx_kl=(logprobs - proximal_logprobs).detach(),
                clip_mask=clip_mask & loss_mask, dual_clip_mask=dual_clip_mask & loss_mask,
                behave_imp_weight=behav_w, behave_approx_kl=behav_kl, behave_mask=behav_mask)
    return loss, stat


def ppo_critic_loss_fn(value: torch.Tensor, old_value: torch.Tensor, target_value: torch.Tensor, value_eps_clip: float,
                       loss_mask: Optional[torch.Tensor] = None, loss_fn_type: str = "mse") -> Tuple[torch.Tensor, Dict]:
    """Clipped value loss (``realhf/impl/model/utils/ppo_functional.py:161-226``)."""
    if loss_fn_type == "huber":
        fn = lambda a, b: torch.nn.functional.huber_loss(a, b, reduction="none", delta=10.0)  # noqa: E731
    elif loss_fn_type == "mse":
        fn = lambda a, b: 0.5 * (a - b) ** 2  # noqa: E731
    else:
        raise NotImplementedError(loss_fn_type)
    orig = fn(value, target_value)
    clipped_v = old_value + (value - old_value).clamp(-value_eps_clip, value_eps_clip)
    clipped = fn(clipped_v, target_value)
    loss = torch.max(orig, clipped)
    clip_mask = clipped.detach() > orig.detach()
    if loss_mask is not None:
        m = loss_mask.bool()
        clip_mask = clip_mask & m
        loss = torch.where(m, loss, 0).sum() / m.count_nonzero().clamp_min(1)
    else:
        loss = loss.mean()
    return loss, dict(clip_mask=clip_mask, loss=orig.detach())


def dynamic_sampling(data: Dict[str, torch.Tensor], group_size: int):
    """Drop groups whose rewards are all identical (DAPO). Group members must be adjacent."""
    rewards = data["rewards"]
    bs = rewards.shape[0]
    if bs % group_size != 0:
        warnings.warn("The batch size is not divisible by the group size. Return the original data")
        return data, dict(n_group_keeped=bs // max(1, group_size), n_group_filtered=0)
    ng = bs // group_size
    r = rewards.view(ng, group_size)
    valid = ~(r == r[:, 0:1]).all(dim=1)
    mask = valid.repeat_interleave(group_size)
    if not mask.any():
        return data, dict(n_group_keeped=0, n_group_filtered=ng)
    out = {}
    for k, v in data.items():
        if torch.is_tensor(v) and v.shape[0] == bs:
            out[k] = v[mask]
        elif isinstance(v, list) and len(v) == bs:
            out[k] = [x for x, keep in zip(v, mask.tolist()) if keep]
        else:
            out[k] = v
    kept = int(valid.sum().item())
    return out, dict(n_group_keeped=kept, n_group_filtered=ng - kept)


def reward_overlong_penalty(data: Dict[str, torch.Tensor], overlong_tokens: int, overlong_penalty_factor: float,
                            max_response_length: int) -> Dict[str, torch.Tensor]:
    """Soft length penalty: responses entering the last `overlong_tokens` of the budget lose up to
    `overlong_penalty_factor` reward, linearly."""
    resp_len = data["loss_mask"].sum(dim=-1).to(data["rewards"].dtype)
    exceed = resp_len - (max_response_length - overlong_tokens)
    penalty = (-exceed / overlong_tokens * overlong_penalty_factor).clamp(max=0)
    data["rewards"] = data["rewards"] + penalty
    return data
