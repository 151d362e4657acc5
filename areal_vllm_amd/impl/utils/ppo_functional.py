"""PPO numerics on packed (1-D, variable-length) batches.

KL controllers, packed reward assembly (sparse terminal reward or dense per-token reward, KL penalty),
packed GAE through the HIP wave-scan kernel, and the actor / critic losses. Counterpart of
``realhf/impl/model/utils/ppo_functional.py:14-400``.

Layout convention ("misaligned" in the reference): for a sequence of length n the log-probs / rewards
have n-1 entries (one per predicted token) while values have n entries (bootstrap value included);
``short1cu_seqlens`` are the cumulative lengths of the n-1 arrays.
"""
from typing import Dict, Optional, Tuple

import torch

from ... import ops
from ...utils.functional import ppo_actor_loss_fn, ppo_critic_loss_fn


class KLController:
    def __init__(self, kl_coef: float):
        self.value = kl_coef

    def update(self, current, n_steps):
        raise NotImplementedError()


class FixedKLController(KLController):
    def update(self, current, n_steps):
        pass


class AdaptiveKLController(KLController):
    """Proportional controller towards a target KL (Ziegler et al., 2019): the coefficient is multiplied
    by ``1 + clip(kl/target - 1, +-0.2) * n_steps / horizon``."""

    def __init__(self, init_kl_coef: float, target: float, horizon: float):
        super().__init__(init_kl_coef)
        self.target, self.horizon = target, horizon

    def update(self, current, n_steps):
        err = max(-0.2, min(0.2, float(current) / self.target - 1.0))
        self.value = self.value * (1.0 + err * n_steps / self.horizon)


def actor_loss_fn(logprobs, old_logprobs, advantages, eps_clip: float, loss_mask, c_clip: Optional[float] = None,
                  proximal_logprobs=None, behav_imp_weight_cap: Optional[float] = None,
                  eps_clip_higher: Optional[float] = None):
    """(Decoupled) clipped policy loss; without `proximal_logprobs` the behaviour policy is the trust-region
    centre (standard PPO)."""
    prox = old_logprobs if proximal_logprobs is None else proximal_logprobs
    return ppo_actor_loss_fn(logprobs, prox, old_logprobs, advantages, eps_clip, loss_mask,
                             eps_clip_higher=eps_clip_higher, c_clip=c_clip,
                             behav_imp_weight_cap=behav_imp_weight_cap if proximal_logprobs is not None else None)


critic_loss_fn = ppo_critic_loss_fn


@torch.no_grad()
def get_packed_rewards(kl_ctl: float, clip_reward_value: float, log_probs: torch.Tensor, ref_log_probs: torch.Tensor,
                       reward_score: torch.Tensor, short1cu_seqlens: torch.Tensor, seq_no_eos_mask: torch.Tensor,
                       mask_no_eos_with_zero: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """KL penalty on every token plus the clipped sequence reward on the last predicted token."""
    tot = -kl_ctl * (log_probs - ref_log_probs)
    kl_rewards = tot.clone()
    score = reward_score.clip(-clip_reward_value, clip_reward_value)
    if mask_no_eos_with_zero:
        score = torch.where(seq_no_eos_mask, torch.zeros_like(score), score)
    tot[short1cu_seqlens[1:].long() - 1] += score
    return kl_rewards, tot


@torch.no_grad()
def get_packed_reward_dense(kl_ctl: float, clip_reward_value: float, log_probs: torch.Tensor,
                            ref_log_probs: torch.Tensor, dense_reward_score: torch.Tensor,
                            short1cu_seqlens: torch.Tensor, seq_no_eos_mask: torch.Tensor, reward_delta: bool):
    """Per-token rewards: `dense_reward_score` has n entries per sequence; token t receives score[t+1]
    (or the increment score[t+1] - score[t] when `reward_delta`)."""
    tot = -kl_ctl * (log_probs - ref_log_probs)
    kl_rewards = tot.clone()
    d = dense_reward_score.clip(-clip_reward_value, clip_reward_value)
    lens = (short1cu_seqlens[1:] - short1cu_seqlens[:-1]).long()
    # index of dense[t+1] for every short token: shift by the sequence index (each sequence has one extra)
    seq_id = torch.repeat_interleave(torch.arange(lens.numel(), device=lens.device), lens)
    pos = torch.arange(tot.numel(), device=tot.device)
    nxt = d[pos + seq_id + 1]
    tot += (nxt - d[pos + seq_id]) if reward_delta else nxt
    return kl_rewards, tot


def pygae1d_nolp_misalign(rewards, values, cu_seqlens, bootstrap, gamma: float, lam: float):
    """Plain Python reference of the packed GAE (`cu_seqlens` over the n-1 arrays)."""
    adv = torch.zeros_like(rewards)
    ret = torch.zeros_like(rewards)
    cu = cu_seqlens.tolist()
    for i in range(len(cu) - 1):
        s, e = cu[i], cu[i + 1]
        v = values[s + i:e + i + 1].clone()
        if not bool(bootstrap[i]):
            v[-1] = 0
        last = 0.0
        for t in reversed(range(e - s)):
            delta = rewards[s + t] + gamma * v[t + 1] - v[t]
            last = delta + gamma * lam * last
            adv[s + t] = last
        ret[s:e] = adv[s:e] + v[:-1]
    return adv, ret


@torch.no_grad()
def get_packed_advantages_and_returns(gamma: float, lam: float, values: torch.Tensor, rewards: torch.Tensor,
                                      short1cu_seqlens: torch.Tensor, seq_no_eos_mask: torch.Tensor):
    """Packed GAE on the device kernel (one wave-scan per sequence)."""
    return ops.gae_1d_packed(rewards.float(), values.float(), short1cu_seqlens.int(), seq_no_eos_mask.bool(),
                             gamma, lam)
