"""PPO / GRPO actor: log-prob recompute, advantage estimation, minibatched decoupled-PPO update.

Algorithm steps and statistics follow ``areal/engine/ppo/actor.py:48-604`` in the reference (reward
shaping, KL-regularised token rewards, GAE with optional bootstrap for truncated sequences, AdvNorm with
independent mean/std levels, dynamic sampling, overlong penalty, per-minibatch optimizer steps). The
differences are mechanical: GAE is one kernel launch (``ops.gae_padded``) instead of a Python loop over
time steps, group statistics are vectorised, and the loss consumes the fused LM-head log-prob kernel via
the engine's ``LogitsProxy`` so fp32 ``[T, V]`` logits never exist.
"""
import functools
from typing import Any, Dict, List, Optional

import numpy as np
import torch
import torch.distributed as dist

from ... import ops
from ...api.cli_args import MicroBatchSpec, PPOActorConfig
from ...api.engine_api import TrainEngine
from ...utils import stats_tracker
from ...utils.data import split_padded_tensor_dict_into_mb_list
from ...utils.functional import (
    dynamic_sampling,
    masked_normalization,
    ppo_actor_loss_fn,
    reward_overlong_penalty,
)
from ..train_engine import LogitsProxy, NativeTrainEngine


def _pp_share(engine, stats):
    """Loss statistics are recorded on the last pipeline stage; give every stage the union."""
    f = getattr(engine, "share_pipeline_stats", None)
    return f(stats) if f is not None else stats


class AdvNorm:
    """Advantage normalisation with independent levels for the mean and the std:
    "batch" (all-reduced over the data-parallel group), "group" (consecutive `group_size` rows), "none"."""

    def __init__(self, cfg):
        if cfg.mean_level not in {"batch", "group", "none"}:
            raise ValueError(f"mean_level must be 'batch', 'group' or 'none', got {cfg.mean_level}")
        if cfg.std_level not in {"batch", "group", "none"}:
            raise ValueError(f"std_level must be 'batch', 'group', or 'none', got {cfg.std_level}")
        if (cfg.mean_level == "group" or cfg.std_level == "group") and cfg.group_size is None:
            raise ValueError("group_size must be provided if using group normalization")
        self.mean_level, self.std_level, self.group_size = cfg.mean_level, cfg.std_level, cfg.group_size

    @staticmethod
    def _reduce(ts, all_reduce, group):
        if all_reduce and dist.is_initialized() and dist.get_world_size(group) > 1:
            flat = torch.stack([t.reshape(()) for t in ts])
            dist.all_reduce(flat, group=group)
            return [flat[i] for i in range(len(ts))]
        return ts

    def _grouped(self, x):
        bs = x.shape[0]
        g = self.group_size
        n = bs // g
        return x[: n * g].reshape(n, -1), n * g

    @torch.no_grad()
    def __call__(self, advantages: torch.Tensor, loss_mask: Optional[torch.Tensor] = None, eps: float = 1e-5,
                 unbiased: bool = False, high_precision: bool = True, reduce_group=None) -> torch.Tensor:
        dtype = torch.float64 if high_precision else torch.float32
        if self.mean_level == "none" and self.std_level == "none":
            return advantages.float()
        if self.mean_level == self.std_level == "batch":
            return masked_normalization(advantages, loss_mask, unbiased=unbiased, eps=eps,
                                        high_precision=high_precision, all_reduce=True, reduce_group=reduce_group)
        x = advantages.to(dtype)
        m = torch.ones_like(x) if loss_mask is None else loss_mask.to(dtype)
        xm = x * m
        shape = x.shape

        def batch_stat(v, w):
            s, c = self._reduce([v.sum(), w.sum()], True, reduce_group)
            return s, c

        def group_view(t):
            gv, n = self._grouped(t)
            return gv, n

        if self.mean_level == self.std_level == "group":
            # per-group masked_normalization (operates on x*mask like the batch form)
            gx, n = group_view(xm)
            gm, _ = group_view(m)
            f = gm.sum(-1, keepdim=True)
            mean = gx.sum(-1, keepdim=True) / f
            var = gx.square().sum(-1, keepdim=True) / f - mean**2
            if unbiased:
                var = var * f / (f - 1)
            out = (gx - mean) / (var.clamp_min(0).sqrt() + eps)
            res = xm.clone()
            res[:n] = out.reshape(n, *shape[1:])
            return res.float()
        # ---- mixed levels
        if self.mean_level == "batch":
            s, c = batch_stat(xm, m)
            mean = (s / c).expand_as(x)
        elif self.mean_level == "group":
            gx, n = group_view(xm)
            gm, _ = group_view(m)
            gmean = gx.sum(-1, keepdim=True) / gm.sum(-1, keepdim=True)
            mean = torch.zeros_like(x)
            mean[:n] = gmean.expand(-1, gx.shape[1]).reshape(n, *shape[1:])
        else:
            mean = torch.zeros_like(x)
        centered = x - mean
        if self.std_level == "none":
            return centered.float()
        cm = xm - mean * m
        if self.std_level == "batch":
            ssq, c = batch_stat(cm.square(), m)
            var = ssq / c
            if unbiased:
                var = var * c / (c - 1)
            std = var.sqrt().expand_as(x)
        else:
            gc, n = group_view(cm)
            gm, _ = group_view(m)
            f = gm.sum(-1, keepdim=True)
            var = gc.square().sum(-1, keepdim=True) / f
            if unbiased:
                var = var * f / (f - 1)
            std = torch.zeros_like(x)
            std[:n] = var.sqrt().expand(-1, gc.shape[1]).reshape(n, *shape[1:])
        return (centered / (std + eps)).float()


class PPOActor:
    def __init__(self, config: PPOActorConfig, engine: TrainEngine):
        self.config = config
        self.engine = engine
        self.reward_bias = config.reward_bias
        self.reward_scaling = config.reward_scaling
        self.reward_clip = config.reward_clip
        self.group_reward_norm = config.group_reward_norm
        self.group_size = config.group_size
        self.kl_ctl = config.kl_ctl
        self.adv_norm = AdvNorm(config.adv_norm) if config.adv_norm is not None else None
        self.discount = config.discount
        self.gae_lambda = config.gae_lambda
        self.mask_no_eos_with_zero = config.mask_no_eos_with_zero
        self.temperature = config.temperature
        self.dynamic_sampling = config.dynamic_sampling

    @torch.no_grad()
    def compute_logp(self, data: Dict[str, Any], temperature: Optional[float] = None) -> torch.Tensor:
        """logp[b, t] = log pi(input_ids[b, t+1] | input_ids[b, :t+1]) padded to [B, S]."""
        temp = temperature or 1.0

        def calc(proxy: LogitsProxy, mb):
            labels = torch.roll(mb["input_ids"], shifts=-1, dims=-1)
            return proxy.logprobs(labels, temp)

        self.engine.eval()
        return self.engine.forward(input_=data, post_hook=calc, aggregate_fn=lambda xs: torch.cat(xs, dim=-1))

    def compute_advantages(self, data: Dict[str, Any]) -> None:
        cfg = self.config
        bs, max_seqlen = data["input_ids"].shape
        dev = data["input_ids"].device
        idx = torch.arange(bs, device=dev)
        if cfg.overlong_reward_penalty:
            reward_overlong_penalty(data, overlong_tokens=cfg.overlong_tokens,
                                    overlong_penalty_factor=cfg.overlong_penalty_factor,
                                    max_response_length=cfg.max_new_tokens)
        reward_score = (data["rewards"].float() + self.reward_bias) * self.reward_scaling
        reward_score = reward_score.clamp(-self.reward_clip, self.reward_clip)
        if self.group_reward_norm:
            n = bs // self.group_size
            r = reward_score[: n * self.group_size].view(n, self.group_size)
            r = (r - r.mean(-1, keepdim=True)) / (r.std(-1, keepdim=True) + 1e-9)
            reward_score = torch.cat([r.reshape(-1), reward_score[n * self.group_size:]])
        loss_mask = torch.roll(data["loss_mask"].float(), shifts=-1, dims=-1)
        if not cfg.use_decoupled_loss and cfg.recompute_logprob:
            old_logp = data["logprobs"] = data["prox_logp"]
        else:
            old_logp = torch.roll(data["logprobs"], shifts=-1, dims=-1)
            if not cfg.use_decoupled_loss:
                data["prox_logp"] = old_logp
        ref_logp = data.get("ref_logp", torch.zeros_like(old_logp))
        ref_logp = ref_logp * loss_mask
        old_logp = old_logp * loss_mask
        attn_mask = data["attention_mask"]
        seqlens = attn_mask.sum(-1).long()
        # truncated sequences: flagged explicitly by the worker runtime, else "filled the whole row"
        seq_no_eos_mask = (data["seq_no_eos_mask"].bool() if "seq_no_eos_mask" in data
                           else seqlens == attn_mask.shape[1])
        rewards = -self.kl_ctl * (old_logp - ref_logp)
        kl_rewards = rewards.clone()
        rewards[idx, seqlens - 1] = 0
        last = torch.clip(seqlens - 2, min=0)
        if self.mask_no_eos_with_zero:
            rewards[idx, last] += torch.where(seq_no_eos_mask, 0, reward_score)
        else:
            rewards[idx, last] += reward_score
        values = data["values"].float() if "values" in data else torch.zeros_like(rewards)
        advantages, returns = ops.gae_padded(rewards, values, seq_no_eos_mask, self.discount, self.gae_lambda)
        if self.adv_norm is not None:
            advantages = self.adv_norm(advantages, loss_mask, reduce_group=getattr(self.engine, "data_parallel_group",
                                                                                  None))
        data["advantages"] = advantages
        data["returns"] = returns
        data["kl_rewards"] = kl_rewards
        data["tot_rewards"] = rewards
        data["loss_mask"] = loss_mask
        data["logprobs"] = old_logp

    def fuses_proximal_logp(self) -> bool:
        c = self.config
        return bool(c.fuse_proximal_logp and c.use_decoupled_loss and c.ppo_n_minibatches == 1)

    def ppo_update(self, data: Dict[str, Any]) -> List[Dict[str, float]]:
        cfg = self.config
        if self.dynamic_sampling and len(data["rewards"]) % self.group_size == 0:
            data, sampling_stat = dynamic_sampling(data, self.group_size)
            stats_tracker.scalar(**sampling_stat)
        attn_mask = data["attention_mask"]
        loss_mask = data["loss_mask"]
        reward_score = data["rewards"]
        seqlens = attn_mask.sum(-1)
        all_stats = []
        result_den = {"correct_n_seqs": (reward_score > 0).bool(), "incorrect_n_seqs": (reward_score <= 0).bool()}
        if cfg.log_agent_stats:
            assert "begin_of_trajectory" in data, "'begin_of_trajectory' is expected to log agent statistics"
            assert len(cfg.log_agent_stats_keys) > 0
            result_den["agent"] = (data["begin_of_trajectory"] > 0).bool()
        global_den = dict(n_seqs=torch.ones_like(reward_score, dtype=torch.bool),
                          n_tokens=torch.ones_like(loss_mask, dtype=torch.bool), n_valid_tokens=loss_mask.bool(),
                          **result_den)
        stats_tracker.denominator(**global_den)
        stats_tracker.stat(correct_seq_len=seqlens.float(), denominator="correct_n_seqs")
        stats_tracker.stat(incorrect_seq_len=seqlens.float(), denominator="incorrect_n_seqs")
        stats_tracker.stat(advantages=data["advantages"].float(), kl_rewards=data["kl_rewards"].float(),
                           final_reward=data["tot_rewards"].float(), denominator="n_valid_tokens")
        prompt_lens = attn_mask.sum(-1) - data["loss_mask"].sum(-1)
        stats_tracker.stat(no_eos_ratios=(seqlens == attn_mask.shape[-1]).float(), task_reward=reward_score.float(),
                           prompt_len=prompt_lens.float(), seq_len=seqlens.float(), denominator="n_seqs")
        scalars = dict(mask_no_eos_with_zero=cfg.mask_no_eos_with_zero, eps_clip=cfg.eps_clip,
                       use_dual_clip=int(cfg.c_clip is not None))
        if cfg.c_clip is not None:
            scalars["c_clip"] = cfg.c_clip
        if cfg.behav_imp_weight_cap is not None:
            scalars["behav_imp_weight_cap"] = cfg.behav_imp_weight_cap
        stats_tracker.scalar(**scalars)
        if cfg.log_agent_stats:
            stats_tracker.stat(**{k: data[k].float() for k in cfg.log_agent_stats_keys}, denominator="agent")
        dp_group = getattr(self.engine, "data_parallel_group", None)
        global_stats = stats_tracker.export(reduce_group=dp_group)
        for k in global_den:
            for k2 in list(global_stats.keys()):
                if k2.endswith(k):
                    global_stats.pop(k2)
        for key in ["rewards", "tot_rewards", "kl_rewards", "versions", "returns", "begin_of_trajectory",
                    *(cfg.log_agent_stats_keys if cfg.log_agent_stats else [])]:
            data.pop(key, None)
        self.engine.train()
        mb_inputs = split_padded_tensor_dict_into_mb_list(data, MicroBatchSpec(n_mbs=cfg.ppo_n_minibatches))
        loss_fn = functools.partial(grpo_loss_fn, temperature=self.temperature, eps_clip=cfg.eps_clip,
                                    eps_clip_higher=cfg.eps_clip_higher, c_clip=cfg.c_clip,
                                    behav_imp_weight_cap=cfg.behav_imp_weight_cap)
        for mb in mb_inputs.mbs:
            train_stat = self.engine.train_batch(mb, loss_fn=loss_fn,
                                                 loss_weight_fn=lambda x: x["loss_mask"].count_nonzero())
            stats_tracker.scalar(**train_stat)
            all_stats.append(_pp_share(self.engine, stats_tracker.export(reduce_group=dp_group)))
        all_stats[0].update(global_stats)
        return all_stats


def grpo_loss_fn(logits, input_data: Dict, temperature: float, eps_clip: float, eps_clip_higher: Optional[float],
                 c_clip: Optional[float], behav_imp_weight_cap: Optional[float]):
    """Actor loss on one packed micro-batch. `logits` is a ``LogitsProxy`` (fused path) or a dense tensor."""
    input_ids = input_data["input_ids"]
    old_logp = input_data["logprobs"]
    advantages = input_data["advantages"]
    loss_mask = input_data["loss_mask"].bool()
    labels = torch.roll(input_ids, shifts=-1, dims=-1)
    if isinstance(logits, LogitsProxy):
        logprobs, entropy = logits.logprobs_entropy(labels, temperature)
        n_rows = logits.shape[0]
    else:
        logprobs, entropy = ops.gather_logprobs_entropy(logits, labels, temperature)
        n_rows = logits.shape[0]
    entropy = entropy.detach()
    # fused proximal policy (config.fuse_proximal_logp): with a single minibatch per update the proximal
    # policy IS the policy this forward pass evaluates, so its log-probs are the detached current ones
    prox_logp = input_data["prox_logp"] if "prox_logp" in input_data else logprobs.detach()
    loss, stat = ppo_actor_loss_fn(logprobs=logprobs, old_logprobs=old_logp, advantages=advantages, eps_clip=eps_clip,
                                   eps_clip_higher=eps_clip_higher, loss_mask=loss_mask, c_clip=c_clip,
                                   proximal_logprobs=prox_logp, behav_imp_weight_cap=behav_imp_weight_cap)
    stats_tracker.denominator(n_tokens=torch.ones(n_rows, dtype=torch.bool, device=logprobs.device),
                              n_valid_tokens=loss_mask, clipped_tokens=stat["clip_mask"],
                              dual_clipped_tokens=stat["dual_clip_mask"])
    stats_tracker.stat(importance_weight=stat["importance_weight"].float(), approx_kl=stat["approx_kl"].float(),
                       new_logp=logprobs.detach().float(), old_logp=old_logp.float(), entropy=entropy.float(),
                       actor_loss=stat["loss"].float(), clip_ratio=stat["clip_mask"].float(),
                       dual_clip_ratio=stat["dual_clip_mask"].float(), denominator="n_valid_tokens")
    stats_tracker.denominator(unclipped_behave_tokens=stat["behave_mask"])
    stats_tracker.stat(behave_imp_weight=stat["behave_imp_weight"].float(),
                       behave_approx_kl=stat["behave_approx_kl"].float(), denominator="unclipped_behave_tokens")
    if not isinstance(logits, LogitsProxy):
        stats_tracker.stat(vocab_min_logits=logits.detach().min(-1).values.float(),
                           vocab_max_logits=logits.detach().max(-1).values.float(), denominator="n_tokens")
    clip_mask = stat["clip_mask"]
    stats_tracker.stat(clipped_new_logp=torch.where(clip_mask, logprobs.detach().float(), 0.0),
                       clipped_old_logp=torch.where(clip_mask, old_logp.float(), 0.0), denominator="clipped_tokens")
    return loss


class NativePPOActor(NativeTrainEngine):
    """Train engine + PPO actor algorithm (the reference's ``FSDPPPOActor``, ``actor.py:274-289``)."""

    def __init__(self, config: PPOActorConfig):
        super().__init__(config)
        self.actor = PPOActor(config, self)

    @torch.no_grad()
    def compute_logp(self, *args, **kwargs):
        return self.actor.compute_logp(*args, **kwargs)

    @torch.no_grad()
    def compute_advantages(self, *args, **kwargs) -> None:
        self.actor.compute_advantages(*args, **kwargs)

    def ppo_update(self, *args, **kwargs) -> List[Dict[str, float]]:
        return self.actor.ppo_update(*args, **kwargs)


FSDPPPOActor = NativePPOActor
MegatronPPOActor = NativePPOActor
