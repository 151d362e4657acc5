"""PPO critic: value inference and the clipped value-loss update on the native train engine.

Counterpart of the reference's critic (``areal/engine/ppo/critic.py``, value loss from
``realhf/impl/model/utils/ppo_functional.py:161-226``): the critic shares the transformer body and ends
in a 1-unit value head (``ModelConfig.is_critic``); ``compute_values`` returns ``[B, S]`` values,
``ppo_update`` regresses them onto the GAE returns with value clipping around the rollout-time values.

Call order within a step: ``values = critic.compute_values(batch)`` -> ``actor.compute_advantages(batch)``
(reads ``batch["values"]``, writes ``returns``) -> ``critic.ppo_update(batch)`` -> ``actor.ppo_update(batch)``
(the actor update drops ``returns`` from the batch).
"""
import functools
from typing import Any, Dict, List

import torch

from ...api.cli_args import MicroBatchSpec, PPOCriticConfig
from ...api.engine_api import TrainEngine
from ...utils import stats_tracker
from ...utils.data import split_padded_tensor_dict_into_mb_list
from ...utils.functional import ppo_critic_loss_fn
from ..train_engine import NativeTrainEngine


def _pp_share(engine, stats):
    """Loss statistics are recorded on the last pipeline stage; give every stage the union."""
    f = getattr(engine, "share_pipeline_stats", None)
    return f(stats) if f is not None else stats


class PPOCritic:
    def __init__(self, config: PPOCriticConfig, engine: TrainEngine):
        self.config = config
        self.engine = engine

    @torch.no_grad()
    def compute_values(self, data: Dict[str, Any]) -> torch.Tensor:
        self.engine.eval()
        return self.engine.forward(input_=data, post_hook=lambda proxy, mb: proxy.values(),
                                   aggregate_fn=lambda xs: torch.cat(xs, dim=-1))

    def ppo_update(self, data: Dict[str, Any]) -> List[Dict[str, float]]:
        cfg = self.config
        keep = ("input_ids", "attention_mask", "loss_mask", "values", "returns")
        batch = {k: data[k] for k in keep}
        stats_tracker.denominator(n_valid_tokens=batch["loss_mask"].bool())
        stats_tracker.stat(returns=batch["returns"].float(), old_values=batch["values"].float(),
                           denominator="n_valid_tokens")
        dp_group = getattr(self.engine, "data_parallel_group", None)
        global_stats = stats_tracker.export(reduce_group=dp_group)
        global_stats.pop("n_valid_tokens", None)
        self.engine.train()
        mbs = split_padded_tensor_dict_into_mb_list(batch, MicroBatchSpec(n_mbs=cfg.ppo_n_minibatches))
        loss_fn = functools.partial(critic_loss_fn, value_eps_clip=cfg.value_eps_clip,
                                    loss_fn_type=cfg.value_loss_type)
        all_stats = []
        for mb in mbs.mbs:
            stat = self.engine.train_batch(mb, loss_fn=loss_fn, loss_weight_fn=lambda x: x["loss_mask"].count_nonzero())
            stats_tracker.scalar(**stat)
            all_stats.append(_pp_share(self.engine, stats_tracker.export(reduce_group=dp_group)))
        all_stats[0].update(global_stats)
        return all_stats


def critic_loss_fn(proxy, input_data: Dict[str, Any], value_eps_clip: float, loss_fn_type: str = "mse"):
    """`proxy.values()` are the packed value predictions; targets / old values / mask are packed [T]."""
    values = proxy.values()
    loss_mask = input_data["loss_mask"].bool()
    loss, stat = ppo_critic_loss_fn(value=values, old_value=input_data["values"].float(),
                                    target_value=input_data["returns"].float(), value_eps_clip=value_eps_clip,
                                    loss_mask=loss_mask, loss_fn_type=loss_fn_type)
    stats_tracker.denominator(n_tokens=torch.ones_like(loss_mask), n_valid_tokens=loss_mask,
                              clipped_tokens=stat["clip_mask"])
    stats_tracker.stat(value_loss=stat["loss"].float(), new_values=values.detach().float(),
                       denominator="n_valid_tokens")
    return loss


class NativePPOCritic(NativeTrainEngine):
    def __init__(self, config: PPOCriticConfig):
        super().__init__(config)
        self.is_critic = True
        self.critic = PPOCritic(config, self)

    @torch.no_grad()
    def compute_values(self, *args, **kwargs) -> torch.Tensor:
        return self.critic.compute_values(*args, **kwargs)

    def ppo_update(self, *args, **kwargs) -> List[Dict[str, float]]:
        return self.critic.ppo_update(*args, **kwargs)


FSDPPPOCritic = NativePPOCritic  # name used by scripts written against the reference API
