"""GRPO / PPO trainer shared by the RL example scripts.

One SPMD process per train GPU runs an `RLTrainer`. A training step is five phases, each a method:

1. ``collect``       a batch of finished rollouts (asynchronously with bounded staleness, or synchronously),
                     handed from the data-parallel head to the ranks of its replica;
2. ``score``         proximal / reference log-probs and critic values where configured, then advantages;
3. ``update``        critic and actor mini-batch updates, learning-rate schedules;
4. ``publish``       rollouts paused -> weights pushed (in-process copy, peer-memory pull, collective push or
                     files) -> versions bumped; generation RESUMES here unless an evaluation is due;
5. ``housekeeping``  periodic checkpoint, evaluation, recover checkpoint, statistics - with the servers already
                     generating the next batches underneath (the reference keeps them paused until the step's
                     log line is written, ``examples/math/gsm8k_grpo.py:218-292``; a checkpoint of a 7B model
                     is seconds of idle rollout GPUs per save).

What a step computes and the order of its side effects on versions follow the reference's
``examples/math/gsm8k_grpo.py:33-295``; every example (math, multi-turn, vision, PPO with a critic, search
agent) drives this one class through `run_grpo`.

Rollout deployment is chosen from the launch environment:

* ``AREAL_LLM_SERVER_ADDRS`` set (decoupled allocation): HTTP client to native rollout servers, weight
  updates by collective broadcast (``nccl``/``xgmi``) or disk.
* otherwise (colocated): every data-parallel head embeds a rollout engine on its own GPU and weights
  move by one device-to-device copy of the flat parameter buffer.
"""
import itertools
import os
from copy import deepcopy
from typing import Any, Callable, Dict, Optional, Tuple

import torch
import torch.distributed as dist

from ..api.alloc_mode import AllocationMode
from ..api.cli_args import GRPOConfig, InferenceEngineConfig, PPOConfig, load_expr_config
from ..api.io_struct import StepInfo, WeightUpdateMeta
from ..engine.ppo.actor import NativePPOActor
from ..utils import logging, seeding, stats_tracker
from ..utils.data import broadcast_tensor_container, to_device
from ..utils.stats_logger import StatsLogger
from .common import PeriodicActions, StepBudget, build_dataloaders, sync_ranks

logger = logging.getLogger("rl-trainer")


def build_rollout_engines(config: GRPOConfig, actor: NativePPOActor, parallel_strategy):
    """-> (rollout, eval_rollout, weight_update_meta, colocated_engine | None)."""
    eval_cfg: InferenceEngineConfig = deepcopy(config.rollout)
    eval_cfg.max_head_offpolicyness = int(1e12)  # evaluation has no staleness control
    if os.getenv("AREAL_LLM_SERVER_ADDRS"):
        from ..engine.remote_engine import RemoteInferenceEngine

        rollout = RemoteInferenceEngine(config.rollout)
        rollout.initialize(train_data_parallel_size=parallel_strategy.dp_size)
        eval_rollout = RemoteInferenceEngine(eval_cfg)
        eval_rollout.initialize()
        am = AllocationMode.from_str(config.allocation_mode)
        transport = config.rollout_weight_update
        moe = getattr(actor.model_config, "n_experts", 0) > 0
        if transport in ("nccl", "xgmi") and parallel_strategy.pp_size > 1:
            # a pipeline stage only holds its own layers: the collective push (whose senders need every
            # parameter) would update the servers partially. The peer-memory pull has no such limit (every server
            # rank reads the ranges it needs from whichever trainer rank owns them); checkpoint files carry all
            # stages as well
            logger.warning("trainer has pipeline stages: the collective weight push does not cover them, using "
                           + ("checkpoint files" if moe else "the peer-memory pull"))
            transport = "disk" if moe else "ipc"
        if transport == "ipc" and moe:
            logger.warning("peer-memory weight updates cover dense models; using "
                           + ("checkpoint files" if parallel_strategy.pp_size > 1 else "the collective push"))
            transport = "disk" if parallel_strategy.pp_size > 1 else "xgmi"
        if transport == "disk":
            meta = WeightUpdateMeta.from_disk(config.experiment_name, config.trial_name, config.cluster.fileroot)
        elif transport == "ipc":
            meta = WeightUpdateMeta.from_ipc(actor)
        else:
            # "xgmi": every data-parallel trainer rank pushes its share of the chunks through its own links;
            # "nccl": the reference's protocol, everything leaves through rank 0
            senders = 1
            if transport == "xgmi":
                senders = min(8, dist.get_world_size())
            box = [WeightUpdateMeta.from_fsdp_nccl(am, actor, n_senders=senders)]
            dist.broadcast_object_list(box, src=0)
            meta = box[0]
        return rollout, eval_rollout, meta, None
    from ..engine.local_engine import LocalInferenceEngine
    from ..inference.engine import NativeInferenceEngine

    scfg = deepcopy(config.sglang)
    scfg.model_path = scfg.model_path or config.actor.path
    scfg.random_seed = scfg.random_seed + actor.data_parallel_rank
    scfg.dtype = config.actor.dtype
    engine = None
    if actor.is_data_parallel_head():
        engine = NativeInferenceEngine(scfg, model_config=actor.model_config, device=actor.device,
                                       kv_cache_gb=config.colocated_kv_cache_gb)
        engine.initialize(load_weights=False)
        actor.connect_local_engine(engine)
    rollout = LocalInferenceEngine(config.rollout, engine) if engine is not None else None
    eval_rollout = LocalInferenceEngine(eval_cfg, engine) if engine is not None else None
    if rollout is not None:
        rollout.initialize(train_data_parallel_size=parallel_strategy.dp_size)
        eval_rollout.initialize()
    return rollout, eval_rollout, WeightUpdateMeta.from_inproc(), engine


class RLTrainer:
    """Owns the engines, data and periodic actions of one trainer rank; `run()` steps until the budget is spent."""

    def __init__(self, argv, reward_fn: Optional[Callable] = None,
                 workflow_builder: Optional[Callable[[Any, Any, Callable, str], Tuple[Any, Any]]] = None,
                 config_cls=GRPOConfig, load_tokenizer: bool = True):
        self.config, _ = load_expr_config(argv, config_cls)
        config = self.config
        self.rank = int(os.getenv("RANK", "0"))
        self.tokenizer = None
        if load_tokenizer and config.tokenizer_path and os.path.isdir(config.tokenizer_path):
            from ..utils.hf_utils import load_hf_tokenizer

            self.tokenizer = load_hf_tokenizer(config.tokenizer_path)
        seeding.set_random_seed(config.seed, key=f"trainer{self.rank}")
        self.parallel_strategy = AllocationMode.from_str(config.allocation_mode).train

        self.actor = NativePPOActor(config=config.actor)
        self.actor.create_process_group(parallel_strategy=self.parallel_strategy)
        self.train_dataloader, self.valid_dataloader, self.ft_spec = build_dataloaders(config, self.actor, "rl",
                                                                                       self.tokenizer)
        self._build_models()
        self.rollout, self.eval_rollout, self.weight_update_meta, self.local_engine = build_rollout_engines(
            config, self.actor, self.parallel_strategy)
        self._build_workflows(reward_fn, workflow_builder)
        self.periodic = PeriodicActions(config, self.ft_spec)
        recovered = self.periodic.load(self.actor, self.train_dataloader, inference_engine=self.rollout,
                                       weight_update_meta=self.weight_update_meta)
        self.budget = StepBudget(config, len(self.train_dataloader), recovered)
        self._sync_batches = itertools.cycle(self.train_dataloader)
        self._generating = True  # rollout engines start un-paused

    # ------------------------------------------------------------------------------ set-up
    def _build_models(self):
        config = self.config
        self.actor.initialize(None, self.ft_spec)
        self.ref = None
        if config.actor.kl_ctl > 0 and config.ref is not None and config.ref.path:
            self.ref = NativePPOActor(config=config.ref)
            self.ref.create_process_group(parallel_strategy=self.parallel_strategy)
            self.ref.initialize(None, self.ft_spec)
        self.critic = None
        if isinstance(config, PPOConfig) and config.critic.path:
            from ..engine.ppo.critic import NativePPOCritic

            self.critic = NativePPOCritic(config=config.critic)
            self.critic.create_process_group(parallel_strategy=self.parallel_strategy)
            self.critic.initialize(None, self.ft_spec)

    def _build_workflows(self, reward_fn, workflow_builder):
        config, tokenizer = self.config, self.tokenizer
        if reward_fn is None:
            from ..reward import get_custom_reward_fn

            reward_fn = get_custom_reward_fn(config.train_dataset.reward_fn or config.train_dataset.path)
        if tokenizer is not None:
            for t in (tokenizer.pad_token_id, tokenizer.eos_token_id):
                if t is not None and t not in config.gconfig.stop_token_ids:
                    config.gconfig.stop_token_ids.append(t)
        log_dir = StatsLogger.get_log_path(config.stats_logger)
        if workflow_builder is not None:
            self.workflow, self.eval_workflow = workflow_builder(config, tokenizer, reward_fn, log_dir)
            return
        from ..workflow.rlvr import RLVRWorkflow

        in_proc = config.train_dataset.path.startswith("synthetic:")
        self.workflow = RLVRWorkflow(reward_fn, config.gconfig, tokenizer, dump_dir=os.path.join(log_dir, "generated"),
                                     reward_in_process=in_proc)
        self.eval_workflow = RLVRWorkflow(reward_fn, config.gconfig.new(temperature=0.6), tokenizer,
                                          rollout_stat_scope="eval-rollout", reward_in_process=in_proc,
                                          dump_dir=os.path.join(log_dir, "generated-eval"))

    # ------------------------------------------------------------------------------ helpers
    def _pause_generation(self):
        if self.rollout is not None and self._generating:
            self.rollout.pause()
        self._generating = False

    def _resume_generation(self):
        if self.rollout is not None and not self._generating:
            self.rollout.resume()
        self._generating = True

    # ------------------------------------------------------------------------------ phases
    def collect(self) -> Dict[str, Any]:
        """The data-parallel head of every replica takes a batch from its rollout engine; the other ranks of the
        replica (tensor / pipeline / context parallel) receive it."""
        actor = self.actor
        with stats_tracker.record_timing("rollout"):
            batch = None
            if actor.is_data_parallel_head():
                if self.config.async_training:
                    batch = self.rollout.prepare_batch(self.train_dataloader, workflow=self.workflow)
                else:
                    batch = self.rollout.rollout_batch(next(self._sync_batches), workflow=self.workflow)
                batch = to_device(batch, actor.device)
            batch = broadcast_tensor_container(batch, src_rank=actor.current_data_parallel_head(),
                                               group=actor.context_and_model_parallel_group)
        sync_ranks()
        return batch

    def score(self, batch: Dict[str, Any]) -> None:
        """Everything the loss needs that is not a gradient: log-probs under the current policy (the proximal
        policy of the decoupled objective; skipped when the update computes it in its own forward), under the
        reference policy, critic values, and the advantages made from them."""
        ac = self.config.actor
        fused_prox = ac.fuse_proximal_logp and ac.use_decoupled_loss and ac.ppo_n_minibatches == 1
        if (ac.recompute_logprob or ac.use_decoupled_loss) and not fused_prox:
            with stats_tracker.record_timing("recompute_logp"):
                batch["prox_logp"] = self.actor.compute_logp(batch)
        if self.ref is not None:
            with stats_tracker.record_timing("ref_logp"):
                batch["ref_logp"] = self.ref.compute_logp(batch)
        if self.critic is not None:
            with stats_tracker.record_timing("critic_values"):
                batch["values"] = self.critic.compute_values(batch)
        with stats_tracker.record_timing("compute_advantage"):
            self.actor.compute_advantages(batch)

    def update(self, batch: Dict[str, Any]):
        critic_stats = []
        if self.critic is not None:
            with stats_tracker.record_timing("critic_train_step"), stats_tracker.scope("ppo_critic"):
                critic_stats = self.critic.ppo_update(batch)
                self.critic.step_lr_scheduler()
        with stats_tracker.record_timing("train_step"), stats_tracker.scope("grpo_actor"):
            stats = self.actor.ppo_update(batch)
            self.actor.step_lr_scheduler()
        if critic_stats:
            stats[0].update(critic_stats[0])
        return stats

    def publish(self, version: int) -> None:
        """New weights to the rollout engines, under a pause; every engine then stamps tokens with `version`."""
        meta = self.weight_update_meta
        self._pause_generation()
        with stats_tracker.record_timing("update_weights"):
            pending = None
            if self.rollout is not None and (meta.type == "inproc" or self.rank == 0):
                pending = self.rollout.update_weights(meta)  # the servers' side of the hand-over
            self.actor.upload_weights(meta)  # the trainer's side
            if pending is not None:
                pending.result()
            sync_ranks()
            self.actor.set_version(version)
            if self.rollout is not None:
                self.rollout.set_version(version)
                self.eval_rollout.set_version(version)

    def _evaluate(self):
        if self.valid_dataloader is None or self.eval_rollout is None:
            return
        self.eval_rollout.resume()
        n = 0
        for data in self.valid_dataloader:
            for item in data:
                self.eval_rollout.submit(item, self.eval_workflow)
                n += 1
        self.eval_rollout.wait(n, timeout=None)
        self.eval_rollout.pause()

    def housekeeping(self, info: StepInfo, stats) -> None:
        """Checkpoint / evaluation / recover checkpoint / statistics. Training rollouts resume before any of it
        unless an evaluation is due (its prompts have the engines to themselves), and at the end otherwise."""
        self.periodic.after_step(self.actor, info, stats, self.train_dataloader, evaluate_fn=self._evaluate,
                                 on_no_eval=self._resume_generation, tokenizer=self.tokenizer)
        self._resume_generation()

    # ------------------------------------------------------------------------------ driver
    def step(self, info: StepInfo) -> None:
        batch = self.collect()
        self.score(batch)
        stats = self.update(batch)
        self.publish(info.global_step + 1)
        self.housekeeping(info, stats)

    def run(self) -> None:
        for info in self.budget:
            self.step(info)

    def close(self) -> None:
        self.periodic.close()
        if self.rollout is not None:
            self.eval_rollout.destroy()
            self.rollout.destroy()
        if self.local_engine is not None:
            self.local_engine.destroy()
        for e in (self.ref, self.critic, self.actor):
            if e is not None:
                e.destroy()


def run_grpo(argv, reward_fn: Optional[Callable] = None,
             workflow_builder: Optional[Callable[[Any, Any, Callable, str], Tuple[Any, Any]]] = None,
             config_cls=GRPOConfig, load_tokenizer: bool = True):
    """`workflow_builder(config, tokenizer, reward_fn, log_dir) -> (train_workflow, eval_workflow)`;
    defaults to single-turn RLVR."""
    trainer = RLTrainer(argv, reward_fn, workflow_builder, config_cls, load_tokenizer)
    trainer.run()
    trainer.close()
