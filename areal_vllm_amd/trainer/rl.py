"""GRPO / PPO training loop shared by the RL example scripts.

One SPMD process per train GPU. Per step: collect a batch (asynchronously with bounded staleness, or
synchronously), recompute proximal log-probs, reference log-probs and critic values when configured,
compute advantages, update, then pause rollouts -> push weights -> bump versions -> save / evaluate /
write recover info -> log -> resume. The control flow is the reference's
``examples/math/gsm8k_grpo.py:33-295``; it is a library function here so that every example (math,
multi-turn, vision, PPO with a critic) reuses one loop.

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
from ..api.io_struct import FinetuneSpec, StepInfo, WeightUpdateMeta
from ..dataset import get_custom_dataset
from ..engine.ppo.actor import NativePPOActor
from ..platforms import current_platform
from ..utils import logging, seeding, stats_tracker
from ..utils.data import broadcast_tensor_container, to_device
from ..utils.dataloader import StatefulDataLoader
from ..utils.evaluator import Evaluator
from ..utils.recover import RecoverHandler
from ..utils.saver import Saver
from ..utils.stats_logger import StatsLogger

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


def run_grpo(argv, reward_fn: Optional[Callable] = None,
             workflow_builder: Optional[Callable[[Any, Any, Callable, str], Tuple[Any, Any]]] = None,
             config_cls=GRPOConfig, load_tokenizer: bool = True):
    """`workflow_builder(config, tokenizer, reward_fn, log_dir) -> (train_workflow, eval_workflow)`;
    defaults to single-turn RLVR."""
    config, _ = load_expr_config(argv, config_cls)
    rank = int(os.getenv("RANK", "0"))
    tokenizer = None
    if load_tokenizer and config.tokenizer_path and os.path.isdir(config.tokenizer_path):
        from ..utils.hf_utils import load_hf_tokenizer

        tokenizer = load_hf_tokenizer(config.tokenizer_path)
    seeding.set_random_seed(config.seed, key=f"trainer{rank}")
    allocation_mode = AllocationMode.from_str(config.allocation_mode)
    parallel_strategy = allocation_mode.train

    actor = NativePPOActor(config=config.actor)
    actor.create_process_group(parallel_strategy=parallel_strategy)
    ds_kw = dict(rank=actor.data_parallel_rank, world_size=actor.data_parallel_world_size, tokenizer=tokenizer)
    train_dataset = get_custom_dataset(path=config.train_dataset.path, split="train",
                                       max_length=config.train_dataset.max_length, type=config.train_dataset.type or "rl",
                                       **ds_kw)
    valid_dataset = None
    if config.valid_dataset is not None and config.valid_dataset.path:
        valid_dataset = get_custom_dataset(path=config.valid_dataset.path, split="test",
                                           max_length=config.valid_dataset.max_length,
                                           type=config.valid_dataset.type or "rl", **ds_kw)
    dp = actor.data_parallel_world_size
    train_dataloader = StatefulDataLoader(train_dataset, batch_size=max(1, config.train_dataset.batch_size // dp),
                                          shuffle=config.train_dataset.shuffle, drop_last=config.train_dataset.drop_last,
                                          seed=config.seed)
    valid_dataloader = None
    if valid_dataset is not None:
        valid_dataloader = StatefulDataLoader(valid_dataset, batch_size=max(1, config.valid_dataset.batch_size // dp),
                                              shuffle=False, drop_last=config.valid_dataset.drop_last)
    ft_spec = FinetuneSpec(total_train_epochs=config.total_train_epochs,
                           dataset_size=len(train_dataloader) * config.train_dataset.batch_size,
                           train_batch_size=config.train_dataset.batch_size)

    actor.initialize(None, ft_spec)
    ref = None
    if config.actor.kl_ctl > 0 and config.ref is not None and config.ref.path:
        ref = NativePPOActor(config=config.ref)
        ref.create_process_group(parallel_strategy=parallel_strategy)
        ref.initialize(None, ft_spec)
    critic = None
    if isinstance(config, PPOConfig) and config.critic.path:
        from ..engine.ppo.critic import NativePPOCritic

        critic = NativePPOCritic(config=config.critic)
        critic.create_process_group(parallel_strategy=parallel_strategy)
        critic.initialize(None, ft_spec)

    rollout, eval_rollout, weight_update_meta, local_engine = build_rollout_engines(config, actor, parallel_strategy)

    if reward_fn is None:
        from ..reward import get_custom_reward_fn

        reward_fn = get_custom_reward_fn(config.train_dataset.reward_fn or config.train_dataset.path)
    if tokenizer is not None:
        for t in (tokenizer.pad_token_id, tokenizer.eos_token_id):
            if t is not None and t not in config.gconfig.stop_token_ids:
                config.gconfig.stop_token_ids.append(t)
    log_dir = StatsLogger.get_log_path(config.stats_logger)
    if workflow_builder is None:
        from ..workflow.rlvr import RLVRWorkflow

        in_proc = config.train_dataset.path.startswith("synthetic:")
        workflow = RLVRWorkflow(reward_fn, config.gconfig, tokenizer, dump_dir=os.path.join(log_dir, "generated"),
                                reward_in_process=in_proc)
        eval_workflow = RLVRWorkflow(reward_fn, config.gconfig.new(temperature=0.6), tokenizer,
                                     rollout_stat_scope="eval-rollout", reward_in_process=in_proc,
                                     dump_dir=os.path.join(log_dir, "generated-eval"))
    else:
        workflow, eval_workflow = workflow_builder(config, tokenizer, reward_fn, log_dir)

    saver = Saver(config.saver, ft_spec)
    stats_logger = StatsLogger(config.stats_logger, ft_spec)
    evaluator = Evaluator(config.evaluator, ft_spec)
    recover_handler = RecoverHandler(config.recover, ft_spec)
    recover_info = recover_handler.load(actor, saver, evaluator, stats_logger, train_dataloader,
                                        inference_engine=rollout, weight_update_meta=weight_update_meta)
    start_step = recover_info.last_step_info.next().global_step if recover_info is not None else 0

    steps_per_epoch = len(train_dataloader)
    max_steps = config.total_train_epochs * steps_per_epoch
    if config.total_train_steps is not None:
        max_steps = min(max_steps, config.total_train_steps)
    data_generator = itertools.cycle(train_dataloader)

    def sync():
        if dist.is_initialized():
            dist.barrier()
        current_platform.synchronize()

    for global_step in range(start_step, max_steps):
        epoch, step = divmod(global_step, steps_per_epoch)
        step_info = StepInfo(global_step=global_step, epoch=epoch, epoch_step=step, steps_per_epoch=steps_per_epoch)

        with stats_tracker.record_timing("rollout"):
            batch = None
            if actor.is_data_parallel_head():
                if config.async_training:
                    batch = rollout.prepare_batch(train_dataloader, workflow=workflow)
                else:
                    batch = rollout.rollout_batch(next(data_generator), workflow=workflow)
                batch = to_device(batch, actor.device)
            batch = broadcast_tensor_container(batch, src_rank=actor.current_data_parallel_head(),
                                               group=actor.context_and_model_parallel_group)
        sync()

        fused_prox = (config.actor.fuse_proximal_logp and config.actor.use_decoupled_loss
                      and config.actor.ppo_n_minibatches == 1)
        if (config.actor.recompute_logprob or config.actor.use_decoupled_loss) and not fused_prox:
            with stats_tracker.record_timing("recompute_logp"):
                batch["prox_logp"] = actor.compute_logp(batch)
        if ref is not None:
            with stats_tracker.record_timing("ref_logp"):
                batch["ref_logp"] = ref.compute_logp(batch)
        if critic is not None:
            with stats_tracker.record_timing("critic_values"):
                batch["values"] = critic.compute_values(batch)
        with stats_tracker.record_timing("compute_advantage"):
            actor.compute_advantages(batch)
        critic_stats = []
        if critic is not None:
            with stats_tracker.record_timing("critic_train_step"), stats_tracker.scope("ppo_critic"):
                critic_stats = critic.ppo_update(batch)
                critic.step_lr_scheduler()
        with stats_tracker.record_timing("train_step"), stats_tracker.scope("grpo_actor"):
            stats = actor.ppo_update(batch)
            actor.step_lr_scheduler()
        if critic_stats:
            stats[0].update(critic_stats[0])

        if rollout is not None:
            rollout.pause()
        with stats_tracker.record_timing("update_weights"):
            future = None
            if rollout is not None and (weight_update_meta.type == "inproc" or rank == 0):
                future = rollout.update_weights(weight_update_meta)
            actor.upload_weights(weight_update_meta)
            if future is not None:
                future.result()
            sync()
            actor.set_version(global_step + 1)
            if rollout is not None:
                rollout.set_version(global_step + 1)
                eval_rollout.set_version(global_step + 1)

        with stats_tracker.record_timing("save"):
            saver.save(actor, epoch, step, global_step, tokenizer=tokenizer)

        with stats_tracker.record_timing("eval"):
            def evaluate_fn():
                if valid_dataloader is None or eval_rollout is None:
                    return
                eval_rollout.resume()
                cnt = 0
                for data in valid_dataloader:
                    for item in data:
                        eval_rollout.submit(item, eval_workflow)
                        cnt += 1
                eval_rollout.wait(cnt, timeout=None)
                eval_rollout.pause()

            evaluator.evaluate(evaluate_fn, epoch, step, global_step)

        with stats_tracker.record_timing("checkpoint_for_recover"):
            recover_handler.dump(actor, step_info, saver, evaluator, stats_logger, train_dataloader,
                                 tokenizer=tokenizer)
        sync()
        stats[0].update(actor.share_pipeline_stats(stats_tracker.export_all(reduce_group=actor.data_parallel_group)))
        stats_logger.commit(epoch, step, global_step, stats)
        sync()
        if rollout is not None:
            rollout.resume()

    stats_logger.close()
    if rollout is not None:
        eval_rollout.destroy()
        rollout.destroy()
    if local_engine is not None:
        local_engine.destroy()
    for e in (ref, critic, actor):
        if e is not None:
            e.destroy()
