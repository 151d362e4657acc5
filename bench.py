#!/usr/bin/env python3
"""Headline benchmark: end-to-end GRPO steps of Qwen2-1.5B (random init, synthetic prompts) on N MI355X.

One step = what the reference's ``examples/math/gsm8k_grpo.py`` loop does per iteration:
  rollout (native paged-KV engine: prompts x n_samples generations, max_new_tokens each, reward)
  -> recompute log-probs (proximal policy) -> advantages (GAE kernel + group norm)
  -> decoupled-PPO update (forward, backward, grad reduce-scatter, AdamW, param all-gather)
  -> weight update of the rollout engine + version bump.
Every rank owns one GPU and runs BOTH a rollout engine and a data-parallel trainer shard on it
(colocated allocation); per-GPU work is fixed as N grows (weak scaling): 32 prompts x 4 samples per GPU
per step, which at N = 8 is the reference config's 256 prompts x 4 samples per step.

Training is asynchronous with bounded staleness like the reference config (``async_training: true``,
``max_head_offpolicyness: 2``): the rollout engine keeps generating on its own HIP stream while the
trainer consumes the oldest finished batch, weight updates interrupt running generations (their KV is
recomputed with the new weights) and every token carries the version that produced it. ``--eta 0`` gives
the synchronous schedule. Before the W warm-up steps the rollout pipeline is primed with eta+1 untimed
steps so that the timed steps see the steady state (staggered batches), not the start-up transient in
which eta+1 batches finish together.

Launch: ``python bench.py`` (1 GPU) or
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W``.
Rank 0 prints ONE JSON line. `value` = RL samples (trajectories) per second over the whole job.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

# published comparator: ~20k tokens/s end-to-end for the 1.5B model on one 8xH800 node
# (reference blog/AReaL_v0_2.md:54, assets/thpt_comparison.png)
BASELINE_TOK_S_8GPU = 20000.0
# the same figure's 7B point: ~26k tokens/s on 32 H800, i.e. per GPU
BASELINE_TOK_S_PER_GPU = {"qwen2-1.5b": BASELINE_TOK_S_8GPU / 8.0, "qwen2.5-7b": 26000.0 / 32.0}
MODEL_LABEL = {"qwen2-1.5b": "Qwen2-1.5B", "qwen2.5-7b": "Qwen2.5-7B"}


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=6)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--model", default="qwen2-1.5b")
    p.add_argument("--prompts-per-gpu", type=int, default=32)
    p.add_argument("--n-samples", type=int, default=4)
    p.add_argument("--max-new-tokens", type=int, default=1024)
    p.add_argument("--prompt-len-min", type=int, default=64)
    p.add_argument("--prompt-len-max", type=int, default=192)
    p.add_argument("--max-tokens-per-mb", type=int, default=32768)
    p.add_argument("--decode-steps", type=int, default=8)
    p.add_argument("--kv-cache-gb", type=float, default=24.0)
    p.add_argument("--no-graph", action="store_true")
    p.add_argument("--eta", type=int, default=2, help="max_head_offpolicyness; 0 = synchronous RL")
    p.add_argument("--retain-kv", action="store_true",
                   help="keep the KV cache of running generations across weight updates (no re-prefill)")
    p.add_argument("--device", default="cuda", choices=["cuda", "cpu"],
                   help="cpu = gloo rehearsal of the multi-rank control flow with a tiny model (not a benchmark)")
    p.add_argument("--fuse-prox", action="store_true",
                   help="take proximal log-probs from the training forward (valid for ppo_n_minibatches=1) instead of "
                        "a separate recompute pass; off by default to keep the reference's per-step work")
    p.add_argument("--profile-phases", action="store_true", help="print per-phase timings on rank 0 (stderr)")
    return p.parse_args()


def main():
    args = parse_args()
    # the result line is the only thing that may reach stdout: libraries (gloo's connection banner) print
    # there too, so everything before the result goes to stderr
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("AREAL_LOG_LEVEL", "WARNING")
    on_gpu = args.device == "cuda"
    if on_gpu:
        assert torch.cuda.is_available(), "bench.py needs an MI355X"
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        os.environ["HIP_VISIBLE_DEVICES"] = os.environ["CUDA_VISIBLE_DEVICES"] = ""
        os.environ["AREAL_PLATFORM"] = "cpu"
        dev = torch.device("cpu")

    def dev_sync(all_streams=False):
        if on_gpu:
            (torch.cuda.synchronize() if all_streams else torch.cuda.current_stream().synchronize())

    from areal_vllm_amd.api.alloc_mode import ParallelStrategy
    from areal_vllm_amd.api.cli_args import (AdvNormConfig, GenerationHyperparameters, InferenceEngineConfig,
                                             InferenceServerConfig, MicroBatchSpec, OptimizerConfig, PPOActorConfig)
    from areal_vllm_amd.api.io_struct import FinetuneSpec, WeightUpdateMeta
    from areal_vllm_amd.engine.local_engine import LocalInferenceEngine
    from areal_vllm_amd.engine.ppo.actor import NativePPOActor
    from areal_vllm_amd.inference.engine import NativeInferenceEngine
    from areal_vllm_amd.utils import seeding, stats_tracker
    from areal_vllm_amd.workflow.rlvr import RLVRWorkflow

    seeding.set_random_seed(1, f"trainer{rank}")
    n_seqs_per_gpu = args.prompts_per_gpu * args.n_samples

    # ---- trainer (hyper-parameters of examples/math/gsm8k_grpo.yaml in the reference)
    actor_cfg = PPOActorConfig(
        experiment_name="bench", trial_name="bench", path=args.model, init_from_scratch=True,
        dtype="bfloat16" if on_gpu else "float32",
        gradient_checkpointing=False, mb_spec=MicroBatchSpec(max_tokens_per_mb=args.max_tokens_per_mb),
        optimizer=OptimizerConfig(type="adam", lr=1.7e-5, weight_decay=0.017, beta1=0.9, beta2=0.999, eps=1e-8,
                                  lr_scheduler_type="constant", gradient_clipping=1.0, warmup_steps_proportion=0.001),
        group_size=args.n_samples, eps_clip=0.4, temperature=1.0, reward_scaling=10.0, reward_bias=-0.5, kl_ctl=0.0,
        ppo_n_minibatches=1, recompute_logprob=True, use_decoupled_loss=True, behav_imp_weight_cap=5.0,
        fuse_proximal_logp=args.fuse_prox,
        adv_norm=AdvNormConfig(mean_level="group", std_level="group", group_size=args.n_samples),
        max_new_tokens=args.max_new_tokens,
    )
    actor = NativePPOActor(actor_cfg)
    actor.create_process_group(ParallelStrategy(data_parallel_size=world))
    actor.initialize(None, FinetuneSpec(total_train_epochs=1, dataset_size=10**6,
                                        train_batch_size=args.prompts_per_gpu * world))

    # ---- rollout engine on the same GPU
    srv = InferenceServerConfig(model_path=args.model, random_seed=1 + rank, dtype="bfloat16" if on_gpu else "float32",
                                context_length=args.prompt_len_max + args.max_new_tokens + 64,
                                max_running_requests=n_seqs_per_gpu * (args.eta + 1), max_prefill_tokens=32768,
                                num_continuous_decode_steps=args.decode_steps, disable_cuda_graph=args.no_graph,
                                schedule_policy="fcfs")
    engine = NativeInferenceEngine(srv, model_config=actor.model_config, device=dev,
                                   kv_cache_gb=args.kv_cache_gb if on_gpu else None,
                                   num_pages=None if on_gpu else 2048)
    engine.initialize(load_weights=False)
    actor.connect_local_engine(engine)
    rollout = LocalInferenceEngine(
        InferenceEngineConfig(experiment_name="bench", trial_name="bench",
                              max_concurrent_rollouts=args.prompts_per_gpu * world * (args.eta + 1),
                              consumer_batch_size=args.prompts_per_gpu * world, max_head_offpolicyness=args.eta,
                              queue_size=args.prompts_per_gpu * 64), engine)
    engine.abort_on_weight_update = not args.retain_kv
    rollout.initialize(train_data_parallel_size=world)
    meta = WeightUpdateMeta.from_inproc()
    actor.upload_weights(meta)  # trainer weights -> engine (both start from the same values)
    engine.set_version(0)

    mc = actor.model_config
    gconfig = GenerationHyperparameters(n_samples=args.n_samples, max_new_tokens=args.max_new_tokens, temperature=1.0,
                                        stop_token_ids=[t for t in (mc.eos_token_id, mc.pad_token_id) if t is not None])

    def reward_fn(prompt, completion, prompt_ids, completion_ids, **kw):
        # synthetic verifiable reward: deterministic in the completion, about half of the samples succeed
        return float((sum(completion_ids[:16]) + len(completion_ids)) % 2)

    workflow = RLVRWorkflow(reward_fn, gconfig, tokenizer=None, reward_in_process=True)
    g = torch.Generator().manual_seed(1234 + rank)

    def make_prompts():
        out = []
        for _ in range(args.prompts_per_gpu):
            n = int(torch.randint(args.prompt_len_min, args.prompt_len_max + 1, (1,), generator=g))
            out.append(dict(input_ids=torch.randint(2, min(100000, mc.vocab_size), (n,), generator=g).tolist()))
        return out

    class PromptLoader:
        """Endless synthetic prompt stream in batches of one consumer batch per rank."""
        batch_size = args.prompts_per_gpu

        def __iter__(self):
            while True:
                yield make_prompts()

    loader = PromptLoader()
    phase = {}

    trace = []  # (--profile-phases) per phase: name, seconds, tokens the rollout engine decoded / prefilled meanwhile

    def timed(name):
        class _T:
            def __enter__(s):
                dev_sync()
                s.t = time.perf_counter()
                s.d, s.p = engine.runner.n_decode_tokens, engine.runner.n_prefill_tokens

            def __exit__(s, *a):
                dev_sync()
                dt = time.perf_counter() - s.t
                phase[name] = phase.get(name, 0.0) + dt
                if args.profile_phases:
                    trace.append((name, round(dt, 3), engine.runner.n_decode_tokens - s.d,
                                  engine.runner.n_prefill_tokens - s.p))
        return _T()

    tot = dict(samples=0, gen_tokens=0, all_tokens=0)

    def one_step(step, count):
        with timed("rollout"):
            if args.eta > 0:
                batch = rollout.prepare_batch(loader, workflow=workflow)
            else:
                batch = rollout.rollout_batch(make_prompts(), workflow=workflow)
            batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
        if count:
            tot["samples"] += batch["input_ids"].shape[0]
            tot["gen_tokens"] += int(batch["loss_mask"].sum())
            tot["all_tokens"] += int(batch["attention_mask"].sum())
            lm = batch["loss_mask"].bool()
            tot["stale_sum"] = tot.get("stale_sum", 0.0) + float((step - batch["versions"][lm]).float().mean())
        if not args.fuse_prox:
            with timed("recompute_logp"):
                batch["prox_logp"] = actor.compute_logp(batch)
        with timed("advantages"):
            actor.compute_advantages(batch)
        with timed("train"):
            stats = actor.ppo_update(batch)
            actor.step_lr_scheduler()
        with timed("update_weights"):
            rollout.pause()
            actor.upload_weights(meta)
            actor.set_version(step + 1)
            rollout.set_version(step + 1)
            rollout.resume()
        stats_tracker.export_all(reduce_group=actor.data_parallel_group)
        return stats

    def barrier():
        if world > 1:
            dist.barrier()
        dev_sync(all_streams=True)

    prime = args.eta + 1 if args.eta > 0 else 0
    for s in range(prime + args.warmup):
        one_step(s, False)
    phase.clear()
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        stats = one_step(prime + args.warmup + s, True)
    barrier()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    cnt = torch.tensor([tot["samples"], tot["gen_tokens"], tot["all_tokens"]], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt)
    elapsed = float(el.item())
    samples, gen_tok, all_tok = (float(x) for x in cnt.tolist())
    if rank == 0:
        value = samples / elapsed
        mean_len = all_tok / max(1.0, samples)
        tok_s = all_tok / elapsed
        per_gpu = BASELINE_TOK_S_PER_GPU.get(args.model)
        baseline_tok_s = per_gpu * world if per_gpu else None
        label = MODEL_LABEL.get(args.model, args.model)
        out = {
            "metric": f"RL samples/sec (whole node), {label} GRPO end-to-end (rollout + train + weight update)",
            "value": round(value, 4), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1000.0 * elapsed / args.steps, 2),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": round(tok_s / baseline_tok_s, 4) if baseline_tok_s else None, "dtype": "bf16",
            "data": "synthetic prompts, random-init weights",
            "gen_tokens_per_s": round(gen_tok / elapsed, 1), "e2e_tokens_per_s": round(tok_s, 1),
            "mean_seq_len": round(mean_len, 1), "mean_gen_len": round(gen_tok / max(1.0, samples), 1),
            "phase_s_per_step_rank0": {k: round(v / args.steps, 4) for k, v in phase.items()},
            "grad_norm_last": stats[-1].get("grad_norm"),
            "mean_token_staleness_rank0": round(tot.get("stale_sum", 0.0) / args.steps, 3),
            "pipeline_prime_steps": prime,
            "baseline_note": "vs_baseline = end-to-end tokens/s over the reference's published per-GPU figure x n_gpus "
                             "(1.5B: 20k tok/s on one 8xH800 node, 7B: 26k tok/s on 32 H800; blog/AReaL_v0_2.md:54); "
                             "it publishes no samples/s number",
            "config": {"model": f"{label} (random init)", "global_batch": int(args.prompts_per_gpu * world),
                       "n_samples": args.n_samples, "seq_len": args.prompt_len_max + args.max_new_tokens,
                       "max_new_tokens": args.max_new_tokens, "parallelism": f"dp{world} colocated rollout+train",
                       "async_training": args.eta > 0, "max_head_offpolicyness": args.eta,
                       "interrupt_on_weight_update": "retain-kv" if args.retain_kv else "abort+recompute",
                       "algo": "GRPO, decoupled PPO loss, recompute_logprob, ppo_n_minibatches=1"
                               + (", proximal log-probs fused into the training forward" if args.fuse_prox else ""),
                       "max_tokens_per_mb": args.max_tokens_per_mb, "decode_steps_per_plan": args.decode_steps},
        }
        sys.stdout.flush()
        os.dup2(result_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
        if args.profile_phases:
            print(json.dumps(phase), file=sys.stderr)
            for row in trace:
                print("phase %-15s %7.3fs  decoded %7d  prefilled %7d" % row, file=sys.stderr)
    rollout.destroy()
    engine.destroy()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
