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

``--mode decoupled`` (needs an even N >= 2) is the reference's own allocation (``sglang.d4p1t1+d4p1t1`` at
N = 8, ``examples/math/gsm8k_grpo.yaml:3``): ranks 0 .. N/2-1 form the data-parallel trainer, ranks N/2 .. N-1
each run a rollout server process-in-rank (HTTP verbs of ``inference/server.py``) on their own GPU; the trainer
ranks reach them through the remote client, and after every step each server pulls the new weights out of a
trainer rank's exported flat buffer (peer memory, device-to-device over xGMI) while it keeps decoding.
``--seq-len S`` sets max_new_tokens = S - prompt_len_max (the 4k / 24k configurations).

Launch: ``python bench.py`` (1 GPU), ``python bench.py --gpus N --steps K --warmup W`` (the parent starts one
rank per GPU itself, picks a free rendezvous port on 127.0.0.1, relays rank 0's result line and fails if any rank
fails), or the same command line under ``python -m torch.distributed.run --nproc-per-node N ...`` (the ranks then read
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher).
Rank 0 prints ONE JSON line. `value` = RL samples (trajectories) per second over the whole job; `rccl_ranks_seen` /
`collective_backend` / `devices` say how many ranks took part in the trainer's collectives, over which backend and on
which PCI devices (asserted distinct), and with an even N >= 2 a short ``--mode decoupled`` pass in fresh child
processes is reported under `decoupled` (``--no-decoupled-pass`` skips it).
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
    p.add_argument("--kv-cache-dtype", default="auto", choices=["auto", "fp8_e4m3"],
                   help="element type of the rollout engine's KV cache. NOT the benchmark's recipe: fp8 is an opt-in "
                        "serving option measured separately; such a run is labelled in `dtype` and `config`")
    p.add_argument("--gradient-checkpointing", action="store_true",
                   help="recompute block activations in the backward (long-context configurations)")
    p.add_argument("--eta", type=int, default=2, help="max_head_offpolicyness; 0 = synchronous RL")
    p.add_argument("--retain-kv", action="store_true",
                   help="keep the KV cache of running generations across weight updates (no re-prefill)")
    p.add_argument("--device", default="cuda", choices=["cuda", "cpu"],
                   help="cpu = gloo rehearsal of the multi-rank control flow with a tiny model (not a benchmark)")
    p.add_argument("--fuse-prox", action="store_true",
                   help="take proximal log-probs from the training forward (valid for ppo_n_minibatches=1) instead of "
                        "a separate recompute pass; off by default to keep the reference's per-step work")
    p.add_argument("--profile-phases", action="store_true", help="print per-phase timings on rank 0 (stderr)")
    p.add_argument("--mode", default="colocated", choices=["colocated", "decoupled"],
                   help="colocated: every rank trains AND generates on its GPU; decoupled: first half of the ranks "
                        "train, second half serve rollouts (the reference's sglang.dXp1t1+dXp1t1 allocation)")
    p.add_argument("--seq-len", type=int, default=0,
                   help="total sequence budget: sets max_new_tokens = seq_len - prompt_len_max (0: keep --max-new-tokens)")
    p.add_argument("--weight-update", default="ipc", choices=["ipc", "disk"],
                   help="decoupled mode: how the rollout servers get new weights (ipc falls back to disk if the "
                        "peer mapping cannot be opened)")
    p.add_argument("--stall-timeout", type=float, default=1800.0,
                   help="seconds without passing a milestone (set-up stage, phase of a step) after which a rank dumps "
                        "its threads' stacks and exits with status 3 instead of hanging; 0 disables")
    p.add_argument("--share-gpu", action="store_true",
                   help="rehearsal on a single GPU: every rank uses device 0 (not a benchmark of N GPUs). With more than one "
                        "trainer rank the collectives are staged through host memory, and the processes time-slice "
                        "the card, so expect minutes per step at the headline config - pair it with --model small")
    p.add_argument("--no-decoupled-pass", action="store_true",
                   help="colocated runs on an even number of ranks >= 2 end with a short second pass in --mode decoupled "
                        "(fresh child processes, reported under 'decoupled'); this flag leaves it out")
    p.add_argument("--decoupled-steps", type=int, default=8, help="timed steps of that second pass (at most --steps)")
    p.add_argument("--decoupled-budget", type=float, default=300.0,
                   help="seconds the second pass may take before it is ended and reported as an error")
    a = p.parse_args()
    if a.seq_len:
        a.max_new_tokens = a.seq_len - a.prompt_len_max
        assert a.max_new_tokens > 0, "--seq-len must exceed --prompt-len-max"
    return a


def _store(args, rank):
    """Host-side rendezvous shared by trainer and rollout ranks (they are in no common process group)."""
    import datetime

    import torch.distributed as dist

    port = int(os.environ.get("MASTER_PORT", "29533")) + 2
    return dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), port, world_size=args.gpus, is_master=rank == 0,
                         timeout=datetime.timedelta(seconds=1800), wait_for_workers=False)


def serve_rollouts(args, rank, local_rank, store):
    """Body of a rollout rank in decoupled mode: one native engine behind the HTTP server, until the trainer
    side says the run is over."""
    import asyncio
    import socket
    import threading

    import torch
    from aiohttp import web

    from areal_vllm_amd.api.cli_args import InferenceServerConfig
    from areal_vllm_amd.inference.engine import NativeInferenceEngine
    from areal_vllm_amd.inference.server import InferenceServer
    from areal_vllm_amd.models.config import preset

    on_gpu = args.device == "cuda"
    dev = torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")
    n_train = args.gpus // 2
    # every trainer rank feeds prompts_per_gpu x n_samples sequences per step and may run eta+1 steps ahead
    n_seqs = args.prompts_per_gpu * args.n_samples
    srv = InferenceServerConfig(model_path=args.model, random_seed=1 + rank, dtype="bfloat16" if on_gpu else "float32",
                                context_length=args.prompt_len_max + args.max_new_tokens + 64,
                                max_running_requests=n_seqs * (args.eta + 1), max_prefill_tokens=32768,
                                num_continuous_decode_steps=args.decode_steps, disable_cuda_graph=args.no_graph,
                                schedule_policy="fcfs", kv_cache_dtype=args.kv_cache_dtype)
    mc = preset(args.model)
    engine = NativeInferenceEngine(srv, model_config=mc, device=dev, kv_cache_gb=args.kv_cache_gb if on_gpu else None,
                                   num_pages=None if on_gpu else 2048)
    engine.initialize(load_weights=False)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    server = InferenceServer(engine)
    loop = asyncio.new_event_loop()
    runner = web.AppRunner(server.app, access_log=None)

    def run():
        asyncio.set_event_loop(loop)
        loop.run_until_complete(runner.setup())
        loop.run_until_complete(web.TCPSite(runner, "127.0.0.1", port).start())
        loop.run_forever()

    threading.Thread(target=run, daemon=True, name="rollout-http").start()
    store.set(f"server_addr_{rank - n_train}", f"127.0.0.1:{port}")
    store.wait(["bench_done"])
    m = engine.metrics()
    store.set(f"server_stats_{rank - n_train}", json.dumps(dict(gen_tokens=m["gen_tokens_total"], aborted=m["aborted"],
                                                                 prefill_time=m["prefill_time"],
                                                                 decode_time=m["decode_time"])))
    loop.call_soon_threadsafe(loop.stop)
    engine.destroy()


def _port_block_free(base, offsets=(0, 1, 2, 16, 17, 18)):
    import socket

    for off in offsets:
        with socket.socket() as sk:
            try:
                sk.bind(("127.0.0.1", base + off))
            except OSError:
                return False
    return True


def _pick_master_port():
    """Base of a free block of rendezvous ports BELOW the range the kernel hands out for `bind(port 0)`: the job's own
    servers and gloo pairs take ports from that range while it runs, and a base taken from there was seen to lose
    its +1 neighbour (the decoupled mode's trainer group) to one of them."""
    import random

    lo = 32768
    try:
        lo = int(open("/proc/sys/net/ipv4/ip_local_port_range").read().split()[0])
    except (OSError, ValueError, IndexError):
        pass
    top = max(12000, min(lo, 65000) - 64)
    rnd = random.SystemRandom()
    for _ in range(256):
        base = rnd.randrange(10000, top)
        if _port_block_free(base):
            return base
    return rnd.randrange(10000, top)


def launch_ranks(args):
    """``python bench.py --gpus N`` with no torchrun environment: start N fresh rank processes (one per GPU,
    torchrun-style environment), relay rank 0's single result line, fail if any rank fails. This parent never
    imports torch and never touches a GPU (role of the reference's local launcher, areal/launcher/local.py:318-330)."""
    import signal
    import subprocess
    import threading

    port = _pick_master_port()
    procs = []
    lines = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AREAL_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "8" if args.device == "cuda" else "1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=r == 0 or None,
                                      start_new_session=True))

    def pump():
        for line in procs[0].stdout:
            if line.strip():
                lines.append(line.strip())

    reader = threading.Thread(target=pump, daemon=True)
    reader.start()

    def stop_all(sig=signal.SIGTERM):
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, sig)  # the rank and whatever it started (its own session: nothing else)
                except ProcessLookupError:
                    pass

    def on_signal(signum, frame):
        stop_all()
        sys.exit(128 + signum)

    signal.signal(signal.SIGTERM, on_signal)
    signal.signal(signal.SIGINT, on_signal)
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            r, rc = bad[0]
            print(f"[bench] rank {r} exited with status {rc}; stopping the other ranks", file=sys.stderr, flush=True)
            stop_all()
            time.sleep(5.0)
            stop_all(signal.SIGKILL)
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.2)
    reader.join(timeout=10.0)
    if rc != 0:
        return rc if 0 < rc < 256 else 1
    results = [l for l in lines if l.startswith("{")]
    if len(results) != 1:
        print(f"[bench] expected one result line from rank 0, got {len(results)}", file=sys.stderr)
        return 1
    print(results[0], flush=True)
    return 0


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))
    run_rank(args)


def run_rank(args):
    if os.environ.get("AREAL_BENCH_HANG_DUMP"):  # debugging aid: stack of every thread after that many seconds
        import faulthandler

        os.makedirs("gpurun_out", exist_ok=True)
        _dump = open(f"gpurun_out/hang_dump_rank{os.environ.get('RANK', '0')}.txt", "w")
        faulthandler.dump_traceback_later(float(os.environ["AREAL_BENCH_HANG_DUMP"]), repeat=False, file=_dump)
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
    import threading

    t_start = time.perf_counter()
    progress = [t_start, "start"]  # (time, label) of the last milestone this rank passed

    def mark(label):
        progress[0], progress[1] = time.perf_counter(), label

    def heartbeat():
        # rank 0: a line a minute shows that a long configuration is alive. Every rank: a run that passes no
        # milestone (set-up stage, phase of a step) for --stall-timeout seconds - a collective waiting for a rank
        # that died, a wedged device - prints every thread's stack and exits non-zero instead of hanging
        tick = min(60.0, max(1.0, args.stall_timeout / 4)) if args.stall_timeout > 0 else 60.0
        said = t_start
        while True:
            time.sleep(tick)
            now = time.perf_counter()
            if rank == 0 and now - said >= 60:
                said = now
                print(f"[bench] running, {now - t_start:.0f} s since start (last: {progress[1]})", file=sys.stderr,
                      flush=True)
            if args.stall_timeout > 0 and now - progress[0] > args.stall_timeout:
                import faulthandler

                print(f"[bench] rank {rank}: no progress for {now - progress[0]:.0f} s after '{progress[1]}', giving up",
                      file=sys.stderr, flush=True)
                faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
                os._exit(3)

    threading.Thread(target=heartbeat, daemon=True, name="bench-heartbeat").start()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("AREAL_LOG_LEVEL", "WARNING")
    on_gpu = args.device == "cuda"
    decoupled = args.mode == "decoupled"
    if args.share_gpu:
        local_rank = 0
        os.environ["LOCAL_RANK"] = "0"
        n_train = world // 2 if decoupled else world
        if n_train > 1:
            # several trainer ranks on ONE device: RCCL cannot host them, the collectives go through the host
            # backend with staged copies (a rehearsal of the control flow and the kernels, not of the transport)
            os.environ["AREAL_HOST_STAGED_COLLECTIVES"] = "1"
    if on_gpu:
        assert torch.cuda.is_available(), "bench.py needs an MI355X"
        if not args.share_gpu:
            assert torch.cuda.device_count() > local_rank, (
                f"rank {rank} wants GPU {local_rank} but only {torch.cuda.device_count()} are visible")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        os.environ["HIP_VISIBLE_DEVICES"] = os.environ["CUDA_VISIBLE_DEVICES"] = ""
        os.environ["AREAL_PLATFORM"] = "cpu"
        dev = torch.device("cpu")
    store = None
    n_gen = 0
    if decoupled:
        assert world >= 2 and world % 2 == 0, "--mode decoupled splits the ranks in two halves: needs an even --gpus >= 2"
        n_gen = world // 2
        store = _store(args, rank)
        if rank >= world - n_gen:
            progress[0] = float("inf")  # a server rank passes no milestones: it lives as long as the trainer side
            serve_rollouts(args, rank, local_rank, store)
            return
        # the trainer half forms its own process group: ranks 0 .. n_train-1 on the next port
        world = world - n_gen
        os.environ["WORLD_SIZE"] = str(world)
        os.environ["MASTER_PORT"] = str(int(os.environ["MASTER_PORT"]) + 1)
        # torchrun's agent store lives on the original port and counts N ranks: the trainer group rendezvouses
        # on a store of its own (rank 0 hosts it)
        os.environ["TORCHELASTIC_USE_AGENT_STORE"] = "False"
        addrs = [store.get(f"server_addr_{i}").decode() for i in range(n_gen)]
        os.environ["AREAL_LLM_SERVER_ADDRS"] = ",".join(addrs)

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
        gradient_checkpointing=args.gradient_checkpointing, mb_spec=MicroBatchSpec(max_tokens_per_mb=args.max_tokens_per_mb),
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
    mark("process group")
    actor.initialize(None, FinetuneSpec(total_train_epochs=1, dataset_size=10**6,
                                        train_batch_size=args.prompts_per_gpu * world))

    mark("trainer initialised")
    engine = None
    if decoupled:
        from areal_vllm_amd.engine.remote_engine import RemoteInferenceEngine

        rollout = RemoteInferenceEngine(
            InferenceEngineConfig(experiment_name="bench", trial_name="bench",
                                  max_concurrent_rollouts=args.prompts_per_gpu * world * (args.eta + 1),
                                  consumer_batch_size=args.prompts_per_gpu * world, max_head_offpolicyness=args.eta,
                                  queue_size=args.prompts_per_gpu * 64, setup_timeout=600, request_timeout=3600))
        rollout.initialize(train_data_parallel_size=world)
        transport = args.weight_update
        meta = None
        if transport == "ipc":
            try:
                meta = WeightUpdateMeta.from_ipc(actor)
            except Exception as e:  # e.g. a platform without device IPC
                print(f"[bench] peer-memory export failed ({e!r}); weights go through checkpoint files", file=sys.stderr)
                transport = "disk"
        if meta is None:
            import tempfile

            box = [tempfile.mkdtemp(prefix="areal_bench_wu_") if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(box, src=0)
            meta = WeightUpdateMeta.from_disk("bench", "bench", box[0])
    # ---- rollout engine on the same GPU
    srv = None if decoupled else InferenceServerConfig(model_path=args.model, random_seed=1 + rank, dtype="bfloat16" if on_gpu else "float32",
                                context_length=args.prompt_len_max + args.max_new_tokens + 64,
                                max_running_requests=n_seqs_per_gpu * (args.eta + 1), max_prefill_tokens=32768,
                                num_continuous_decode_steps=args.decode_steps, disable_cuda_graph=args.no_graph,
                                schedule_policy="fcfs", kv_cache_dtype=args.kv_cache_dtype)
    if not decoupled:
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

    wu = dict(transport="inproc" if not decoupled else meta.type, fallback=None)

    def push_weights(version):
        """Trainer weights -> rollout side, then the version bump both sides agree on."""
        nonlocal meta
        if not decoupled:
            rollout.pause()
            actor.upload_weights(meta)
            actor.set_version(version)
            rollout.set_version(version)
            rollout.resume()
            return
        try:
            fut = rollout.update_weights(meta) if rank == 0 else None
            actor.upload_weights(meta)
            ok = True
            if fut is not None:
                try:
                    fut.result(timeout=1800)
                except Exception as e:
                    ok = False
                    print(f"[bench] weight update over {meta.type} failed: {e!r}", file=sys.stderr)
            flag = torch.tensor([1 if ok else 0], device=dev)
            if world > 1:
                dist.broadcast(flag, src=0)
            if not int(flag.item()):
                raise RuntimeError("rollout servers could not take the weights")
        except RuntimeError:
            if meta.type != "ipc":
                raise
            import tempfile

            wu["fallback"] = "ipc -> disk"
            box = [tempfile.mkdtemp(prefix="areal_bench_wu_") if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(box, src=0)
            meta = WeightUpdateMeta.from_disk("bench", "bench", box[0])
            wu["transport"] = "disk"
            rollout.pause()
            fut = rollout.update_weights(meta) if rank == 0 else None
            actor.upload_weights(meta)
            if fut is not None:
                fut.result(timeout=1800)
            rollout.resume()
        if world > 1:
            dist.barrier()
        actor.set_version(version)
        rollout.set_version(version)

    if decoupled:
        push_weights(0)  # the servers start from their own random init: give them the trainer's weights

    mark("rollout side ready")
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
                mark(name)
                dev_sync()
                s.t = time.perf_counter()
                if engine is not None:
                    s.d, s.p = engine.runner.n_decode_tokens, engine.runner.n_prefill_tokens

            def __exit__(s, *a):
                dev_sync()
                dt = time.perf_counter() - s.t
                phase[name] = phase.get(name, 0.0) + dt
                if args.profile_phases and engine is not None:
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
            push_weights(step + 1)
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
    # proof of what the run saw: every rank adds a one over the trainer's collective backend (RCCL on GPUs) and
    # names the device it computed on
    ones = torch.ones(1, dtype=torch.float32, device=dev)
    if world > 1:
        dist.all_reduce(ones)
    ranks_seen = int(ones.item())
    if on_gpu:
        pr = torch.cuda.get_device_properties(local_rank)
        if all(hasattr(pr, k) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
            my_dev = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
        else:
            my_dev = str(getattr(pr, "uuid", f"cuda:{local_rank}"))
    else:
        my_dev = "cpu"
    devices = [my_dev]
    if world > 1:
        devices = [None] * world
        dist.all_gather_object(devices, my_dev)
    if on_gpu and not args.share_gpu:
        assert len(set(devices)) == world, f"ranks share a GPU: {devices}"
    backend = "none (single rank)"
    if world > 1:
        backend = "host-staged gloo" if os.environ.get("AREAL_HOST_STAGED_COLLECTIVES") == "1" else (
            "rccl" if on_gpu else "gloo")
    out = None
    if rank == 0:
        value = samples / elapsed
        mean_len = all_tok / max(1.0, samples)
        tok_s = all_tok / elapsed
        per_gpu = BASELINE_TOK_S_PER_GPU.get(args.model)
        baseline_tok_s = per_gpu * args.gpus if per_gpu else None
        label = MODEL_LABEL.get(args.model, args.model)
        server_stats = None
        if decoupled:
            store.set("bench_done", "1")
            server_stats = [json.loads(store.get(f"server_stats_{i}").decode()) for i in range(n_gen)]
        layout = (f"decoupled: {n_gen} rollout server rank(s) + dp{world} trainer" + (" sharing ONE GPU" if args.share_gpu else "")
                  if decoupled else f"dp{world} colocated rollout+train")
        out = {
            "metric": f"RL samples/sec (whole node), {label} GRPO end-to-end (rollout + train + weight update)",
            "value": round(value, 4), "unit": "samples/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1000.0 * elapsed / args.steps, 2),
            "higher_is_better": True, "scaling": "weak",
            # the reference publishes no samples/s figure for this configuration (nor any MI355X number)
            "vs_baseline": None,
            "dtype": "bf16" if args.kv_cache_dtype == "auto" else f"bf16 compute, {args.kv_cache_dtype} KV cache",
            "data": "synthetic prompts, random-init weights",
            "gen_tokens_per_s": round(gen_tok / elapsed, 1), "e2e_tokens_per_s": round(tok_s, 1),
            "mean_seq_len": round(mean_len, 1), "mean_gen_len": round(gen_tok / max(1.0, samples), 1),
            "phase_s_per_step_rank0": {k: round(v / args.steps, 4) for k, v in phase.items()},
            "grad_norm_last": stats[-1].get("grad_norm"),
            "mean_token_staleness_rank0": round(tot.get("stale_sum", 0.0) / args.steps, 3),
            "pipeline_prime_steps": prime,
            "mode": args.mode, "weight_update": wu,
            "rccl_ranks_seen": ranks_seen + n_gen, "trainer_ranks_in_collective": ranks_seen,
            "collective_backend": backend, "devices": devices,
            "rollout_servers": server_stats,
            "other_hardware_context": {
                "note": "NOT the same configuration: the reference's closest published figure is end-to-end tokens/s "
                        "of synchronous long-generation RL (1.5B: ~20k tok/s on one 8xH800 node, 7B: ~26k tok/s on "
                        "32 H800; blog/AReaL_v0_2.md:54); scaled per GPU for orientation only",
                "e2e_tokens_per_s_over_that_figure_per_gpu": round(tok_s / baseline_tok_s, 3) if baseline_tok_s else None},
            "config": {"model": f"{label} (random init)", "global_batch": int(args.prompts_per_gpu * world),
                       "n_samples": args.n_samples, "seq_len": args.prompt_len_max + args.max_new_tokens,
                       "max_new_tokens": args.max_new_tokens, "parallelism": layout,
                       "async_training": args.eta > 0, "max_head_offpolicyness": args.eta,
                       "interrupt_on_weight_update": "retain-kv" if args.retain_kv else "abort+recompute",
                       "algo": "GRPO, decoupled PPO loss, recompute_logprob, ppo_n_minibatches=1"
                               + (", proximal log-probs fused into the training forward" if args.fuse_prox else ""),
                       "max_tokens_per_mb": args.max_tokens_per_mb, "decode_steps_per_plan": args.decode_steps,
                       "gradient_checkpointing": bool(args.gradient_checkpointing),
                       "kv_cache_dtype": "bf16" if args.kv_cache_dtype == "auto" else args.kv_cache_dtype},
        }
        if args.profile_phases:
            print(json.dumps(phase), file=sys.stderr)
            for row in trace:
                print("phase %-15s %7.3fs  decoded %7d  prefilled %7d" % row, file=sys.stderr)
    mark("teardown")
    rollout.destroy()
    if engine is not None:
        engine.destroy()
    if getattr(actor, "_ipc_export", None) is not None:
        actor._ipc_export.close()
    second_port = None
    if world > 1:
        if not decoupled and not args.no_decoupled_pass:
            # the second pass rendezvouses on a port block of its own, picked by rank 0 below the ephemeral range
            box = [_pick_master_port() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            second_port = box[0]
        dist.barrier()
        dist.destroy_process_group()
    if (not decoupled and args.gpus >= 2 and args.gpus % 2 == 0 and not args.no_decoupled_pass
            and not (args.share_gpu and on_gpu)):
        # the reference's own allocation (half the ranks serve, half train) as a short second pass in fresh child
        # processes, one per rank: the headline value above stays the colocated one
        del actor, rollout, engine, workflow
        import gc

        gc.collect()
        if on_gpu:
            torch.cuda.empty_cache()
        mark("decoupled pass")
        second = decoupled_second_pass(args, rank, second_port)
        if out is not None:
            out["decoupled"] = second
    if out is not None:
        sys.stdout.flush()
        os.dup2(result_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)


def decoupled_second_pass(args, rank, port=None):
    """Run ``--mode decoupled`` on the same ranks in one fresh child process per rank and return rank 0's
    summary of its result line ({"error": ...} if the pass failed or overran its budget: the headline result
    must not depend on it)."""
    import subprocess

    steps = max(1, min(args.steps, args.decoupled_steps))
    env = dict(os.environ, MASTER_PORT=str(port or int(os.environ.get("MASTER_PORT", "29533")) + 16),
               TORCHELASTIC_USE_AGENT_STORE="False")
    cmd = [sys.executable, os.path.abspath(__file__), *sys.argv[1:], "--mode", "decoupled", "--steps", str(steps),
           "--warmup", "1", "--no-decoupled-pass", "--stall-timeout", str(args.decoupled_budget)]
    t0 = time.perf_counter()
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, text=True)
    try:
        stdout, _ = p.communicate(timeout=args.decoupled_budget)
    except subprocess.TimeoutExpired:
        p.kill()
        p.communicate()
        return {"error": f"no result within {args.decoupled_budget:.0f} s"}
    if rank != 0:
        return None
    if p.returncode != 0:
        return {"error": f"exit status {p.returncode}"}
    try:
        res = json.loads([l for l in stdout.splitlines() if l.startswith("{")][-1])
    except (IndexError, ValueError):
        return {"error": "no result line"}
    upd = res.get("phase_s_per_step_rank0", {}).get("update_weights")
    return {"value": res["value"], "unit": res["unit"], "gen_tokens_per_s": res["gen_tokens_per_s"],
            "weight_update_ms": None if upd is None else round(1000.0 * upd, 1),
            "transport": res["weight_update"]["transport"], "fallback": res["weight_update"]["fallback"],
            "steps": res["steps"], "warmup": res["warmup"], "ms_per_step": res["ms_per_step"],
            "layout": res["config"]["parallelism"], "rccl_ranks_seen": res.get("rccl_ranks_seen"),
            "wall_s": round(time.perf_counter() - t0, 1)}


if __name__ == "__main__":
    main()
