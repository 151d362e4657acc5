"""HTTP rollout server + remote client on the CPU platform: generation parity with the trainer's
log-probs, interruptible generation across a weight update, disk weight update protocol and metrics.
Mirrors ``areal/tests/test_sglang_engine.py`` (remote generate / rollout / disk update) without SGLang."""
import asyncio
import os
import socket
import sys
import tempfile
import threading
import time

import pytest
import requests
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def server():
    from aiohttp import web

    from areal_vllm_amd.api.cli_args import InferenceServerConfig
    from areal_vllm_amd.inference.engine import NativeInferenceEngine
    from areal_vllm_amd.inference.server import InferenceServer

    cfg = InferenceServerConfig(model_path="tiny", dtype="float32", context_length=256, max_running_requests=8,
                                num_continuous_decode_steps=2)
    eng = NativeInferenceEngine(cfg, device=torch.device("cpu"), num_pages=64).initialize(load_weights=False)
    srv = InferenceServer(eng)
    port = _free_port()
    loop = asyncio.new_event_loop()
    started = threading.Event()

    def run():
        asyncio.set_event_loop(loop)
        runner = web.AppRunner(srv.app)
        loop.run_until_complete(runner.setup())
        loop.run_until_complete(web.TCPSite(runner, "127.0.0.1", port).start())
        started.set()
        loop.run_forever()

    t = threading.Thread(target=run, daemon=True)
    t.start()
    assert started.wait(30)
    yield f"127.0.0.1:{port}", eng
    loop.call_soon_threadsafe(loop.stop)
    eng.destroy()


def test_health_metrics_and_generate(server):
    addr, eng = server
    assert requests.get(f"http://{addr}/health", timeout=10).status_code == 200
    r = requests.post(f"http://{addr}/generate", json={
        "input_ids": [5, 6, 7, 8], "sampling_params": {"max_new_tokens": 6, "temperature": 1.0},
        "return_logprob": True}, timeout=60).json()
    meta = r["meta_info"]
    assert meta["finish_reason"]["type"] in ("length", "stop")
    assert len(meta["output_token_logprobs"]) == meta["completion_tokens"] == len(r["output_ids"])
    assert all(lp <= 0 for lp, _, _ in meta["output_token_logprobs"])
    assert meta["output_versions"] == [0] * meta["completion_tokens"]
    text = requests.get(f"http://{addr}/metrics", timeout=10).text
    assert "sglang:num_running_reqs" in text and "areal:gen_tokens_total" in text


def test_remote_engine_rollout_and_disk_update(server, tmp_path):
    addr, eng = server
    from areal_vllm_amd.api.cli_args import GenerationHyperparameters, InferenceEngineConfig
    from areal_vllm_amd.api.io_struct import WeightUpdateMeta
    from areal_vllm_amd.engine.remote_engine import RemoteInferenceEngine
    from areal_vllm_amd.models import hf_io
    from areal_vllm_amd.utils import name_resolve, names
    from areal_vllm_amd.workflow.rlvr import RLVRWorkflow

    name_resolve.reconfigure("memory")
    cfg = InferenceEngineConfig(experiment_name="srv", trial_name="t", max_concurrent_rollouts=4,
                                consumer_batch_size=2, setup_timeout=30)
    client = RemoteInferenceEngine(cfg).initialize(addr=addr, train_data_parallel_size=1)
    wf = RLVRWorkflow(lambda p, c, pi, ci, **kw: float(len(ci) % 2), GenerationHyperparameters(
        n_samples=2, max_new_tokens=8, stop_token_ids=[1]), reward_in_process=True)
    batch = client.rollout_batch([dict(input_ids=[3, 4, 5]), dict(input_ids=[9, 8, 7, 6, 5])], workflow=wf)
    assert batch["input_ids"].shape[0] == 4
    assert (batch["versions"][batch["loss_mask"].bool()] == 0).all()

    # disk weight update: write the engine's own (perturbed) weights as a HF checkpoint
    before = eng.model.flat.flat.clone()
    eng.model.flat.flat.mul_(1.5)
    path = str(tmp_path / "weight_update")
    wm = hf_io.save_hf_checkpoint(eng.model, path)
    hf_io.finalize_hf_checkpoint(path, eng.model_config, [wm])
    eng.model.flat.flat.copy_(before)
    name_resolve.add(names.update_weights_from_disk("srv", "t", 0), str(time.time_ns()), replace=True)
    client.pause()
    fut = client.update_weights(WeightUpdateMeta(type="disk", path=path))
    assert fut.result(timeout=120)
    client.set_version(1)
    client.resume()
    torch.testing.assert_close(eng.model.flat.flat, before * 1.5)
    assert eng.get_version() == 1
    batch = client.rollout_batch([dict(input_ids=[3, 4, 5])], workflow=wf)
    assert (batch["versions"][batch["loss_mask"].bool()] == 1).all()
    client.destroy()


def test_interrupted_generation_spans_versions(server):
    """A long generation that is aborted by a weight update is re-issued and finishes with tokens from
    both versions (partial-rollout semantics)."""
    addr, eng = server
    from areal_vllm_amd.api.cli_args import GenerationHyperparameters, InferenceEngineConfig
    from areal_vllm_amd.api.io_struct import ModelRequest
    from areal_vllm_amd.engine.remote_engine import RemoteInferenceEngine

    base = eng.get_version()
    client = RemoteInferenceEngine(InferenceEngineConfig(setup_timeout=30)).initialize(addr=addr)
    client.set_version(base)

    async def go():
        req = ModelRequest(input_ids=[4, 5, 6], gconfig=GenerationHyperparameters(max_new_tokens=240, min_new_tokens=240))
        done_before = eng.stats["decode_steps"]  # counts decode plans as they complete
        task = asyncio.create_task(client.agenerate(req))
        # interrupt once the request is under way - by progress, not by wall time (a fast host finishes 100 tokens of
        # this model in a fraction of a second)
        for _ in range(4000):
            if eng.stats["decode_steps"] - done_before >= 4:
                break
            await asyncio.sleep(0.002)
        loop = asyncio.get_running_loop()
        await loop.run_in_executor(None, lambda: requests.post(f"http://{addr}/pause_generation", timeout=30))
        await loop.run_in_executor(None, lambda: requests.post(
            f"http://{addr}/abort_request", json={}, timeout=30))
        await loop.run_in_executor(None, lambda: requests.post(
            f"http://{addr}/set_version", json={"version": base + 1}, timeout=30))
        await loop.run_in_executor(None, lambda: requests.post(f"http://{addr}/continue_generation", timeout=30))
        return await task

    resp = asyncio.run(go())
    assert resp.output_len == 240
    vs = set(resp.output_versions)
    assert vs <= {base, base + 1} and (base + 1) in vs
    assert resp.output_versions == sorted(resp.output_versions)
    client.destroy()


@pytest.mark.parametrize("transport", ["collective", "ipc"])
def test_collective_weight_update_into_server_process(tmp_path, transport):
    """`ipc`: the server maps the trainer's exported flat buffer (POSIX shared memory on the CPU platform,
    hipIpc device memory on GPUs) and pulls it while it keeps serving. `collective`: default weight path of the decoupled allocation: trainer rank 0 and the rollout server join a second
    process group and the trainer broadcasts every HF-named parameter chunk by chunk (gloo on CPU, RCCL on
    GPUs). Afterwards the server scores its own samples like the trainer does."""
    import subprocess

    from areal_vllm_amd.api.alloc_mode import AllocationMode
    from areal_vllm_amd.api.cli_args import (AdvNormConfig, GenerationHyperparameters, InferenceEngineConfig,
                                             MicroBatchSpec, OptimizerConfig, PPOActorConfig)
    from areal_vllm_amd.api.io_struct import FinetuneSpec, WeightUpdateMeta
    from areal_vllm_amd.engine.ppo.actor import NativePPOActor
    from areal_vllm_amd.engine.remote_engine import RemoteInferenceEngine
    from areal_vllm_amd.workflow.rlvr import RLVRWorkflow

    os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
                      AREAL_PLATFORM="cpu", GLOO_SOCKET_IFNAME="lo")
    port = _free_port()
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    srv = subprocess.Popen([sys.executable, "-m", "areal_vllm_amd.inference.server", "--model-path", "tiny", "--dtype",
                            "float32", "--device", "cpu", "--num-pages", "64", "--context-length", "128",
                            "--max-running-requests", "8", "--port", str(port)], env=env, cwd=ROOT)
    actor = client = None
    try:
        deadline = time.time() + 120
        while True:
            try:
                if requests.get(f"http://127.0.0.1:{port}/health", timeout=2).status_code == 200:
                    break
            except requests.RequestException:
                pass
            assert srv.poll() is None and time.time() < deadline
            time.sleep(0.5)
        cfg = PPOActorConfig(experiment_name="col", trial_name="t", path="tiny", init_from_scratch=True, dtype="float32",
                             optimizer=OptimizerConfig(lr=1e-3), group_size=2, mb_spec=MicroBatchSpec(max_tokens_per_mb=2048),
                             ppo_n_minibatches=1, use_decoupled_loss=True, recompute_logprob=True, kl_ctl=0.0,
                             adv_norm=AdvNormConfig("group", "group", 2), gradient_checkpointing=False)
        actor = NativePPOActor(cfg)
        actor.create_process_group()
        actor.initialize(None, FinetuneSpec(1, 16, 4))
        actor.model.flat.flat.mul_(1.3)
        client = RemoteInferenceEngine(InferenceEngineConfig(experiment_name="col", trial_name="t", setup_timeout=60,
                                                             max_concurrent_rollouts=8, consumer_batch_size=4))
        client.initialize(addr=f"127.0.0.1:{port}", train_data_parallel_size=1)
        if transport == "ipc":
            meta = WeightUpdateMeta.from_ipc(actor)
            assert meta.ipc_buffers and meta.ipc_used_bytes > 0
        else:
            meta = WeightUpdateMeta.from_fsdp_nccl(AllocationMode.from_str("native:d1+native:d1"), actor,
                                                   weight_chunked_mem_mb=1)
            meta.nccl_master_address = "127.0.0.1"
            assert len(meta.nccl_param_specs) > 1  # several chunks
        wf = RLVRWorkflow(lambda p, c, pi, ci, **kw: 1.0, GenerationHyperparameters(n_samples=2, max_new_tokens=10),
                          reward_in_process=True)
        prompts = [dict(input_ids=[5, 6, 7, 8]), dict(input_ids=[9, 10, 11])]

        def gap():
            b = client.rollout_batch(prompts, workflow=wf)
            prox = actor.compute_logp(b)
            lm = torch.roll(b["loss_mask"].bool(), -1, 1)
            lm[:, -1] = False
            return float((prox - torch.roll(b["logprobs"], -1, 1)).abs()[lm].max()), b

        g0, _ = gap()
        assert g0 > 1e-2
        for version in (1, 2):  # the second round reuses the group
            client.pause()
            fut = client.update_weights(meta)
            actor.upload_weights(meta)
            assert fut.result(timeout=120)
            actor.set_version(version)
            client.set_version(version)
            client.resume()
            g, b = gap()
            assert g < 1e-3, g
            assert (b["versions"][b["loss_mask"].bool()] == version).all()
            actor.model.flat.flat.mul_(0.9)
    finally:
        if client is not None:
            client.destroy()
        if actor is not None:
            actor.destroy()
        srv.terminate()
        try:
            srv.wait(timeout=20)
        except subprocess.TimeoutExpired:
            srv.kill()


def test_colocated_weight_copy_with_trainer_side_padding():
    """The trainer pads its flat parameter buffer to a multiple of the data-parallel size (ZeRO shards); the
    colocated copy into the rollout engine must move the parameters, not compare buffer lengths."""
    from areal_vllm_amd.api.cli_args import InferenceServerConfig
    from areal_vllm_amd.inference.engine import NativeInferenceEngine
    from areal_vllm_amd.models.config import preset
    from areal_vllm_amd.models.transformer import Transformer

    cfg = InferenceServerConfig(model_path="tiny", dtype="float32", context_length=128, max_running_requests=4)
    eng = NativeInferenceEngine(cfg, device=torch.device("cpu"), num_pages=32).initialize(load_weights=False)
    try:
        for world in (1, 3, 8):
            trainer = Transformer(preset("tiny"), dtype=torch.float32, device="cpu")
            trainer.flatten(pad_to=world)
            trainer.init_weights(seed=10 + world)
            eng.load_weights_from_trainer(trainer, version=world)
            for (n, a), (_, b) in zip(eng.model.named_parameters(), trainer.named_parameters()):
                assert torch.equal(a, b), n
            assert eng.get_version() == world
        if trainer.flat.numel == eng.model.flat.numel:  # make sure the case that used to fail is covered
            pytest.skip("padding did not change the buffer length for this model")
    finally:
        eng.destroy()


def _sharded_push_worker(rank, world, port, srv_port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), AREAL_PLATFORM="cpu", GLOO_SOCKET_IFNAME="lo")
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    import torch.distributed as dist

    from areal_vllm_amd.api.alloc_mode import AllocationMode, ParallelStrategy
    from areal_vllm_amd.api.cli_args import InferenceEngineConfig, MicroBatchSpec, OptimizerConfig, TrainEngineConfig
    from areal_vllm_amd.api.io_struct import FinetuneSpec, WeightUpdateMeta
    from areal_vllm_amd.engine.remote_engine import RemoteInferenceEngine
    from areal_vllm_amd.engine.train_engine import NativeTrainEngine

    cfg = TrainEngineConfig(experiment_name="push", trial_name="t", path="tiny", init_from_scratch=True, dtype="float32",
                            gradient_checkpointing=False, optimizer=OptimizerConfig(lr=1e-3),
                            mb_spec=MicroBatchSpec(max_tokens_per_mb=2048))
    eng = NativeTrainEngine(cfg)
    eng.create_process_group(ParallelStrategy(data_parallel_size=world))
    eng.initialize(None, FinetuneSpec(1, 16, 4))
    client = None
    try:
        box = [None]
        if rank == 0:
            client = RemoteInferenceEngine(InferenceEngineConfig(experiment_name="push", trial_name="t", setup_timeout=60,
                                                                 max_concurrent_rollouts=8, consumer_batch_size=4))
            client.initialize(addr=f"127.0.0.1:{srv_port}", train_data_parallel_size=world)
            meta = WeightUpdateMeta.from_fsdp_nccl(AllocationMode.from_str("native:d1+native:d2"), eng,
                                                   weight_chunked_mem_mb=1, n_senders=2)
            meta.nccl_master_address = "127.0.0.1"
            assert len(meta.nccl_param_specs) >= 3 and len(set(meta.nccl_master_ports)) == 2
            box = [meta]
        dist.broadcast_object_list(box, src=0)
        meta = box[0]
        for version, scale in ((1, 1.3), (2, 0.7)):  # the second round reuses both groups
            eng.model.flat.flat.mul_(scale)  # identical replicas on both ranks
            fut = None
            if rank == 0:
                client.pause()
                fut = client.update_weights(meta)
            eng.upload_weights(meta)
            if rank == 0:
                assert fut.result(timeout=120)
                client.set_version(version)
                client.resume()
                r = requests.post(f"http://127.0.0.1:{srv_port}/generate", json=dict(
                    input_ids=[5, 6, 7, 8, 9], sampling_params=dict(max_new_tokens=6, temperature=0.0)), timeout=60).json()
                toks = [5, 6, 7, 8, 9] + r["output_ids"]
                ids = torch.tensor(toks)
                m = eng.model
                with torch.no_grad():
                    h = m(ids, torch.arange(len(toks), dtype=torch.int32), torch.tensor([0, len(toks)], dtype=torch.int32),
                          len(toks))
                    lp = torch.log_softmax(m.logits(h).float(), -1)
                want = lp[torch.arange(4, len(toks) - 1), ids[5:]]
                got = torch.tensor([x[0] for x in r["meta_info"]["output_token_logprobs"]])
                assert float((got - want).abs().max()) < 1e-3, (got, want)
                assert (lp[torch.arange(4, len(toks) - 1)].argmax(-1) == ids[5:]).all()  # greedy under the NEW weights
        if rank == 0:
            open(os.path.join(out_dir, "ok"), "w").write("ok")
    finally:
        if client is not None:
            client.destroy()
        eng.destroy()


def test_sharded_weight_push_from_two_trainer_ranks(tmp_path):
    """Two data-parallel trainer ranks each broadcast every other parameter chunk into their own group with
    the rollout server; the server starts the receives of a round together. Afterwards it decodes greedily
    with exactly the trainer's new weights (two rounds: the groups are reused)."""
    import subprocess

    import torch.multiprocessing as mp

    srv_port = _free_port()
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1", AREAL_PLATFORM="cpu", GLOO_SOCKET_IFNAME="lo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    srv = subprocess.Popen([sys.executable, "-m", "areal_vllm_amd.inference.server", "--model-path", "tiny", "--dtype",
                            "float32", "--device", "cpu", "--num-pages", "64", "--context-length", "128",
                            "--max-running-requests", "8", "--port", str(srv_port)], env=env, cwd=ROOT)
    try:
        deadline = time.time() + 120
        while True:
            try:
                if requests.get(f"http://127.0.0.1:{srv_port}/health", timeout=2).status_code == 200:
                    break
            except requests.RequestException:
                pass
            assert srv.poll() is None and time.time() < deadline
            time.sleep(0.5)
        mp.spawn(_sharded_push_worker, args=(2, _free_port(), srv_port, str(tmp_path)), nprocs=2, join=True)
        assert (tmp_path / "ok").exists()
    finally:
        srv.terminate()
        try:
            srv.wait(timeout=20)
        except subprocess.TimeoutExpired:
            srv.kill()
