"""Unit tests of the worker runtime's building blocks: samples, data-flow graph, buffer, streams, worker
control, the generation-server manager against mock servers (staleness, routing, weight roll-out), chunked
generation and the agents. Patterns of the reference's ``realhf/tests/{data,system,agent}``."""
import asyncio
import dataclasses
import json
import os
import threading
import time

import pytest
import torch
from aiohttp import web

from areal_vllm_amd.api.cli_args import GenerationHyperparameters, MicroBatchSpec
from areal_vllm_amd.utils import name_resolve, names


@pytest.fixture(autouse=True)
def _memory_name_resolve():
    cfg = type("C", (), dict(type="memory"))()
    name_resolve.reconfigure(cfg)
    yield
    name_resolve.reconfigure(type("C", (), dict(type="nfs", nfs_record_root="/tmp/areal/name_resolve"))())


# ------------------------------------------------------------------------------- samples
def _sample(ids, lens, group=1):
    from areal_vllm_amd.system.data_api import SequenceSample

    tot = sum(lens)
    return SequenceSample.from_default(ids=ids, seqlens=lens, data=dict(
        packed_input_ids=torch.arange(tot), packed_logprobs=torch.arange(tot - len(lens)).float(),
        rewards=torch.arange(len(lens)).float()))


def test_sequence_sample_gather_split_json():
    from areal_vllm_amd.system.data_api import SequenceSample, SequenceSplitSpec

    s = _sample(["a", "b", "c", "d"], [3, 5, 2, 4])
    assert s.seqlens["packed_logprobs"] == [[2], [4], [1], [3]] and s.seqlens["rewards"] == [[1]] * 4
    parts = s.split_with_spec(SequenceSplitSpec(sizes=[1, 3]))
    assert parts[1].ids == ["b", "c", "d"] and parts[1].data["packed_input_ids"].tolist() == list(range(3, 14))
    back = SequenceSample.gather(parts)
    assert back.ids == s.ids and torch.equal(back.data["packed_logprobs"], s.data["packed_logprobs"])
    mbs, fwd, bwd = s.split(MicroBatchSpec(max_tokens_per_mb=8))
    assert all(sum(m._item_tokens("packed_input_ids")) <= 8 for m in mbs)
    re = SequenceSample.gather(mbs).reorder(bwd)
    assert re.ids == s.ids
    js = json.loads(json.dumps(s.as_json_compatible()))
    s2 = SequenceSample.from_json_compatible(js)
    assert s2.ids == s.ids and torch.equal(s2.data["packed_input_ids"], s.data["packed_input_ids"])
    assert s2.data["packed_input_ids"].dtype == torch.int64
    s.remap_keys_({"rewards": "scores"})
    assert "scores" in s.keys and "rewards" not in s.keys
    meta = s.meta()
    assert meta.data is None and meta.seqlens["scores"] == [[1]] * 4
    with pytest.raises(AssertionError):
        SequenceSample(keys={"x"}, ids=["a", "a"], seqlens=dict(x=[[1], [1]]))


def test_dfg_edges_and_buffer_readiness():
    from areal_vllm_amd.system.buffer import AsyncIOSequenceBuffer
    from areal_vllm_amd.system.dfg import MFCDef, ModelInterfaceAbstraction, ModelInterfaceType, build_graph, \
        topological_levels

    itf = ModelInterfaceAbstraction("null")
    gen = MFCDef("gen", 2, ModelInterfaceType.GENERATE, itf, "actor", input_keys=("packed_prompts",),
                 output_keys=("packed_input_ids",))
    rew = MFCDef("rew", 2, ModelInterfaceType.INFERENCE, itf, "reward", input_keys=("packed_input_ids",),
                 output_keys=("rewards",))
    ref = MFCDef("ref", 2, ModelInterfaceType.INFERENCE, itf, "ref", input_keys=("packed_input_ids",),
                 output_keys=("ref_logp",))
    trn = MFCDef("train", 2, ModelInterfaceType.TRAIN_STEP, itf, "actor",
                 input_keys=("packed_input_ids", "rewards", "ref_logp"))
    G = build_graph([gen, rew, ref, trn])
    assert topological_levels(G) == [["gen"], ["ref", "rew"], ["train"]]
    assert gen.is_src and trn.is_dst and not gen.is_dst_of_model_role("actor") and trn.is_dst_of_model_role("actor")
    assert G.graph["dataset_keys"] == ["packed_prompts"]
    with pytest.raises(AssertionError):
        build_graph([gen, MFCDef("gen2", 2, ModelInterfaceType.GENERATE, itf, "actor", output_keys=("packed_input_ids",))])

    async def run():
        from areal_vllm_amd.system.data_api import SequenceSample

        buf = AsyncIOSequenceBuffer([gen, rew, ref, trn], max_size=8)
        items = [SequenceSample(keys={"packed_prompts"}, ids=[f"q{i}"], seqlens=dict(packed_prompts=[[3]]))
                 for i in range(3)]
        await buf.put_batch(items)
        assert buf.n_ready("gen") == 3 and buf.n_ready("train") == 0
        ids, s = await buf.get_batch_for_rpc(gen)
        assert ids == ["q0", "q1"]  # oldest first
        with pytest.raises(asyncio.TimeoutError):
            await buf.get_batch_for_rpc(rew, timeout=0.05)
        await buf.amend_batch(ids, [SequenceSample(keys={"packed_input_ids"}, ids=[i],
                                                   seqlens=dict(packed_input_ids=[[7]])) for i in ids])
        ids2, s2 = await buf.get_batch_for_rpc(rew)
        assert ids2 == ids and s2.keys == {"packed_input_ids"}
        await buf.get_batch_for_rpc(ref)
        await buf.amend_batch(ids, [SequenceSample(keys={"rewards", "ref_logp"}, ids=[i],
                                                   seqlens=dict(rewards=[[1]], ref_logp=[[6]])) for i in ids])
        await buf.get_batch_for_rpc(trn)
        assert buf.size == 1  # consumed by every call -> gone; q2 still waits for generation

    asyncio.run(run())


# ------------------------------------------------------------------------------- streams
def test_push_pull_grouping_and_request_reply_handshake():
    from areal_vllm_amd.system.streams import (NameResolvingPuller, NameResolvingPusher, NameResolvingReplyServer,
                                               NameResolvingRequestClient, NoMessage, Payload, grouping)

    assert grouping(5, 2) == {0: [0, 1, 2], 1: [3, 4]}
    pullers = [NameResolvingPuller("e", "t", i) for i in range(2)]
    pushers = [NameResolvingPusher("e", "t", i, 5) for i in range(5)]
    for i, p in enumerate(pushers):
        p.push(dict(who=i))
    got0 = sorted(pullers[0].pull(2000)["who"] for _ in range(3))
    got1 = sorted(pullers[1].pull(2000)["who"] for _ in range(2))
    assert got0 == [0, 1, 2] and got1 == [3, 4]
    [p.close() for p in pushers + pullers]

    servers = [NameResolvingReplyServer("e", "t", "master", handler=i) for i in range(2)]
    stop = threading.Event()

    def serve(s):
        while not stop.is_set():
            try:
                p = s.poll(block=True, timeout=0.05)
            except NoMessage:
                continue
            s.post(Payload(handler="master", handle_name=p.handle_name, request_id=p.request_id,
                           data=(s.handler, p.data, p.pre_hooks)))

    ts = [threading.Thread(target=serve, args=(s,), daemon=True) for s in servers]
    [t.start() for t in ts]
    client = NameResolvingRequestClient("e", "t", "master", [0, 1])
    assert client.call([0, 1], "echo", ["x", "y"]) == [(0, "x", []), (1, "y", [])]
    ids = client.request([0, 1], "echo", [1, 2], no_syn=False, pre_hooks=["data_transfer"])  # syn/ack round
    assert client.gather(ids) == [(0, 1, ["data_transfer"]), (1, 2, ["data_transfer"])]
    stop.set()
    [t.join(2) for t in ts]
    client.close()
    [s.close() for s in servers]


def test_worker_control_lifecycle():
    from areal_vllm_amd.system.worker_base import (PollResult, Worker, WorkerControlPanel, WorkerInformation,
                                                   WorkerServer, WorkerServerStatus)

    class Counter(Worker):
        def _configure(self, config):
            self.n = 0
            self.step = config["step"]

        def _poll(self):
            self.n += self.step
            if self.n >= 5 * self.step:
                self.exit()
            return PollResult(1, 1)

    server = WorkerServer("e", "t", "toy/0")
    w = Counter(server)
    th = threading.Thread(target=w.run, daemon=True)
    th.start()
    panel = WorkerControlPanel("e", "t")
    panel.connect(["toy/0"])
    assert panel.request("toy/0", "ping") == "pong"
    assert panel.get_worker_status("toy/0") == WorkerServerStatus.READY
    panel.request("toy/0", "configure", worker_info=WorkerInformation("e", "t", "toy", 0, 1), config=dict(step=3))
    panel.group_request("start")
    th.join(10)
    assert not th.is_alive() and w.n == 15
    assert panel.get_worker_status("toy/0") == WorkerServerStatus.COMPLETED
    from areal_vllm_amd.system.worker_base import WorkerException

    with pytest.raises(WorkerException):
        panel2 = WorkerControlPanel("e", "t")
        server2 = WorkerServer("e", "t", "toy/1")
        t2 = threading.Thread(target=Counter(server2).run, daemon=True)
        t2.start()
        panel2.connect(["toy/1"])
        try:
            panel2.request("toy/1", "no_such_command")
        finally:
            panel2.request("toy/1", "exit")
            t2.join(5)
    panel.close()
    server.close()


# ------------------------------------------------------- generation-server manager + mocks
class MockServer:
    """Speaks the rollout server's control verbs and a deterministic /generate."""

    def __init__(self):
        self.version = 0
        self.n_generate = 0
        self.events = []
        self.used_tokens = 0.0
        self.loop = asyncio.new_event_loop()
        self.started = threading.Event()
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()
        self.started.wait(10)

    def _run(self):
        asyncio.set_event_loop(self.loop)
        app = web.Application()

        async def ok(name):
            self.events.append(name)
            return web.json_response(dict(success=True))

        async def upd(req):
            body = await req.json()
            self.version = int(body["version"])
            self.events.append(("update", body["model_path"]))
            return web.json_response(dict(success=True))

        async def gen(req):
            body = await req.json()
            self.n_generate += 1
            n = int(body["sampling_params"]["max_new_tokens"])
            base = len(body["input_ids"])
            toks = [100 + base + i for i in range(n)]
            stop = base + n >= 14  # "EOS" once the sequence is 14 tokens long
            return web.json_response(dict(output_ids=toks, meta_info=dict(
                finish_reason=dict(type="stop" if stop else "length"),
                output_token_logprobs=[(-0.5, t, None) for t in toks], weight_version=self.version)))

        async def metrics(req):
            return web.Response(text=f"# TYPE x gauge\nsglang:num_used_tokens{{model_name=\"m\"}} {self.used_tokens}\n")

        app.router.add_post("/pause_generation", lambda r: ok("pause"))
        app.router.add_post("/continue_generation", lambda r: ok("continue"))
        app.router.add_post("/update_weights_from_disk", upd)
        app.router.add_post("/generate", gen)
        app.router.add_get("/metrics", metrics)
        runner = web.AppRunner(app)
        self.loop.run_until_complete(runner.setup())
        from areal_vllm_amd.utils.network import find_free_ports

        self.port = find_free_ports(1)[0]
        self.loop.run_until_complete(web.TCPSite(runner, "127.0.0.1", self.port).start())
        self.started.set()
        self.loop.run_forever()

    @property
    def addr(self):
        return f"127.0.0.1:{self.port}"

    def close(self):
        self.loop.call_soon_threadsafe(self.loop.stop)
        self.thread.join(3)


def _manager(tmp_path, n_servers=2, **kw):
    from areal_vllm_amd.system.gserver_manager import GserverManager, GserverManagerConfig
    from areal_vllm_amd.system.worker_base import ExpStatus, WorkerInformation

    servers = [MockServer() for _ in range(n_servers)]
    for s in servers:
        name_resolve.add_subentry(names.gen_servers("e", "t"), s.addr)
    name_resolve.add(names.experiment_status("e", "t"), ExpStatus.RUNNING.value, replace=True)
    cfg = GserverManagerConfig(model_name="actor", n_servers=n_servers, realloc_root=str(tmp_path), poll_interval=0.0,
                               **kw)
    m = GserverManager()
    m.configure(WorkerInformation("e", "t", "gserver_manager", 0, 1), cfg)
    m._poll()
    return m, servers


def _post(addr, route, payload):
    import requests

    r = requests.post(f"http://{addr}{route}", json=payload, timeout=10)
    r.raise_for_status()
    return r.json()


def test_gserver_manager_routing_staleness_and_weight_rollout(tmp_path):
    bs, eta = 4, 1
    m, servers = _manager(tmp_path, schedule_policy="round_robin", train_batch_size=bs, max_head_offpolicyness=eta,
                          max_concurrent_rollouts=100)
    try:
        a = m.manager_addr
        meta = dict(qid="q0", prompt_len=5, group_size=2, new_token_budget=8, previous_server_url="",
                    previous_version=-1)
        u0 = _post(a, "/schedule_request", meta)
        u1 = _post(a, "/schedule_request", dict(meta, qid="q1"))
        assert {u0["url"], u1["url"]} == {f"http://{s.addr}" for s in servers} and u0["version"] == 0
        # sticky while the version is unchanged
        again = _post(a, "/schedule_request", dict(meta, previous_server_url=u0["url"], previous_version=0))
        assert again["url"] == u0["url"]
        # staleness gate: with nothing trained, exactly (1 + eta) * bs rollouts may start
        granted = [_post(a, "/allocate_rollout", dict(qid=f"r{i}"))["success"] for i in range(3 * bs)]
        assert sum(granted) == (1 + eta) * bs and not granted[-1]
        for i in range(sum(granted)):
            _post(a, "/finish_rollout", dict(qid=f"r{i}", accepted=True, n_tokens=10))
        assert not _post(a, "/allocate_rollout", dict(qid="x"))["success"]  # finished samples still count
        # the trainer consumes a batch and publishes version 1: servers are paused, updated, resumed
        name_resolve.add(names.training_samples("e", "t"), str(bs), replace=True)
        name_resolve.add(names.model_version("e", "t", "actor"), "1", replace=True)
        os.makedirs(os.path.join(str(tmp_path), "actor", "1"))
        m._poll()
        for s in servers:
            assert s.version == 1 and s.events[:1] == ["pause"] and s.events[-1] == "continue"
            assert s.events[1] == ("update", os.path.join(str(tmp_path), "actor", "1"))
        assert _post(a, "/get_model_version", {})["version"] == 1
        more = [_post(a, "/allocate_rollout", dict(qid=f"s{i}"))["success"] for i in range(2 * bs)]
        assert sum(more) == bs  # one more batch per version
        moved = _post(a, "/schedule_request", dict(meta, previous_server_url=u0["url"], previous_version=0))
        assert moved["version"] == 1  # new weights: the sticky route is dropped
        # capacity gate
        m.config.max_concurrent_rollouts = m.rollout_stat.running
        res = _post(a, "/allocate_rollout", dict(qid="cap"))
        assert not res["success"] and "capacity" in res["reason"]
    finally:
        m._exit_hook(None)
        [s.close() for s in servers]


def test_gserver_manager_least_token_usage_and_gc(tmp_path):
    m, servers = _manager(tmp_path, schedule_policy="least_token_usage", train_batch_size=100)
    try:
        servers[0].used_tokens, servers[1].used_tokens = 5000.0, 10.0
        m._poll()
        meta = dict(qid="q0", prompt_len=5, group_size=1, new_token_budget=8, previous_server_url="", previous_version=-1)
        assert _post(m.manager_addr, "/schedule_request", meta)["url"] == f"http://{servers[1].addr}"
        for v in (1, 2, 3):
            os.makedirs(os.path.join(str(tmp_path), "actor", str(v)))
            name_resolve.add(names.model_version("e", "t", "actor"), str(v), replace=True)
            m._poll()
        assert sorted(os.listdir(os.path.join(str(tmp_path), "actor"))) == ["2", "3"]  # current + previous kept
    finally:
        m._exit_hook(None)
        [s.close() for s in servers]


def test_partial_rollout_chunks_and_agents(tmp_path):
    from areal_vllm_amd.impl.agent.math_single_step_agent import MathSingleStepAgent
    from areal_vllm_amd.impl.environment.toy_env import TokenParityEnv
    from areal_vllm_amd.system.data_api import SequenceSample
    from areal_vllm_amd.system.partial_rollout import PartialRolloutManager

    m, servers = _manager(tmp_path, n_servers=1, train_batch_size=100)

    async def run():
        req, rep = asyncio.Queue(), asyncio.Queue()
        prm = PartialRolloutManager(0, req, rep, new_tokens_per_chunk=3, stop_token_ids=[1], timeout=30,
                                    experiment_name="e", trial_name="t", gserver_manager_addr=m.manager_addr)
        g = GenerationHyperparameters(n_samples=2, max_new_tokens=8)
        agent = MathSingleStepAgent(g, answer_save_path=str(tmp_path / "ans"), reward_scaling=2.0)
        prompt = SequenceSample(keys={"packed_prompts"}, ids=["q7"], seqlens=dict(packed_prompts=[[5]]),
                                data=dict(packed_prompts=torch.tensor([11, 12, 13, 14, 15])))
        act = asyncio.Queue()
        task = asyncio.create_task(agent.collect_trajectory(prompt, TokenParityEnv(), req, act))
        while not task.done():
            await prm.run_step()
            while not rep.empty():
                act.put_nowait(rep.get_nowait())
        await prm.close()
        return task.result()

    try:
        (traj,) = asyncio.run(run())
        # 8 new tokens in chunks of 3 -> 3 calls per answer, 2 answers
        assert servers[0].n_generate == 6
        assert traj.seqlens["packed_input_ids"] == [[13, 13]] and traj.seqlens["packed_logprobs"] == [[12, 12]]
        assert traj.data["seq_no_eos_mask"].tolist() == [True, True]  # budget exhausted before the mock's EOS
        assert traj.data["prompt_mask"].view(2, 13)[:, :5].all() and not traj.data["prompt_mask"].view(2, 13)[:, 5:].any()
        lp = traj.data["packed_logprobs"].view(2, 12)
        assert (lp[:, :4] == 0).all() and (lp[:, 4:] == -0.5).all()
        assert set(traj.data["rewards"].tolist()) <= {-2.0, 2.0}
        assert traj.data["version_start"].tolist() == [0, 0]
        assert os.listdir(tmp_path / "ans" / "0") == ["q7.txt", "q7.jsonl"] or len(os.listdir(tmp_path / "ans" / "0")) == 2
    finally:
        m._exit_hook(None)
        [s.close() for s in servers]


def test_multi_task_reward_interface_and_function_call_client():
    from areal_vllm_amd.impl.interface.math_rw_interface import MultiTaskRewardInterface
    from areal_vllm_amd.system.dfg import ModelName
    from areal_vllm_amd.system.model_api import Model

    id2info = {"q1": dict(query_id="q1", task="math", solutions=["\\boxed{42}"])}

    class Tok:
        def decode(self, ids, **kw):
            return "so the answer is \\boxed{42}" if 9 in ids else "no idea"

    itf = MultiTaskRewardInterface(id2info=id2info, output_scaling=5.0)
    from areal_vllm_amd.system.data_api import SequenceSample

    x = SequenceSample(keys={"packed_input_ids", "prompt_mask"}, ids=["q1"],
                       seqlens=dict(packed_input_ids=[[4, 3]], prompt_mask=[[4, 3]]),
                       data=dict(packed_input_ids=torch.tensor([5, 5, 9, 9, 5, 7, 7]),
                                 prompt_mask=torch.tensor([1, 1, 0, 0, 1, 0, 0]).bool()))
    out = itf.inference(Model(name=ModelName("reward"), module=None, tokenizer=Tok()), x, None)
    assert out.data["rewards"].tolist() == [5.0, -5.0] and out.seqlens["rewards"] == [[1, 1]]


def test_quickstart_builds_experiment_from_overrides(tmp_path):
    from areal_vllm_amd.apps.quickstart import build_experiment

    y = tmp_path / "exp.yaml"
    y.write_text("layout: d2t2\nactor:\n  path: qwen2-1.5b\nppo:\n  kl_ctl: 0.05\n  gen:\n    max_new_tokens: 512\n"
                 "train_bs_n_seqs: 64\nmode: slurm\n")
    exp, launcher = build_experiment("async-ppo-math", str(y), ["group_size=8", "max_head_offpolicyness=4",
                                                                "recover_mode=auto"])
    assert exp.layout == "d2t2" and exp.actor.path == "qwen2-1.5b" and exp.ppo.gen.max_new_tokens == 512
    assert exp.group_size == 8 and exp.max_head_offpolicyness == 4
    assert launcher == dict(mode="slurm", recover_mode="auto", recover_retries=1, platform="")
    cfg = exp.initial_setup().lazy_init()
    assert len(cfg.model_worker) == 4 and cfg.gserver_manager[0].max_head_offpolicyness == 4
    assert cfg.gserver_manager[0].train_batch_size == 64
    with pytest.raises(KeyError):
        build_experiment("async-ppo-math", None, ["no_such_field=1"])
    with pytest.raises(KeyError):
        build_experiment("no-such-experiment")


def test_automatic_evaluator_orders_and_resumes(tmp_path):
    import sys

    from areal_vllm_amd.evaluation import pass_at_k
    from areal_vllm_amd.evaluation.evaluator import AutomaticEvaluator, AutomaticEvaluatorConfig, EvaluationStepStatus

    assert pass_at_k(8, 0, 4) == 0.0 and pass_at_k(8, 8, 1) == 1.0 and abs(pass_at_k(8, 2, 4) - 11 / 14) < 1e-9
    ck, ev = tmp_path / "ckpt", tmp_path / "eval"
    for g in (2, 4):
        (ck / f"epoch1epochstep{g}globalstep{g}").mkdir(parents=True)
    (ck / "recover_checkpoint").mkdir()
    logged = []

    def cmd(step):  # stand-in for the evaluation job: slow for step 2 so that step 4 finishes first
        delay = 0.6 if step.global_step == 2 else 0.0
        code = (f"import json,time; time.sleep({delay}); "
                f"json.dump(dict(metrics={{'mean@1': {step.global_step} / 10}}), open(r'{step.output_path}', 'w'))")
        return [sys.executable, "-c", code]

    e = AutomaticEvaluator(AutomaticEvaluatorConfig(max_concurrent_jobs=2), str(ck), str(ev),
                           log_fn=lambda m, g: logged.append((g, m)), cmd_builder=cmd)
    deadline = time.time() + 30
    while (e.pending() or not e.steps) and time.time() < deadline:
        e.step()
        time.sleep(0.05)
    assert logged == [(2, {"eval/mean@1": 0.2}), (4, {"eval/mean@1": 0.4})]  # step order, despite finish order
    assert sorted(e.steps) == [2, 4]  # the recover checkpoint is not a numbered checkpoint
    (ck / "epoch2epochstep2globalstep6").mkdir()
    e2 = AutomaticEvaluator(AutomaticEvaluatorConfig(), str(ck), str(ev), cmd_builder=cmd)
    assert e2.steps[2].status == EvaluationStepStatus.LOGGED and e2.steps[4].status == EvaluationStepStatus.LOGGED
    while (e2.pending() or 6 not in e2.steps) and time.time() < deadline:
        e2.step()
        time.sleep(0.05)
    rows = [json.loads(l) for l in open(ev / "eval_results.jsonl")]
    assert [r["global_step"] for r in rows] == [2, 4, 6]


def test_multi_turn_agent_discounts_rewards_and_masks_feedback():
    from areal_vllm_amd.impl.agent.math_multi_turn_agent import MathMultiTurnAgent
    from areal_vllm_amd.system.agent_api import EnvironmentService
    from areal_vllm_amd.system.data_api import SequenceSample
    from areal_vllm_amd.system.model_api import BundledGenerationOutputs

    class Env(EnvironmentService):
        def __init__(self):
            self.calls = 0

        async def step(self, action):
            self.calls += 1
            return None, [self.calls == 3], True, False, {}  # right on the third attempt

    async def run():
        agent = MathMultiTurnAgent(GenerationHyperparameters(max_new_tokens=4), num_turns=3, turn_level_discount=0.5)
        obs, act = asyncio.Queue(), asyncio.Queue()
        prompt = SequenceSample(keys={"packed_prompts"}, ids=["q1"], seqlens=dict(packed_prompts=[[3]]),
                                data=dict(packed_prompts=torch.tensor([10, 11, 12])))
        task = asyncio.create_task(agent.collect_trajectory(prompt, Env(), obs, act))
        seen = []
        for turn in range(3):
            qid, ids, g = await obs.get()
            seen.append((qid, list(ids)))
            out = [50 + turn, 60 + turn]
            act.put_nowait(BundledGenerationOutputs(qid=qid, prompt_ids=list(ids), output_ids=[out], seqs=[list(ids) + out],
                                                    logprobs=[[0.0] * (len(ids) - 1) + [-0.1, -0.2]], no_eos=[False],
                                                    version_start=[turn], version_end=[turn]))
        return seen, await task

    seen, (traj,) = asyncio.run(run())
    assert [q for q, _ in seen] == ["q1-0", "q1-1", "q1-2"] and g_ok(seen)
    # rewards -1, -1, +1 discounted backwards with 0.5: r1 = -1 + 0.5 * 1 = -0.5, r0 = -1 + 0.5 * r1 = -1.25
    assert traj.data["rewards"].tolist() == [-1.25, -0.5, 1.0]
    assert traj.seqlens["packed_input_ids"] == [[5, 9, 13]]
    pm = traj.data["prompt_mask"].tolist()
    assert pm[:5] == [1, 1, 1, 0, 0] and pm[5:14] == [1] * 7 + [0, 0]  # earlier turns + feedback are context
    assert traj.data["version_start"].tolist() == [0, 1, 2]


def g_ok(seen):
    # every turn's context = previous sequence + the (tokenizer-free) "wrong answer" feedback ids [5, 6]
    return seen[1][1] == seen[0][1] + [50, 60, 5, 6] and seen[2][1] == seen[1][1] + [51, 61, 5, 6]


def test_paired_dataset_feeds_dpo_loss(tmp_path):
    import json

    from areal_vllm_amd.impl.dataset import make_dataset
    from areal_vllm_amd.impl.utils.dpo_functional import dpo_loss
    from areal_vllm_amd.system.system_api import DatasetAbstraction

    rows = [dict(id=str(i), prompt_ids=[5, 6, 7], pos_answer_ids=[[10 + i], [11, 12]], neg_answer_ids=[[20], [21, 22, 23]])
            for i in range(6)]
    p = tmp_path / "pairs.jsonl"
    p.write_text("\n".join(json.dumps(r) for r in rows))
    ds = make_dataset(DatasetAbstraction("rw_pair", args=dict(dataset_path=str(p), max_pairs_per_prompt=2,
                                                              eos_token_id=1)), seed=1, dp_rank=0, world_size=2)
    assert len(ds) == 3
    item = ds[0]
    lens = item.seqlens["packed_input_ids"][0]
    assert len(lens) == 4 and int(item.data["prompt_lens"][0]) == 3
    flat = item.data["packed_input_ids"].tolist()
    off = 0
    for j, n in enumerate(lens):  # chosen / rejected alternate and share the prompt
        seq = flat[off:off + n]
        assert seq[:3] == [5, 6, 7] and seq[-1] == 1 and (seq[3] < 20) == (j % 2 == 0)
        off += n
    with pytest.raises(RuntimeError):
        make_dataset(DatasetAbstraction("rw_pair", args=dict(dataset_builder=lambda: [
            dict(id="x", prompt_ids=[1], pos_answer_ids=[[2]], neg_answer_ids=[])])), seed=1, dp_rank=0, world_size=1)
    pi = torch.tensor([-1.0, -3.0, -2.0, -2.5], requires_grad=True)
    ref = torch.tensor([-2.0, -2.0, -2.0, -2.0])
    loss, pos, neg, kl = dpo_loss(pi, ref, beta=0.5)
    want = -torch.nn.functional.logsigmoid(torch.tensor([0.5 * 2.0, 0.5 * 0.5])).mean()
    assert float(loss) == pytest.approx(float(want)) and float(pos) == pytest.approx(0.5) and float(neg) == pytest.approx(-0.75)
    loss.backward()
    assert pi.grad[0] < 0 < pi.grad[1]  # raise the chosen log-prob, lower the rejected one


def test_experiment_checks_and_cluster_spec(tmp_path):
    import json

    from areal_vllm_amd.api.alloc_mode import ParallelStrategy
    from areal_vllm_amd.base.cluster import ClusterSpec, load_spec_from_file
    from areal_vllm_amd.experiments import check
    from areal_vllm_amd.experiments.common import ModelTrainEvalConfig

    ok = ModelTrainEvalConfig(path="tiny", init_from_scratch=True)
    check.check_experiment({"actor": ok}, ParallelStrategy(data_parallel_size=2), 8)
    with pytest.raises(ValueError):
        check.check_valid_model_and_path("actor", ModelTrainEvalConfig(path="tiny", init_from_scratch=False))
    with pytest.raises(FileNotFoundError):
        check.check_valid_model_and_path("actor", ModelTrainEvalConfig(path="/nonexistent/ckpt"))
    with pytest.raises(ValueError):
        check.check_valid_parallel_batch_size("b", 4, ParallelStrategy(data_parallel_size=2, pipeline_parallel_size=2))
    with pytest.raises(ValueError):
        check.check_valid_rollout_server(0.9, colocated_with_trainer=True)
    bad = ModelTrainEvalConfig(path="tiny", init_from_scratch=True)
    bad.optimizer.lr = 0.0
    with pytest.raises(ValueError):
        check.check_valid_optimizer("actor", bad)
    f = tmp_path / "cluster.json"
    f.write_text(json.dumps(dict(cluster_name="c", n_nodes=3, n_gpus_per_node=8, node_name_prefix="mi", junk=1)))
    spec = load_spec_from_file(str(f))
    assert spec.n_gpus == 24 and spec.node_names() == ["mi01", "mi02", "mi03"] and ClusterSpec().gpu_memory_gb == 288


def test_codeforces_elo_estimate_on_a_synthetic_contest(tmp_path):
    """A virtual participant that solves what the top 10% of a 400-player field solved gets ranked there and
    is assigned the rating whose expected rank equals that rank; a model that solves nothing lands at the
    bottom; contests with too few rated participants are skipped."""
    import json

    import numpy as np

    from areal_vllm_amd.evaluation import cf_elo

    rng = np.random.default_rng(0)
    n = 400
    ratings = np.sort(rng.normal(1500, 350, n))[::-1].astype(int)  # standings order = rating order
    problems = [dict(contestId=1900, index=c, points=p) for c, p in zip("ABCD", (500, 1000, 1500, 2000))]
    pts = [5000 if i < 40 else 3000 if i < 120 else 1500 if i < 300 else 500 for i in range(n)]
    rows = [dict(party=dict(members=[dict(handle=f"u{i}")]), points=pts[i], penalty=0) for i in range(n)]
    changes = [dict(handle=f"u{i}", oldRating=int(ratings[i])) for i in range(n)]
    contest = dict(standings=dict(status="OK", result=dict(problems=problems, rows=rows)),
                   rating_changes=dict(status="OK", result=changes))
    small = dict(standings=dict(status="OK", result=dict(problems=problems, rows=rows[:50])),
                 rating_changes=dict(status="OK", result=changes[:50]))
    contests = {"1900": contest, "1901": small}
    hist = {str(r): 1 for r in range(800, 2600, 10)}
    ids = ["1900A", "1900B", "1900C", "1900D", "1901A"]

    # all four solved, D on the second attempt: 500 + 1000 + 1500 + 1950 = 4950 points -> behind the 40 with 5000
    assert cf_elo.virtual_score(problems, {"1900A": [True], "1900D": [False, True]}, None) == (2450.0, 0)
    strong = cf_elo.estimate_elo(ids, [[True], [True], [True], [False, True], [True]], contests,
                                 sorted(float(k) for k in hist), pass_n=None)
    weak = cf_elo.estimate_elo(ids, [[False]] * 5, contests, sorted(float(k) for k in hist), pass_n=1)
    assert strong["contests_processed"] == 1 and strong["skipped_contest_ids"] == [1901]
    r = strong["estimated_rating"]
    assert ratings[40] <= r <= ratings[0] + 100, (r, ratings[40])  # rank 40 of 400 (win probabilities are smooth)
    seed = 1 + np.sum(1 / (1 + 10 ** ((r - ratings) / 400)))
    assert abs(seed - 40) < 1.0
    assert weak["estimated_rating"] < ratings[-40] and strong["estimated_percentile"] > weak["estimated_percentile"]
    # ICPC-style contests (no points): solved count, 10 penalty minutes per failed attempt
    assert cf_elo.virtual_score([dict(contestId=7, index="A"), dict(contestId=7, index="B")],
                                {"7A": [False, False, True], "7B": [False]}) == (1.0, 20)

    for name, obj in (("meta.json", ids), ("hist.json", hist), ("contests.json", contests)):
        (tmp_path / name).write_text(json.dumps(obj))
    again = cf_elo.estimate_elo_from_files(
        [dict(idx=i, score=s) for i, s in enumerate([[1], [1], [1], [0, 1], [1]])], str(tmp_path / "meta.json"),
        str(tmp_path / "hist.json"), str(tmp_path / "contests.json"), pass_n=None)
    assert again["estimated_rating"] == strong["estimated_rating"]


def test_python_executor_and_result_table():
    from areal_vllm_amd.evaluation.eval_and_aggregate import PROMPT_TYPES, format_table
    from areal_vllm_amd.evaluation.python_executor import PythonExecutor, extract_program

    ex = PythonExecutor(timeout=5.0)
    text = "first\n```python\nprint(1)\n```\nthen\n```python\nimport sys\nprint(int(sys.stdin.read()) * 6)\n```"
    assert extract_program(text) == "import sys\nprint(int(sys.stdin.read()) * 6)\n"
    assert ex.run(extract_program(text), stdin="7") == ("42", "Done")
    out = ex.batch_run(["print(sum(range(10)))", "raise ValueError('boom')", "while True: pass"])
    assert out[0] == ("45", "Done") and "ValueError: boom" in out[1][1] and out[2][1] == "Timeout Error"
    assert PythonExecutor(get_answer_from="solution()").run("def solution():\n    return 3 * 4")[0] == "12"
    table = format_table({"aime24": {"greedy_acc": 30.0, "pass@8": 61.25}, "codeforces": {"greedy_acc": 5.0, "cf_rating": 1234.4}})
    assert "aime24" in table and "61.2" in table and "1234" in table and "-" in table
    assert all("{prompt}" in t for t in PROMPT_TYPES.values())
    assert PROMPT_TYPES["r1-distilled-qwen"].format(prompt="2+2?").count("\\boxed{}") == 1


def test_best_of_k_keeps_the_highest_scored_longest_answers():
    from types import SimpleNamespace

    from areal_vllm_amd.impl.interface.ppo_interface import PPOActorInterface

    texts = {1: "so the answer is \\boxed{4}", 2: "I think \\boxed{5}", 3: "long wrong " * 3 + "\\boxed{7}",
             4: "it is clearly, after some thought, \\boxed{4}"}

    class Tok:
        def decode(self, ids, **kw):
            return texts[ids[0]]

    iface = PPOActorInterface(group_size=2, generation_size=4, generation_config=dict(n_samples=2, max_new_tokens=8),
                              id2info={"q0": dict(query_id="q0", task="math", solutions=["\\boxed{4}"])})
    group = [SimpleNamespace(output_tokens=[k] * n) for k, n in ((1, 3), (2, 9), (3, 12), (4, 6))]
    kept = iface._best_of(SimpleNamespace(tokenizer=Tok()), "q0", group)
    # the two correct answers first (the longer of them leading), wrong ones dropped however long they are
    assert [r.output_tokens[0] for r in kept] == [4, 1]
    with pytest.raises(AssertionError):
        PPOActorInterface(group_size=4, generation_size=2)


def test_generation_only_model_under_the_reference_backend_names(tmp_path):
    """A model worker's generation-only model (backend `vllm` / `sglang` in configurations written for the reference):
    the PPO interface's generate call and the reference's ``module.generate`` convention both run on the hosted
    rollout engine; weights arrive from a checkpoint directory."""
    from types import SimpleNamespace

    from areal_vllm_amd.api.cli_args import GenerationHyperparameters
    from areal_vllm_amd.api.io_struct import FinetuneSpec
    from areal_vllm_amd.impl.backend.native import GenerationEngineModule
    from areal_vllm_amd.impl.interface.ppo_interface import PPOActorInterface
    from areal_vllm_amd.models import hf_io
    from areal_vllm_amd.models.config import preset
    from areal_vllm_amd.models.transformer import Transformer
    from areal_vllm_amd.system.data_api import SequenceSample
    from areal_vllm_amd.system.model_api import (ALL_BACKEND_CLASSES, ModelAbstraction, ModelBackendAbstraction,
                                                 make_backend, make_model)

    model = make_model(ModelAbstraction("native", dict(path="tiny", init_from_scratch=True, dtype="float32")),
                       "actor_gen", "cpu")
    backend = make_backend(ModelBackendAbstraction("vllm", dict(max_num_seqs=8, max_model_len=256, hybrid_train=True)))
    assert ALL_BACKEND_CLASSES["sglang"] is ALL_BACKEND_CLASSES["vllm"]
    model = backend.initialize(model, FinetuneSpec(1, 8, 4))
    try:
        assert isinstance(model.module, GenerationEngineModule) and model.backend_name == "native_generation"
        prompts = [[5, 9, 11, 200, 7], [3, 4, 5]]
        data = SequenceSample.from_default(ids=["a", "b"], seqlens=[5, 3],
                                           data=dict(packed_prompts=torch.tensor(sum(prompts, []))))
        iface = PPOActorInterface(group_size=2, generation_config=dict(n_samples=2, max_new_tokens=6, greedy=True))
        out = iface.generate(model, data, None)
        lens = [x for g in out.seqlens["packed_input_ids"] for x in g]
        assert len(lens) == 4 and all(6 <= n <= 11 for n in lens)
        assert out.data["packed_logprobs"].numel() == sum(lens) - 4
        first = out.data["packed_input_ids"][:lens[0]].tolist()
        assert first[:5] == prompts[0]
        # the reference's convention: padded ids and log-probs of the generated part only
        tok = SimpleNamespace(eos_token_id=1, pad_token_id=0)
        ids, lps, mask = model.module.generate(data, None, tok, GenerationHyperparameters(n_samples=1, max_new_tokens=6,
                                                                                           greedy=True))
        assert ids.shape == lps.shape and ids.shape[0] == 2 and mask is None
        assert ids[0, :lens[0] - 5].tolist() == first[5:]
        # new weights from a checkpoint directory change what is generated
        other = Transformer(preset("tiny"), dtype=torch.float32, device="cpu")
        other.flatten()
        other.init_weights(seed=77)
        wm = hf_io.save_hf_checkpoint(other, str(tmp_path / "ckpt"))
        hf_io.finalize_hf_checkpoint(str(tmp_path / "ckpt"), other.cfg, [wm], None)
        backend.load(model, str(tmp_path / "ckpt"))
        ids2, _, _ = model.module.generate(data, None, tok, GenerationHyperparameters(n_samples=1, max_new_tokens=6,
                                                                                      greedy=True))
        assert ids2.tolist() != ids.tolist()
    finally:
        model.module.destroy()


# ------------------------------------------------------------------------ slurm resource allocation
SCONTROL = """NodeName=gpu01 Arch=x86_64 CoresPerSocket=64
   CPUAlloc=0 CPUTot=128 State=IDLE ThreadsPerCore=1
   CfgTRES=cpu=128,mem=2000G,billing=128,gres/gpu=8
   AllocTRES=

NodeName=gpu02 Arch=x86_64
   State=MIXED
   CfgTRES=cpu=128,mem=2000G,billing=128,gres/gpu:mi355x=8
   AllocTRES=cpu=32,mem=500G,gres/gpu:mi355x=2

NodeName=gpu03 Arch=x86_64
   State=IDLE+DRAIN
   CfgTRES=cpu=128,mem=2000G,gres/gpu=8
   AllocTRES=

NodeName=cpu01 Arch=x86_64
   State=IDLE
   CfgTRES=cpu=64,mem=256G,billing=64
   AllocTRES=cpu=8,mem=16G
"""


def test_slurm_free_resources_and_packed_allocation(tmp_path):
    from areal_vllm_amd.scheduler import slurm_utils as su

    nodes = su.parse_scontrol_nodes(SCONTROL)
    assert set(nodes) == {"gpu01", "gpu02", "cpu01"}  # the draining node is not offered
    assert nodes["gpu01"] == su.NodeResource(128, 2000 * 1024, 8)
    assert nodes["gpu02"] == su.NodeResource(96, 1500 * 1024, 6) and nodes["cpu01"] == su.NodeResource(56, 240 * 1024, 0)
    assert su.expand_nodelist("gpu[01-03,07],cpu1") == ["gpu01", "gpu02", "gpu03", "gpu07", "cpu1"]
    reqs = [su.LaunchRequest("master_worker", 1, su.NodeResource(4, 16000, 0)),
            su.LaunchRequest("model_worker", 12, su.NodeResource(8, 64000, 1)),
            su.LaunchRequest("rollout_worker", 4, su.NodeResource(2, 4000, 0)),
            su.LaunchRequest("generation_server", 2, su.NodeResource(8, 64000, 1))]
    placed, left = su.allocate_resources(reqs, nodes)
    by = {p.name: p for p in placed}
    # model workers 0-7 fill the emptiest node in order, 8-11 continue on the next: rank neighbours share a node
    assert by["model_worker"].hosts == ["gpu01"] * 8 + ["gpu02"] * 4
    assert by["model_worker"].local_ids == list(range(8)) + list(range(4))
    assert by["generation_server"].hosts == ["gpu02", "gpu02"] and left["gpu02"].gpu == 0 and left["gpu01"].gpu == 0
    assert len(set(by["rollout_worker"].hosts)) >= 2  # CPU-only arrays spread out
    assert all(v.valid() for v in left.values())
    assert by["model_worker"].multiprog("run -i {index} -l {local}").splitlines()[9] == "9 run -i 9 -l 1"
    with pytest.raises(su.SlurmResourceNotEnoughException, match="model_worker"):
        su.allocate_resources([su.LaunchRequest("model_worker", 15, su.NodeResource(8, 64000, 1))], nodes)
    only = su.allocate_resources([su.LaunchRequest("x", 2, su.NodeResource(1, 100, 1), nodelist="gpu[02-03]")], nodes)[0][0]
    assert only.hosts == ["gpu02", "gpu02"]


def test_slurm_client_pins_the_allocation_with_hostfile_and_multiprog(tmp_path):
    from areal_vllm_amd.scheduler import slurm_utils as su
    from areal_vllm_amd.scheduler.client import SlurmSchedulerClient

    nodes = su.parse_scontrol_nodes(SCONTROL)
    c = SlurmSchedulerClient("e", "t", fileroot=str(tmp_path), allocate=True, node_resources=nodes)
    script = c.render_allocated("model_worker", "python -m areal_vllm_amd.apps.remote worker -w model_worker -i {index}",
                                10, gpu=1, cpu=8, mem=64000, env_vars={"A": "1"})
    assert "#SBATCH --nodelist=gpu01,gpu02" in script and "#SBATCH --gres=gpu:8" in script
    assert "--distribution=arbitrary" in script and "export SLURM_HOSTFILE=" in script and "export A=1" in script
    d = c._dir()
    assert open(f"{d}/model_worker.hostfile").read().split() == ["gpu01"] * 8 + ["gpu02"] * 2
    mp = open(f"{d}/model_worker.multiprog").read().splitlines()
    assert mp[8].startswith("8 bash -c 'export LOCAL_RANK=0; python") and mp[8].rstrip().endswith("-i 8'")
    # the second array sees what the first one left
    c.render_allocated("generation_server", "serve {index}", 4, gpu=1, cpu=8, mem=64000)
    assert open(f"{d}/generation_server.hostfile").read().split() == ["gpu02"] * 4
    with pytest.raises(su.SlurmResourceNotEnoughException):
        c.render_allocated("one_more", "x", 1, gpu=1)


def test_engine_offload_parks_weights_and_optimizer_state_and_training_resumes(tmp_path):
    """`offload` frees the flat parameter / gradient / optimizer storages (contents parked in host memory), the
    next call brings them back, and training continues exactly as without the round trip."""
    import torch

    from areal_vllm_amd.api.cli_args import MicroBatchSpec, OptimizerConfig, TrainEngineConfig
    from areal_vllm_amd.api.io_struct import FinetuneSpec
    from areal_vllm_amd.engine.train_engine import NativeTrainEngine

    def make():
        cfg = TrainEngineConfig(experiment_name="t", trial_name="off", path="tiny", init_from_scratch=True,
                                dtype="float32", optimizer=OptimizerConfig(lr=1e-2, lr_scheduler_type="constant",
                                                                           warmup_steps_proportion=0.0),
                                mb_spec=MicroBatchSpec())
        e = NativeTrainEngine(cfg)
        e.create_process_group()
        e.initialize(None, FinetuneSpec(1, 8, 4))
        return e

    g = torch.Generator().manual_seed(0)
    ids = torch.randint(2, 500, (3, 12), generator=g)
    batch = dict(input_ids=ids, attention_mask=torch.ones_like(ids, dtype=torch.bool),
                 loss_mask=torch.ones_like(ids, dtype=torch.bool))

    def loss(proxy, mb):
        return -proxy.logprobs(torch.roll(mb["input_ids"], -1, 0)).mean()

    w = lambda mb: float(mb["input_ids"].numel())  # noqa: E731
    a = make()
    try:
        a.train_batch(batch, loss, w)
        before = a.model.flat.flat.clone()
        full = a.model.flat.flat.untyped_storage().size()
        freed = a.offload()
        assert a.offloaded and freed >= full * 2  # parameters + fp32 gradients + optimizer shards
        assert a.model.flat.flat.untyped_storage().size() == 0 and a.optimizer.m.untyped_storage().size() == 0
        assert a.offload() == 0  # idempotent
        s1 = a.train_batch(batch, loss, w)  # comes back by itself
        assert not a.offloaded and a.model.flat.flat.untyped_storage().size() == full
        after_a = a.model.flat.flat.clone()
        assert not torch.equal(before, after_a)
    finally:
        a.destroy()
    b = make()
    try:
        b.train_batch(batch, loss, w)
        assert torch.equal(b.model.flat.flat, before)
        s2 = b.train_batch(batch, loss, w)
        assert torch.equal(b.model.flat.flat, after_a) and s1["grad_norm"] == s2["grad_norm"]
    finally:
        b.destroy()


def test_offload_hooks_are_attached_to_roles_configured_with_offload():
    from areal_vllm_amd.experiments.common import ModelTrainEvalConfig
    from areal_vllm_amd.experiments.ppo_math_exp import PPOHyperparameters, PPOMATHExperiment
    from areal_vllm_amd.system.dfg import OffloadHook

    ppo = PPOHyperparameters(kl_ctl=0.1, disable_value=True)
    e = PPOMATHExperiment(actor=ModelTrainEvalConfig(path="tiny", init_from_scratch=True),
                          ref=ModelTrainEvalConfig(path="tiny", init_from_scratch=True, backend="inference", offload=True),
                          ppo=ppo, layout="d1", device="cpu")
    cfg = e.initial_setup()
    hooked = {r.name for r in cfg.model_rpcs if any(isinstance(h, OffloadHook) for h in r._post_hooks)}
    assert hooked == {"ref_inf"}
