"""Worker-runtime tests on CPU: planner / data manager units, then whole experiments (master + model
workers + generation server + manager + rollout workers as real sub-processes, gloo, tiny model).
Pattern of the reference's ``realhf/tests/experiments`` (run_test_exp) and ``tests/system``."""
import json
import os
import random
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------ units
def test_balanced_partition_and_planner():
    from areal_vllm_amd.system.redistributor import GlobalStorageTracker, RedistribPlanner
    from areal_vllm_amd.utils.datapack import balanced_partition

    parts = balanced_partition([5, 1, 1, 1, 9, 2], 2, equal_size=True)
    assert sorted(map(len, parts)) == [3, 3] and sorted(sum(parts, [])) == list(range(6))
    tr = GlobalStorageTracker(3)
    tr.add_data(0, ["a", "b"], ["x"])
    tr.add_data(1, ["c"], ["x"])
    tr.add_data(2, ["a", "b", "c"], ["y"])
    pl = RedistribPlanner(tr)
    plan = pl.derive_plan({0: ["a", "c"], 1: ["b", "c"]}, ["x", "y"])
    moved = {(s.src, d, s.key, i) for s in plan for d in s.dsts for i in s.ids}
    assert moved == {(1, 0, "x", "c"), (0, 1, "x", "b"), (2, 0, "y", "a"), (2, 0, "y", "c"), (2, 1, "y", "b"),
                     (2, 1, "y", "c")}
    pl.apply(plan)
    assert pl.derive_plan({0: ["a", "c"], 1: ["b", "c"]}, ["x", "y"]) == []
    with pytest.raises(RuntimeError):
        pl.derive_plan({0: ["zz"]}, ["x"])
    tr.clear_data(["a", "b", "c"])
    assert tr.n_items() == 0


def _dm_worker(rank, world, port, q):
    os.environ["GLOO_SOCKET_IFNAME"] = "lo"
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    from areal_vllm_amd.system.data_api import SequenceSample
    from areal_vllm_amd.system.data_manager import DataManager
    from areal_vllm_amd.system.redistributor import RedistribStep

    dm = DataManager(rank)
    ids = ["s0", "s1", "s2"]
    lens = [3, 5, 2]
    full = SequenceSample.from_default(ids=ids, seqlens=lens, data=dict(
        packed_input_ids=torch.arange(10), prompt_mask=torch.arange(10) % 2 == 0, rewards=torch.tensor([1., 2., 3.])))
    if rank == 0:
        dm.store(full)
    plan = [RedistribStep(src=0, dsts=[1, 2], key="packed_input_ids", ids=["s1", "s2"]),
            RedistribStep(src=0, dsts=[1], key="prompt_mask", ids=["s1"]),
            RedistribStep(src=0, dsts=[2], key="rewards", ids=["s0", "s2"])]
    dm.redistribute(full.meta(), plan)
    ok = True
    if rank == 1:
        x = dm.get(["s1", "s2"], ["packed_input_ids"])
        ok &= x.data["packed_input_ids"].tolist() == list(range(3, 10))
        ok &= dm.get(["s1"], ["prompt_mask"]).data["prompt_mask"].tolist() == [(i % 2 == 0) for i in range(3, 8)]
        ok &= dm.get(["s1"], ["prompt_mask"]).data["prompt_mask"].dtype == torch.bool
    if rank == 2:
        ok &= dm.get(["s0", "s2"], ["rewards"]).data["rewards"].tolist() == [1.0, 3.0]
        ok &= not dm.has_data("s1", "rewards")
    dm.clear_data(["s1"])
    ok &= not dm.has_data("s1")
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_data_manager_point_to_point_redistribution():
    from areal_vllm_amd.utils.network import find_free_ports

    port = find_free_ports(1)[0]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_dm_worker, args=(r, 3, port, q)) for r in range(3)]
    [p.start() for p in ps]
    res = dict(q.get(timeout=120) for _ in range(3))
    [p.join(30) for p in ps]
    assert res == {0: True, 1: True, 2: True}


def test_experiment_config_routing_tables():
    from areal_vllm_amd.experiments.async_ppo_math_exp import AsyncPPOMATHExperiment
    from areal_vllm_amd.experiments.ppo_math_exp import PPOHyperparameters, PPOMATHExperiment
    from areal_vllm_amd.system.dfg import ModelName, ParamReallocHook
    from areal_vllm_amd.system.system_api import ModelShardID

    e = PPOMATHExperiment(layout="d2t2", ppo=PPOHyperparameters(kl_ctl=0.1, disable_value=False, use_decoupled_loss=True))
    cfg = e.initial_setup().lazy_init()
    assert [r.name for r in cfg.model_rpcs] == ["actor_gen", "rew_inf", "ref_inf", "actor_inf", "critic_inf",
                                                "critic_train", "actor_train"]
    assert len(cfg.model_worker) == 4 and e.scheduling_setup().model_worker.count == 4
    mw = cfg.model_worker[3]
    assert mw.msid2mwid[ModelShardID(ModelName("critic"), 1, 1, 0)] == 3
    assert mw.model_topos[ModelName("actor")].tp_size == 2
    levels = {r.name: {p.name for p in r.parents} for r in cfg.model_rpcs}
    assert levels["actor_train"] >= {"actor_gen", "rew_inf", "ref_inf", "actor_inf", "critic_inf"}
    a = AsyncPPOMATHExperiment(n_generation_servers=2, n_rollout_workers=3)
    ac = a.initial_setup().lazy_init()
    assert [r.name for r in ac.model_rpcs] == ["actor_train"]
    assert any(isinstance(h, ParamReallocHook) for h in ac.model_rpcs[0]._post_hooks)
    assert len(ac.generation_server) == 2 and len(ac.rollout_worker) == 3 and ac.gserver_manager[0].n_servers == 2
    infos = ac.get_worker_infos("e", "t")
    assert infos["rollout_worker"][2].worker_index == 2 and infos["rollout_worker"][2].worker_count == 3


def test_slurm_scheduler_renders_worker_array(tmp_path):
    from areal_vllm_amd.scheduler.client import SlurmSchedulerClient, remote_worker_cmd

    c = SlurmSchedulerClient("e", "t", fileroot=str(tmp_path), container_type="apptainer", container_image="img.sif")
    s = c.render("model_worker", remote_worker_cmd("e", "t", "model_worker"), 16, gpu=1, cpu=8, mem=8192,
                 env_vars={"A": "1"}, time_limit="1:00:00")
    assert "--ntasks=16" in s and "--nodes=2" in s and "--gres=gpu:8" in s and "--time=1:00:00" in s
    assert "-i $SLURM_PROCID" in s and "LOCAL_RANK=$SLURM_LOCALID" in s and "apptainer exec --rocm img.sif" in s


# ------------------------------------------------------------------ whole experiments
def _actor():
    from areal_vllm_amd.api.cli_args import OptimizerConfig
    from areal_vllm_amd.experiments.common import ModelTrainEvalConfig

    return ModelTrainEvalConfig(path="tiny", init_from_scratch=True, dtype="float32", backend="mock_train",
                                optimizer=OptimizerConfig(lr=1e-3, lr_scheduler_type="constant",
                                                          warmup_steps_proportion=0.0))


def _stats(root, exp, trial):
    with open(os.path.join(root, "logs", exp, trial, "master_stats.jsonl")) as f:
        return [json.loads(l) for l in f]


@pytest.mark.slow
def test_sft_experiment_dp2_loss_decreases_and_recovers(tmp_path):
    from areal_vllm_amd.apps.main import main_start
    from areal_vllm_amd.experiments.sft_exp import SFTExperiment
    from areal_vllm_amd.system.system_api import DatasetAbstraction, ExperimentSaveEvalControl

    root = str(tmp_path)
    rng = random.Random(0)
    rows = [dict(id=str(i), prompt_ids=[rng.randint(2, 400) for _ in range(rng.randint(3, 10))],
                 answer_ids=[7, 8, 9, 10][: rng.randint(1, 4)]) for i in range(32)]
    p = os.path.join(root, "data.jsonl")
    with open(p, "w") as f:
        f.write("\n".join(json.dumps(r) for r in rows))
    ds = DatasetAbstraction("prompt_answer", args=dict(dataset_path=p, eos_token_id=1))

    def exp(epochs, benchmark_steps=None):
        return SFTExperiment(experiment_name="t-sft", trial_name="t0", model=_actor(), layout="d2", dataset=ds,
                             eval_dataset=ds, train_bs_n_seqs=8, device="cpu", fileroot=root,
                             exp_ctrl=ExperimentSaveEvalControl(total_train_epochs=epochs, ckpt_freq_steps=2,
                                                                eval_freq_epochs=1, benchmark_steps=benchmark_steps))

    # first run stops after 3 of 8 steps (as if it had been killed after the checkpoint of step 2)
    main_start(exp(2, benchmark_steps=3), "t-sft", "t0", fileroot=root, platform="cpu", timeout=300)
    st = _stats(root, "t-sft", "t0")
    assert [s["global_step"] for s in st] == [1, 2, 3]
    ck = os.path.join(root, "checkpoints", "t-sft", "t0", "default", "recover_checkpoint")
    assert os.path.exists(os.path.join(ck, "model.safetensors"))
    assert any(f.startswith("optim_world_size_2_rank_") for f in os.listdir(ck))
    with open(os.path.join(root, "recover", "t-sft", "t0", "recover_info.json")) as f:
        info = json.load(f)
    assert info["recover_start"]["global_step"] == 2 and len(info["hash_vals_to_ignore"]) == 16
    # the recover run resumes after the checkpointed step and skips the samples already used in the epoch
    main_start(exp(2), "t-sft", "t0", fileroot=root, platform="cpu", timeout=300, recover_run=True)
    st = _stats(root, "t-sft", "t0")
    assert [s["global_step"] for s in st] == [1, 2, 3, 3, 4, 5, 6, 7, 8]
    losses = [s["trainDefault/loss/avg"] for s in st]
    assert losses[-1] < 0.6 * losses[0], losses


@pytest.mark.slow
def test_sync_ppo_experiment_with_reference_and_critic(tmp_path):
    from areal_vllm_amd.api.cli_args import GenerationHyperparameters
    from areal_vllm_amd.apps.main import main_start
    from areal_vllm_amd.experiments.ppo_math_exp import PPOHyperparameters, PPOMATHExperiment
    from areal_vllm_amd.system.system_api import DatasetAbstraction, ExperimentSaveEvalControl

    root = str(tmp_path)
    rng = random.Random(0)
    rows = [dict(query_id=str(i), input_ids=[rng.randint(2, 400) for _ in range(rng.randint(3, 10))], task="math",
                 solutions=["\\boxed{1}"]) for i in range(16)]
    p = os.path.join(root, "data.jsonl")
    with open(p, "w") as f:
        f.write("\n".join(json.dumps(r) for r in rows))
    ppo = PPOHyperparameters(gen=GenerationHyperparameters(max_new_tokens=8), ppo_n_minibatches=2, kl_ctl=0.1,
                             use_decoupled_loss=True, disable_value=False)
    import dataclasses

    # the reference copy is parked in host memory between its calls (offload post-hook), the critic stays resident
    ref = dataclasses.replace(_actor(), backend="inference", offload=True)
    e = PPOMATHExperiment(experiment_name="t-ppo", trial_name="sync", actor=_actor(), ref=ref, ppo=ppo, layout="d2",
                          dataset=DatasetAbstraction("math_code_prompt", args=dict(dataset_path=p)),
                          reward_args=dict(dataset_path=p), train_bs_n_seqs=4, group_size=2, device="cpu", fileroot=root,
                          exp_ctrl=ExperimentSaveEvalControl(total_train_epochs=1, benchmark_steps=3))
    main_start(e, "t-ppo", "sync", fileroot=root, platform="cpu", timeout=400)
    st = _stats(root, "t-ppo", "sync")
    assert len(st) == 3
    for s in st:
        for k in ("actor_gen/time", "rew_inf/time", "ref_inf/time", "actor_inf/time", "critic_inf/time",
                  "critic_train/time", "actor_train/time"):
            assert k in s, (k, sorted(s))
        assert s["actor_train/head_offpolicyness"] == 0.0  # synchronous: always on-policy
        assert s["actor_train/update_successful"] == 1.0 and s["critic_train/update_successful"] == 1.0
    assert st[-1]["critic_train/grad_norm"] > 0
    logs = ""
    for base, _, files in os.walk(os.path.join(root, "logs")):
        for fn in files:
            if fn.startswith("model_worker"):
                with open(os.path.join(base, fn), errors="ignore") as f:
                    logs += f.read()
    assert logs.count("ref@0") and logs.count("parked in host memory") >= 2 * 3, logs[-1500:]  # 2 workers x 3 steps


@pytest.mark.slow
def test_async_ppo_experiment_staleness_and_weight_publication(tmp_path):
    from areal_vllm_amd.api.cli_args import GenerationHyperparameters, InferenceServerConfig
    from areal_vllm_amd.apps.main import main_start
    from areal_vllm_amd.experiments.async_ppo_math_exp import AsyncPPOMATHExperiment
    from areal_vllm_amd.experiments.ppo_math_exp import PPOHyperparameters
    from areal_vllm_amd.system.agent_api import AgentAbstraction, EnvServiceAbstraction
    from areal_vllm_amd.system.system_api import DatasetAbstraction, ExperimentSaveEvalControl

    root = str(tmp_path)
    rng = random.Random(0)
    rows = [dict(query_id=str(i), input_ids=[rng.randint(2, 400) for _ in range(rng.randint(3, 10))]) for i in range(16)]
    p = os.path.join(root, "data.jsonl")
    with open(p, "w") as f:
        f.write("\n".join(json.dumps(r) for r in rows))
    ppo = PPOHyperparameters(gen=GenerationHyperparameters(max_new_tokens=8), ppo_n_minibatches=1,
                             use_decoupled_loss=True, group_adv_norm=True)
    eta = 1
    e = AsyncPPOMATHExperiment(
        experiment_name="t-ppo", trial_name="async", actor=_actor(), ppo=ppo, layout="d1",
        dataset=DatasetAbstraction("prompt", args=dict(dataset_path=p)), dataset_size=16,
        agent=AgentAbstraction("math-single-step", args=dict(answer_save_path=os.path.join(root, "answers"))),
        env=EnvServiceAbstraction("token-parity"),
        server=InferenceServerConfig(model_path="tiny", context_length=256, max_running_requests=16,
                                     disable_cuda_graph=True),
        n_generation_servers=1, n_rollout_workers=2, train_bs_n_seqs=4, group_size=2, max_head_offpolicyness=eta,
        max_concurrent_rollouts=8, new_tokens_per_chunk=4, device="cpu", fileroot=root,
        exp_ctrl=ExperimentSaveEvalControl(total_train_epochs=100, benchmark_steps=4))
    main_start(e, "t-ppo", "async", fileroot=root, platform="cpu", timeout=500)
    st = _stats(root, "t-ppo", "async")
    assert len(st) == 4
    # bounded staleness. The gate works at ALLOCATION time (expected consumption step of the n-th admitted sample
    # <= version + eta, as in the reference's manager): an answer that finishes behind later-admitted ones is trained
    # on later than expected, by at most the in-flight window (8 rollouts = 2 train batches here). When the answers
    # finish in admission order the bound is eta itself; one step of slack is what reordering costs in practice
    # (the per-batch MEAN below stays inside eta either way).
    heads = [s["actor_train/max_head_offpolicyness"] for s in st]
    assert all(h <= eta + 1 for h in heads) and heads[0] <= eta, st
    assert all(s["actor_train/head_offpolicyness"] <= eta for s in st), st  # the mean stays inside the bound
    assert any(s["actor_train/grad_norm"] > 0 for s in st)  # rewards differ inside groups
    # the trainer published weights after every step and the manager rolled them out; old versions are GC'ed
    pub = os.path.join(root, "param_realloc", "t-ppo", "async", "actor")
    assert "4" in os.listdir(pub) and "1" not in os.listdir(pub)
    log_dir = os.path.join(root, "logs", "root" if os.path.isdir(os.path.join(root, "logs", "root")) else "",
                           "t-ppo", "async")
    with open(os.path.join(log_dir, "gserver_manager.log")) as f:
        txt = f.read()
    assert "updated weights" in txt and "(version 3)" in txt
    # chunked generation: 8 new tokens in chunks of 4 means every answer crossed the router at least twice
    assert os.listdir(os.path.join(root, "answers"))


def test_null_sft_experiment_runs_the_runtime_without_models(tmp_path):
    """`null-sft`: master + 2 model workers + dataset + data-flow graph with the null interface."""
    from areal_vllm_amd.apps.main import main_start
    from areal_vllm_amd.system.system_api import DatasetAbstraction, ExperimentSaveEvalControl, make_experiment

    root = str(tmp_path)
    rows = [dict(id=str(i), prompt_ids=[3 + i, 4, 5], answer_ids=[7, 8]) for i in range(16)]
    p = os.path.join(root, "data.jsonl")
    with open(p, "w") as f:
        f.write("\n".join(json.dumps(r) for r in rows))
    exp = make_experiment("null-sft")
    exp.experiment_name, exp.trial_name, exp.layout, exp.fileroot = "t-null", "t0", "d2", root
    exp.dataset = DatasetAbstraction("prompt_answer", args=dict(dataset_path=p, eos_token_id=1))
    exp.train_bs_n_seqs = 8
    exp.exp_ctrl = ExperimentSaveEvalControl(total_train_epochs=2)
    main_start(exp, "t-null", "t0", fileroot=root, platform="cpu", timeout=300)
    st = _stats(root, "t-null", "t0")
    assert [s["global_step"] for s in st] == [1, 2, 3, 4]
    # every data-parallel head reports its half of the batch (prompt 3 + answer 2 + eos tokens per row)
    assert all(s["trainDefault/n_seqs"] in (4, 8) and s["trainDefault/n_tokens"] == 6 * s["trainDefault/n_seqs"]
               for s in st), st


@pytest.mark.slow
def test_sync_ppo_generation_replica_is_fed_by_parameter_reallocation(tmp_path, monkeypatch):
    """The actor trains tensor-parallel (t2) while a second, data-parallel copy of it (d2) generates on the same
    two workers: after every train step the trainer's shards are re-allocated into the generation copy, worker
    to worker (no checkpoint); AREAL_CHECK_REALLOC makes the workers compare the two models after each hand-over."""
    from areal_vllm_amd.api.cli_args import GenerationHyperparameters, OptimizerConfig
    from areal_vllm_amd.apps.main import main_start
    from areal_vllm_amd.experiments.common import ModelTrainEvalConfig
    from areal_vllm_amd.experiments.ppo_math_exp import PPOHyperparameters, PPOMATHExperiment
    from areal_vllm_amd.system.dfg import ModelName, ParamReallocHook
    from areal_vllm_amd.system.system_api import DatasetAbstraction, ExperimentSaveEvalControl

    monkeypatch.setenv("AREAL_CHECK_REALLOC", "1")
    root = str(tmp_path)
    rng = random.Random(0)
    rows = [dict(query_id=str(i), input_ids=[rng.randint(2, 400) for _ in range(rng.randint(3, 10))], task="math",
                 solutions=["\\boxed{1}"]) for i in range(16)]
    p = os.path.join(root, "data.jsonl")
    with open(p, "w") as f:
        f.write("\n".join(json.dumps(r) for r in rows))
    actor = ModelTrainEvalConfig(path="tiny", init_from_scratch=True, dtype="float32", backend="native_train",
                                 optimizer=OptimizerConfig(lr=1e-3, lr_scheduler_type="constant",
                                                           warmup_steps_proportion=0.0))
    ppo = PPOHyperparameters(gen=GenerationHyperparameters(max_new_tokens=8), ppo_n_minibatches=1)

    def exp():
        return PPOMATHExperiment(experiment_name="t-ppo", trial_name="realloc", actor=actor, ppo=ppo, layout="t2",
                                 gen_layout="d2", dataset=DatasetAbstraction("math_code_prompt", args=dict(dataset_path=p)),
                                 reward_args=dict(dataset_path=p), train_bs_n_seqs=4, group_size=2, device="cpu",
                                 fileroot=root, exp_ctrl=ExperimentSaveEvalControl(total_train_epochs=1, benchmark_steps=2))

    cfg = exp().initial_setup().lazy_init()
    gen = next(r for r in cfg.model_rpcs if r.name == "actor_gen")
    train = next(r for r in cfg.model_rpcs if r.name == "actor_train")
    assert gen.model_name == ModelName("actor", 1)
    assert [(h.source, h.target) for h in train._post_hooks if isinstance(h, ParamReallocHook)] == \
        [(ModelName("actor", 0), ModelName("actor", 1))]
    topos = cfg.model_worker[0].model_topos
    assert topos[ModelName("actor", 0)].tensor_parallel_size == 2 and topos[ModelName("actor", 1)].data_parallel_size == 2
    main_start(exp(), "t-ppo", "realloc", fileroot=root, platform="cpu", timeout=400)
    st = _stats(root, "t-ppo", "realloc")
    assert len(st) == 2 and all(s["actor_train/update_successful"] == 1.0 for s in st)
    logs = ""
    for base, _, files in os.walk(os.path.join(root, "logs")):
        for fn in files:
            if fn.startswith("model_worker"):
                with open(os.path.join(base, fn), errors="ignore") as f:
                    logs += f.read()
    # start-up hand-over + one per train step, on both workers
    assert logs.count("re-allocation actor@0 -> actor@1 verified") >= 2 * 3, logs[-2000:]


@pytest.mark.slow
def test_sync_ppo_pipeline_parallel_actor_generates_through_its_stages(tmp_path):
    """No separate generation copy: the p2 actor generates through its own two stages (tokens ring back from the
    last stage to the first), then trains on what it generated."""
    from areal_vllm_amd.api.cli_args import GenerationHyperparameters, OptimizerConfig
    from areal_vllm_amd.apps.main import main_start
    from areal_vllm_amd.experiments.common import ModelTrainEvalConfig
    from areal_vllm_amd.experiments.ppo_math_exp import PPOHyperparameters, PPOMATHExperiment
    from areal_vllm_amd.system.system_api import DatasetAbstraction, ExperimentSaveEvalControl

    root = str(tmp_path)
    rng = random.Random(0)
    rows = [dict(query_id=str(i), input_ids=[rng.randint(2, 400) for _ in range(rng.randint(3, 10))], task="math",
                 solutions=["\\boxed{1}"]) for i in range(16)]
    p = os.path.join(root, "data.jsonl")
    with open(p, "w") as f:
        f.write("\n".join(json.dumps(r) for r in rows))
    actor = ModelTrainEvalConfig(path="tiny", init_from_scratch=True, dtype="float32", backend="native_train",
                                 optimizer=OptimizerConfig(lr=1e-3, lr_scheduler_type="constant",
                                                           warmup_steps_proportion=0.0))
    ppo = PPOHyperparameters(gen=GenerationHyperparameters(max_new_tokens=8), ppo_n_minibatches=1)
    e = PPOMATHExperiment(experiment_name="t-ppo", trial_name="ppgen", actor=actor, ppo=ppo, layout="p2",
                          dataset=DatasetAbstraction("math_code_prompt", args=dict(dataset_path=p)),
                          reward_args=dict(dataset_path=p), train_bs_n_seqs=4, group_size=2, device="cpu", fileroot=root,
                          exp_ctrl=ExperimentSaveEvalControl(total_train_epochs=1, benchmark_steps=2))
    main_start(e, "t-ppo", "ppgen", fileroot=root, platform="cpu", timeout=400)
    st = _stats(root, "t-ppo", "ppgen")
    assert len(st) == 2 and all(s["actor_train/update_successful"] == 1.0 for s in st)
    assert all("actor_gen/time" in s for s in st)


class _FakeRay:
    """Stand-in for the `ray` module (not installed here): tasks are spawned processes, resources a fixed dict."""

    class exceptions:  # noqa: N801
        class GetTimeoutError(Exception):
            pass

    def __init__(self, resources):
        self.resources, self.launched, self.inited = resources, [], False

    def is_initialized(self):
        return self.inited

    def init(self, runtime_env=None):
        self.inited, self.runtime_env = True, runtime_env

    def shutdown(self):
        self.inited = False
        for _o, p in self.launched:
            if p.is_alive():
                p.terminate()

    def available_resources(self):
        return dict(self.resources)

    def remote(self, **opts):
        import multiprocessing as mp

        ray = self

        class _Fn:
            def __init__(self, fn):
                self.fn = fn

            def remote(self, *args):
                p = mp.get_context("spawn").Process(target=self.fn, args=args, daemon=True)
                p.start()
                ray.launched.append((opts, p))
                return p

        return _Fn

    def get(self, refs, timeout=None):
        import time

        deadline = time.time() + (timeout or 0)
        for p in refs:
            p.join(max(0.0, deadline - time.time()))
            if p.is_alive():
                raise self.exceptions.GetTimeoutError()
            assert p.exitcode == 0, f"ray task exited with {p.exitcode}"


def test_ray_placement_plan_packs_gpu_workers_node_by_node():
    from areal_vllm_amd.system.controller import plan_ray_placement
    from areal_vllm_amd.system.system_api import Scheduling, TasksGroup

    res = {"CPU": 256, "GPU": 16, "memory": 4000 * 1024 ** 3, "node:10.0.0.1": 1.0, "node:10.0.0.2": 1.0}
    counts = [("master_worker", 1, TasksGroup(1, Scheduling(cpu=4, mem=16000))),
              ("model_worker", 12, TasksGroup(12, Scheduling(cpu=8, gpu=1, mem=64000))),
              ("generation_server", 4, TasksGroup(4, Scheduling(cpu=8, gpu=1, mem=64000)))]
    plan = plan_ray_placement(counts, res)
    mw = [o for wt, _i, o in plan if wt == "model_worker"]
    assert [list(o["resources"])[0] for o in mw] == ["node:10.0.0.1"] * 8 + ["node:10.0.0.2"] * 4
    assert all(o["resources"][list(o["resources"])[0]] == 1 / 8 and o["num_gpus"] == 1 for o in mw)
    gs = [o for wt, _i, o in plan if wt == "generation_server"]
    assert [list(o["resources"])[0] for o in gs] == ["node:10.0.0.2"] * 4  # continue after the model workers
    assert "resources" not in [o for wt, _i, o in plan if wt == "master_worker"][0]
    with pytest.raises(RuntimeError, match="required"):
        plan_ray_placement(counts, dict(res, GPU=8))


def test_ray_controller_runs_an_experiment_with_ray_tasks_as_workers(tmp_path, monkeypatch):
    """`RayController` launches the workers as (fake) Ray tasks with the planned options, then configures and
    supervises them like the process-based controller: the null SFT experiment completes."""
    from areal_vllm_amd.system.controller import RayController
    from areal_vllm_amd.system.system_api import DatasetAbstraction, ExperimentSaveEvalControl, make_experiment
    from areal_vllm_amd.utils import name_resolve, names

    root = str(tmp_path)
    rows = [dict(id=str(i), prompt_ids=[3 + i, 4, 5], answer_ids=[7, 8]) for i in range(16)]
    p = os.path.join(root, "data.jsonl")
    with open(p, "w") as f:
        f.write("\n".join(json.dumps(r) for r in rows))
    exp = make_experiment("null-sft")
    exp.experiment_name, exp.trial_name, exp.layout, exp.fileroot, exp.device = "t-ray", "t0", "d2", root, "cpu"
    exp.dataset = DatasetAbstraction("prompt_answer", args=dict(dataset_path=p, eos_token_id=1))
    exp.train_bs_n_seqs = 8
    exp.exp_ctrl = ExperimentSaveEvalControl(total_train_epochs=1)
    monkeypatch.setenv("AREAL_PLATFORM", "cpu")
    # the tasks are fresh processes with the default (file-based) name service; an earlier test of this session
    # may have left the in-memory one selected here
    name_resolve.reconfigure("nfs", nfs_record_root="/tmp/areal/name_resolve")
    name_resolve.clear_subtree(names.trial_root("t-ray", "t0"))
    ray = _FakeRay({"CPU": 64, "GPU": 2, "memory": 512 * 1024 ** 3, "node:127.0.0.1": 1.0})
    RayController("t-ray", "t0", ray_module=ray).start(exp, env=dict(AREAL_PLATFORM="cpu"), timeout=300)
    assert sorted(o["name"] for o, _p in ray.launched) == ["master_worker/0", "model_worker/0", "model_worker/1"]
    assert all(p.exitcode == 0 for _o, p in ray.launched) and not ray.inited
    st = _stats(root, "t-ray", "t0")
    assert [s["global_step"] for s in st] == [1, 2]
