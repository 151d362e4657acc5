"""Unit tests of the API layer: allocation-mode grammar, YAML config system, staleness control."""
import dataclasses
import os
import time

import pytest
import torch

from areal_vllm_amd.api.alloc_mode import AllocationMode, AllocationType
from areal_vllm_amd.api.cli_args import (
    AdvNormConfig,
    GenerationHyperparameters,
    GRPOConfig,
    InferenceEngineConfig,
    InferenceServerConfig,
    NormConfig,
    apply_overrides,
    load_expr_config,
    resolve_interpolations,
    to_structured_cfg,
)


# ------------------------------------------------------------------------- allocation mode
@pytest.mark.parametrize("s,gen_ws,train_ws,tp,pp,dp", [
    ("d2p2t1", 4, 4, 1, 2, 2),
    ("sglang.d4p1t1+d4p1t1", 4, 4, 1, 1, 4),
    ("native.d4p1t2+d2p2t2", 8, 8, 2, 2, 2),
    ("vllm.d2p1t4+d8p1t1", 8, 8, 1, 1, 8),
    ("sglang:d4t2+fsdp:d8", 8, 8, 1, 1, 8),
    ("sglang:d2t4+megatron:d2p2t2", 8, 8, 2, 2, 2),
])
def test_alloc_mode_shapes(s, gen_ws, train_ws, tp, pp, dp):
    a = AllocationMode.from_str(s)
    assert a.gen.world_size == gen_ws and a.train.world_size == train_ws
    assert (a.train.tp_size, a.train.pp_size, a.train.dp_size) == (tp, pp, dp)


def test_alloc_mode_types_and_backends():
    assert AllocationMode.from_str("d4p1t1").type_ == AllocationType.COLOCATE
    a = AllocationMode.from_str("sglang.d4p1t1+d4p1t1")
    assert a.type_ == AllocationType.DECOUPLED_TRAIN and a.gen_backend == "sglang"
    assert a.gen_instance_size == 1
    e = AllocationMode.from_str("sglang.d4p1t2+eval")
    assert e.type_ == AllocationType.DECOUPLED_EVAL and e.gen.world_size == 8 and e.gen_instance_size == 2
    s = AllocationMode.from_str("sglang.d4p1t1")
    assert s.type_ == AllocationType.LLM_SERVER_ONLY
    m = AllocationMode.from_str("sglang:d4t2+megatron:d2p2t2")
    assert m.train_backend == "megatron"
    c = AllocationMode.from_str("sglang:d2t2|fsdp:d4")
    assert c.type_ == AllocationType.COLOCATE


def test_alloc_mode_context_and_expert_parallel():
    a = AllocationMode.from_str("sglang:d4t2+fsdp:d2c2t2")
    assert a.train.cp_size == 2 and a.train.world_size == 8
    m = AllocationMode.from_str("sglang:d2t4+megatron:(attn:d4p1t2c1|ffn:d2p1t1e4)")
    assert m.train.ep_size == 4 and m.train.world_size == 8


@pytest.mark.parametrize("bad", ["", "sglang.d4p1t1+", "foo:d4", "d0p1t1"])
def test_alloc_mode_rejects(bad):
    with pytest.raises(Exception):
        AllocationMode.from_str(bad)


# -------------------------------------------------------------------------------- config
def test_overrides_and_interpolation(tmp_path):
    y = tmp_path / "c.yaml"
    y.write_text("""
experiment_name: e
trial_name: t
allocation_mode: d1p1t1
gconfig: {n_samples: 4, max_new_tokens: 32}
actor:
  experiment_name: ${experiment_name}
  trial_name: ${trial_name}
  path: tiny
  group_size: ${gconfig.n_samples}
  adv_norm: {mean_level: group, std_level: batch, group_size: "${gconfig.n_samples}"}
train_dataset: {path: "synthetic:low-token", batch_size: 8}
""")
    cfg, path = load_expr_config(["--config", str(y), "gconfig.n_samples=8", "actor.optimizer.lr=0.5",
                                  "+actor.eps_clip=0.3", "seed=7"], GRPOConfig)
    assert cfg.actor.group_size == 8 and cfg.actor.adv_norm.group_size == 8
    assert cfg.actor.optimizer.lr == 0.5 and cfg.actor.eps_clip == 0.3 and cfg.seed == 7
    assert cfg.actor.experiment_name == "e" and path == str(y)
    assert isinstance(cfg.actor.adv_norm, AdvNormConfig)


def test_structured_cfg_rejects_unknown_and_bad_types():
    with pytest.raises(Exception):
        to_structured_cfg({"no_such_field": 1}, InferenceEngineConfig)
    with pytest.raises(Exception):
        to_structured_cfg({"consumer_batch_size": "many"}, InferenceEngineConfig)
    c = to_structured_cfg({"max_concurrent_rollouts": "12", "request_timeout": 5}, InferenceEngineConfig)
    assert c.max_concurrent_rollouts == 12 and c.request_timeout == 5.0


def test_norm_config_family():
    n = AdvNormConfig(mean_level="group", std_level="batch", group_size=4)
    assert isinstance(n, NormConfig) and dataclasses.asdict(n)["group_size"] == 4
    assert AdvNormConfig() == AdvNormConfig()


def test_generation_hyperparameters_new():
    g = GenerationHyperparameters(n_samples=4, temperature=1.0)
    h = g.new(n_samples=1, temperature=0.6)
    assert (h.n_samples, h.temperature, g.n_samples) == (1, 0.6, 4)


def test_server_cmd_roundtrip():
    from areal_vllm_amd.inference.server import parse_server_args

    cfg = InferenceServerConfig(model_path="tiny", context_length=512, max_running_requests=16, dtype="float32",
                                num_continuous_decode_steps=4, cuda_graph_bs=[1, 2, 4], disable_radix_cache=True)
    cmd = InferenceServerConfig.build_cmd(cfg, tp_size=1, base_gpu_id=0, host="127.0.0.1", port=1234,
                                          dist_init_addr="127.0.0.1:999")
    assert cmd.startswith("python3 -m areal_vllm_amd.inference.server ")
    args, parsed = parse_server_args(cmd.split()[3:])
    assert args.port == 1234 and args.host == "127.0.0.1"
    for f in ("model_path", "context_length", "max_running_requests", "dtype", "num_continuous_decode_steps",
              "cuda_graph_bs", "disable_radix_cache"):
        assert getattr(parsed, f) == getattr(cfg, f), f


# ----------------------------------------------------------------------- staleness control
class _FakeEngine:
    def __init__(self):
        self.v = 0

    def get_version(self):
        return self.v


def _executor(max_conc=8, batch=2, eta=1):
    from areal_vllm_amd.api.workflow_api import WorkflowExecutor

    eng = _FakeEngine()
    ex = WorkflowExecutor(InferenceEngineConfig(max_concurrent_rollouts=max_conc, consumer_batch_size=batch,
                                                max_head_offpolicyness=eta, queue_size=64), eng)
    ex.initialize(train_data_parallel_size=1)
    return eng, ex


class _EchoWorkflow:
    def __init__(self, delay=0.0):
        self.delay = delay

    async def arun_episode(self, engine, data):
        import asyncio

        await asyncio.sleep(self.delay)
        n = data["n"]
        return dict(input_ids=torch.full((1, n), data["i"]), attention_mask=torch.ones(1, n, dtype=torch.bool),
                    rewards=torch.tensor([float(data["i"])]))


def test_capacity_formula_and_version_gate():
    eng, ex = _executor(max_conc=8, batch=2, eta=1)
    try:
        # (eta + version + 1) * batch = 4 trajectories may exist at version 0
        assert ex.get_capacity() == 4
        for i in range(6):
            ex.submit(dict(i=i, n=3 + i), _EchoWorkflow())
        out = ex.wait(4, timeout=20)
        assert out["input_ids"].shape[0] == 4
        # the two remaining inputs stay queued until the version advances
        with pytest.raises(TimeoutError):
            ex.wait(1, timeout=0.5)
        eng.v = 1
        ex.notify_version_changed()
        out = ex.wait(2, timeout=20)
        assert out["input_ids"].shape[0] == 2
    finally:
        ex.destroy()


def test_wait_rejects_and_pads():
    eng, ex = _executor(max_conc=8, batch=8, eta=4)
    try:
        for i in range(6):
            ex.submit(dict(i=i, n=2 + i), _EchoWorkflow())
        out = ex.wait(3, timeout=20, should_accept=lambda t: float(t["rewards"][0]) % 2 == 0)
        assert sorted(out["rewards"].tolist()) == [0.0, 2.0, 4.0]
        assert out["input_ids"].shape[1] == out["attention_mask"].sum(-1).max()
    finally:
        ex.destroy()


def test_pause_blocks_new_episodes():
    eng, ex = _executor(max_conc=4, batch=4, eta=8)
    try:
        ex.pause()
        ex.submit(dict(i=1, n=2), _EchoWorkflow())
        with pytest.raises(TimeoutError):
            ex.wait(1, timeout=0.5)
        ex.resume()
        assert ex.wait(1, timeout=20)["input_ids"].shape[0] == 1
    finally:
        ex.destroy()


def test_worker_runtime_allocation_mode_spelling():
    """`m` is the older spelling of the tensor-parallel degree; experiments derive their trainer layout and
    generation-server counts from `allocation_mode`."""
    from areal_vllm_amd.api.alloc_mode import AllocationMode, AllocationType
    from areal_vllm_amd.experiments.async_ppo_math_exp import AsyncPPOMATHExperiment
    from areal_vllm_amd.experiments.sft_exp import SFTExperiment

    a = AllocationMode.from_str("sglang.d4p1m2+d2p2m1")
    assert a.type_ == AllocationType.DECOUPLED_TRAIN and a.gen.tp_size == 2 and a.gen.dp_size == 4
    assert a.train.pp_size == 2 and a.train.dp_size == 2 and a.train.tp_size == 1
    with pytest.raises(ValueError):
        AllocationMode.from_str("d2m2t2")  # the same dimension twice
    e = AsyncPPOMATHExperiment(allocation_mode="sglang.d4p1m2+d2p2m1")
    sched = e.scheduling_setup()
    assert (e.layout, e.n_generation_servers, e.gen_tp_size) == ("d2t1p2", 4, 2)
    assert sched.generation_server.count == 4 and sched.model_worker.count == 4
    s = SFTExperiment(allocation_mode="d4p2m1")
    s.scheduling_setup()
    assert s.layout == "d4t1p2"
