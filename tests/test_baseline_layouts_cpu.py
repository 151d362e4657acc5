"""The benchmark allocations of the reference's headline set-ups, launched for real (local launcher, one process
per rank, CPU platform / gloo, tiny random-init model, synthetic verifiable task):

* two tensor-parallel (tp 2) rollout servers + four data-parallel trainer ranks, asynchronous decoupled PPO
  (the 7B math layout; reference ``examples/math/gsm8k_grpo.yaml`` + ``docs/best_practices/handling_oom.md``),
* one tp-4 rollout server + a trainer laid out tensor-parallel 4 x pipeline-parallel 2 (the 32B layout).

Both update weights through peer memory: every server rank pulls the ranges of ITS shard out of the buffers the
trainer ranks export. The full-size YAMLs under ``examples/baseline/`` name the same allocations; they are parsed
here, not run (no 8-GPU node, no checkpoints)."""
import getpass
import os
import tempfile

import pytest
import yaml

from test_e2e_cpu import ROOT, _rewards, _run_launcher

BASE = os.path.join(ROOT, "examples", "baseline")


def _launch(tag, overrides):
    root = tempfile.mkdtemp(prefix=f"areal_base_{tag}_")
    r = _run_launcher("examples/synthetic/toy_grpo.py", "examples/synthetic/toy_grpo.yaml",
                      [f"cluster.fileroot={root}", f"cluster.name_resolve.nfs_record_root={root}/nr", f"trial_name={tag}",
                       "async_training=true", "total_train_epochs=1", "rollout_weight_update=ipc", *overrides],
                      timeout=900, GLOO_SOCKET_IFNAME="lo")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    text = open(os.path.join(root, "logs", getpass.getuser(), "toy-grpo", tag, "trainer.log")).read()
    assert "Training completes" in text
    assert "peer-memory weight update" in text
    return _rewards(os.path.join(root, "logs", getpass.getuser(), "toy-grpo", tag, "trainer.log"))


def test_7b_layout_two_tp2_servers_four_trainer_ranks():
    rewards = _launch("c", ["allocation_mode=native:d2t2+native:d4"])
    assert len(rewards) >= 4 and rewards[-1] > 0.6 > rewards[0], rewards


def test_32b_layout_tp4_server_tp4_pp2_trainer():
    # q/k-norm model: the tensor-parallel server runs the fused decode path with per-head norms
    rewards = _launch("d", ["allocation_mode=native:d1t4+native:d1t4p2", "actor.path=tiny-qknorm",
                            "ref.path=tiny-qknorm", "sglang.model_path=tiny-qknorm", "cluster.n_gpus_per_node=16"])
    assert len(rewards) >= 4 and rewards[-1] > 0.6 > rewards[0], rewards


@pytest.mark.parametrize("name,gen,train,gpus", [
    ("a_cpu_plumbing.yaml", 1, 1, 2),
    ("b_qwen2_1.5b_grpo_4gen_4train_seq4k.yaml", 4, 4, 8),
    ("c_qwen2.5_7b_ppo_gen_tp2_train_dp4_seq24k.yaml", 4, 4, 8),
    ("d_qwen2.5_32b_grpo_gen_tp4_train_tp4pp2.yaml", 4, 8, 12),
    ("e_qwen2_1.5b_search_agent.yaml", 4, 4, 8),
])
def test_baseline_yaml_parses_to_its_allocation(name, gen, train, gpus):
    from areal_vllm_amd.api.alloc_mode import AllocationMode

    cfg = yaml.safe_load(open(os.path.join(BASE, name)))
    mode = AllocationMode.from_str(cfg["allocation_mode"])
    assert mode.gen.world_size == gen and mode.train.world_size == train
    nodes, per = cfg["cluster"]["n_nodes"], cfg["cluster"]["n_gpus_per_node"]
    assert nodes * per >= gpus, (nodes, per)


def test_server_options_this_engine_does_not_implement_are_errors():
    from areal_vllm_amd.api.cli_args import InferenceServerConfig

    InferenceServerConfig(model_path="tiny").reject_unsupported()
    for bad in (dict(kv_cache_dtype="fp8_e5m2"), dict(cpu_offload_gb=4), dict(enable_memory_saver=True),
                dict(torchao_config="int8wo"), dict(dtype="float16")):
        with pytest.raises(ValueError):
            InferenceServerConfig(model_path="tiny", **bad).reject_unsupported()
        with pytest.raises(ValueError):
            InferenceServerConfig.build_cmd(InferenceServerConfig(model_path="tiny", **bad), 1, 0, "127.0.0.1", 1234)
