"""End-to-end runs on the CPU platform through the real launcher / trainer / rollout stack.

Modelled on the reference's ``areal/tests/grpo/test_grpo.py`` (local launcher on a tiny config, final
reward threshold) and ``areal/tests/sft/test_sft.py`` (loss curve); data are synthetic because the box
has no network."""
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_launcher(entry, cfg, overrides, timeout=900, **extra_env):
    env = dict(os.environ, PYTHONPATH=ROOT, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", **extra_env)
    cmd = [sys.executable, "-m", "areal_vllm_amd.launcher.local", os.path.join(ROOT, entry), "--config",
           os.path.join(ROOT, cfg), *overrides]
    return subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)


def _rewards(log_path):
    vals = []
    for line in open(log_path):
        m = re.search(r"grpo_actor/task_reward/avg\s*\|\s*([-\d.e]+)", line)
        if m:
            vals.append(float(m.group(1)))
    return vals


def test_grpo_learns_through_local_launcher():
    root = tempfile.mkdtemp(prefix="areal_e2e_")
    r = _run_launcher("examples/synthetic/toy_grpo.py", "examples/synthetic/toy_grpo.yaml",
                      [f"cluster.fileroot={root}", f"cluster.name_resolve.nfs_record_root={root}/nr",
                       "trial_name=e2e"])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    log = os.path.join(root, "logs", os.environ.get("USER", "root"), "toy-grpo", "e2e", "trainer.log")
    if not os.path.exists(log):
        import getpass

        log = os.path.join(root, "logs", getpass.getuser(), "toy-grpo", "e2e", "trainer.log")
    rewards = _rewards(log)
    assert len(rewards) >= 16
    assert rewards[0] < 0.3, rewards
    assert rewards[-1] > 0.6, rewards


@pytest.mark.parametrize("transport", ["xgmi", "disk"])
def test_decoupled_allocation_through_local_launcher(transport):
    """The reference's default deployment: the launcher starts a rollout server process and a two-rank
    trainer (torchrun); rollouts travel over HTTP, weights over the collective push (every trainer rank a
    sender, its own group with the server) or over checkpoint files; asynchronous with staleness 2."""
    root = tempfile.mkdtemp(prefix="areal_e2e_dec_")
    r = _run_launcher("examples/synthetic/toy_grpo.py", "examples/synthetic/toy_grpo.yaml",
                      [f"cluster.fileroot={root}", f"cluster.name_resolve.nfs_record_root={root}/nr", "trial_name=dec",
                       "allocation_mode=native:d1+native:d2", "async_training=true", "total_train_epochs=1",
                       f"rollout_weight_update={transport}"], timeout=600, GLOO_SOCKET_IFNAME="lo")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    import getpass

    log = os.path.join(root, "logs", getpass.getuser(), "toy-grpo", "dec", "trainer.log")
    rewards = _rewards(log)
    assert len(rewards) >= 8, rewards
    assert rewards[-1] > 0.6 > rewards[0], rewards  # the servers really generate with the updated weights
    text = open(log).read()
    assert "Training completes" in text


@pytest.mark.parametrize("layout", ["d2p1t1", "d1p1t2"])
def test_colocated_allocation_with_parallel_trainers(layout):
    """Colocated rollout engines under a data-parallel (one engine per rank, flat-buffer copy) and a
    tensor-parallel trainer (engine on the replica's head, shards gathered and loaded by name)."""
    root = tempfile.mkdtemp(prefix="areal_e2e_col_")
    r = _run_launcher("examples/synthetic/toy_grpo.py", "examples/synthetic/toy_grpo.yaml",
                      [f"cluster.fileroot={root}", f"cluster.name_resolve.nfs_record_root={root}/nr", "trial_name=col",
                       f"allocation_mode={layout}", "async_training=true", "total_train_epochs=1"], timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    import getpass

    rewards = _rewards(os.path.join(root, "logs", getpass.getuser(), "toy-grpo", "col", "trainer.log"))
    assert len(rewards) >= 8 and rewards[-1] > 0.6 > rewards[0], rewards


@pytest.mark.parametrize("layout", [None, "d1p1t2"])
def test_grpo_recover_run_continues(tmp_path, layout):
    """Kill-free recover check: run 1 step budget with recover checkpoints, then relaunch with
    AREAL_RECOVER_RUN semantics (`recover.mode=resume`) and see the step counter continue. The colocated
    tensor-parallel layout re-uploads the recovered weights with a gather every trainer rank takes part in."""
    root = str(tmp_path)
    common = [f"cluster.fileroot={root}", f"cluster.name_resolve.nfs_record_root={root}/nr", "trial_name=rec",
              "recover.freq_steps=1", "recover.freq_epochs=null", "recover.freq_secs=null",
              "train_dataset.path=synthetic:low-token:n=32:len=8"]
    if layout is not None:
        common += [f"allocation_mode={layout}", "async_training=true"]
    r = _run_launcher("examples/synthetic/toy_grpo.py", "examples/synthetic/toy_grpo.yaml",
                      common + ["recover.mode=auto", "total_train_steps=2"])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    r = _run_launcher("examples/synthetic/toy_grpo.py", "examples/synthetic/toy_grpo.yaml",
                      common + ["recover.mode=auto", "total_train_steps=4"])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    import getpass

    log = open(os.path.join(root, "logs", getpass.getuser(), "toy-grpo", "rec", "trainer.log")).read()
    steps = [int(m) for m in re.findall(r"Train step (\d+)/", log)]
    # first run logs steps 1,2; the recovered run continues with 3,4 instead of starting over
    assert steps == [1, 2, 3, 4], steps


def test_sft_loss_decreases():
    sys.path.insert(0, ROOT)
    root = tempfile.mkdtemp(prefix="areal_sft_")
    code = (
        "import sys, json; sys.path.insert(0, %r)\n"
        "from areal_vllm_amd.trainer.sft import run_sft\n"
        "l = run_sft(['--config', %r, 'cluster.fileroot=%s', 'cluster.name_resolve.nfs_record_root=%s/nr'])\n"
        "print('LOSSES', json.dumps(l))\n" % (ROOT, os.path.join(ROOT, "examples/synthetic/arith_sft.yaml"), root, root))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29631",
               CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    import json

    losses = json.loads(re.search(r"LOSSES (.*)", r.stdout).group(1))
    assert len(losses) == 16
    assert losses[-1] < 0.7 * losses[0], losses
