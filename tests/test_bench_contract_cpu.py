"""The benchmark entry point's contract, rehearsed on CPU (gloo): launched the way the scaling runs launch it
(one rank per device through torch.distributed.run), it prints exactly ONE line on stdout - the JSON result -
with the whole-job aggregate, on 1 and on 4 ranks (the trainer pads its flat parameter buffer per data-parallel
size, which the colocated weight copy must tolerate); launched as plain ``python bench.py --gpus N`` it starts the
N ranks itself."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--device", "cpu", "--model", "tiny", "--steps", "2", "--warmup", "1", "--prompts-per-gpu", "3", "--n-samples", "2",
        "--max-new-tokens", "8", "--prompt-len-min", "4", "--prompt-len-max", "8"]


def _free_port():
    """A rendezvous port whose neighbours (+1, +2: trainer group and host store of the decoupled mode; +16..18: the
    decoupled second pass) are free as well - chosen the way `bench.py` chooses it for the ranks it spawns itself."""
    sys.path.insert(0, ROOT)
    import bench

    return bench._pick_master_port()


def _err(p):
    """The children's stderr without the C++ frames of torch's error reports (they push the Python traceback out)."""
    return "\n".join(l for l in p.stderr.splitlines() if "frame #" not in l)[-6000:]


@pytest.mark.parametrize("n,mode", [(1, "colocated"), (4, "colocated"), (2, "decoupled"), (8, "decoupled")])
def test_bench_prints_one_json_line(n, mode):
    env = dict(os.environ, AREAL_PLATFORM="cpu", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "AREAL_LLM_SERVER_ADDRS"):
        env.pop(k, None)
    if n == 1:
        cmd = [sys.executable, "bench.py", *ARGS]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", str(n), "--mode", mode, "--no-decoupled-pass", *ARGS]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
    assert p.returncode == 0, _err(p)
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config"):
        assert key in out, key
    assert out["n_gpus"] == n and out["steps"] == 2 and out["warmup"] == 1 and out["higher_is_better"] is True
    n_train = n if mode == "colocated" else n // 2
    assert out["scaling"] == "weak" and out["config"]["global_batch"] == 3 * n_train
    assert out["vs_baseline"] is None  # the reference publishes no number for this metric / configuration
    # whole-job aggregate: samples per step = prompts x samples x trainer ranks
    assert out["value"] == pytest.approx(3 * 2 * n_train / (out["ms_per_step"] / 1000.0), rel=0.02)
    if mode == "decoupled":
        assert out["weight_update"]["transport"] == "ipc" and out["weight_update"]["fallback"] is None
        assert len(out["rollout_servers"]) == n // 2 and all(sv["gen_tokens"] > 0 for sv in out["rollout_servers"])


@pytest.mark.parametrize("n", [2, 4, 8])
def test_plain_invocation_spawns_its_own_ranks(n):
    """``python bench.py --gpus N`` with no torchrun environment starts N rank processes itself, prints ONE line,
    proves the rank count through a collective (``rccl_ranks_seen``) and, N being even, appends the short
    decoupled pass."""
    env = dict(os.environ, AREAL_PLATFORM="cpu", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "AREAL_LLM_SERVER_ADDRS"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "bench.py", "--gpus", str(n), *ARGS], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, _err(p)
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["rccl_ranks_seen"] == n and len(out["devices"]) == n
    assert out["mode"] == "colocated" and out["config"]["global_batch"] == 3 * n
    assert out["value"] == pytest.approx(3 * 2 * n / (out["ms_per_step"] / 1000.0), rel=0.02)
    dec = out["decoupled"]
    assert "error" not in dec, dec
    assert dec["value"] > 0 and dec["gen_tokens_per_s"] > 0 and dec["transport"] == "ipc" and dec["rccl_ranks_seen"] == n
    assert dec["layout"].startswith(f"decoupled: {n // 2} rollout server")


def test_a_failing_rank_fails_the_self_spawned_run():
    env = dict(os.environ, AREAL_PLATFORM="cpu", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "3", "--mode", "decoupled", *ARGS], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and not p.stdout.strip() and "even --gpus" in p.stderr


def test_decoupled_mode_refuses_an_odd_rank_count():
    env = dict(os.environ, AREAL_PLATFORM="cpu", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "bench.py", "--mode", "decoupled", *ARGS], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "even --gpus" in p.stderr


def test_a_stalled_run_exits_with_its_stacks_instead_of_hanging():
    """No milestone within --stall-timeout: every thread's stack goes to stderr and the process exits with
    status 3 (a scaling run with a dead rank then fails in bounded time; here the limit is simply set below the
    start-up time)."""
    env = dict(os.environ, AREAL_PLATFORM="cpu", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "bench.py", "--stall-timeout", "0.01", *ARGS, "--steps", "500"], cwd=ROOT,
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 3, (p.returncode, p.stderr[-2000:])
    assert "no progress for" in p.stderr and "Thread" in p.stderr and not p.stdout.strip()
