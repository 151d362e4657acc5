"""The Ray launcher (``launcher/ray.py``) against a stand-in `ray` package (Ray is not installed here): remote
functions run as spawned processes with the task's `runtime_env` variables, placement groups are always ready,
`ray.wait` / `ray.get` / `ray.cancel` follow those processes. Checked: the launcher's own logic - rollout-server
task + trainer ranks with the torch.distributed environment from the placement-group hook, waiting for the
trainer, cancelling the servers - on a decoupled toy GRPO run. Ray's scheduling is not covered."""
import getpass
import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_to_files(cmd, env, outdir, timeout):
    """Run `cmd` with its output in files (not pipes: processes it leaves behind would keep a pipe open and the
    call waiting); returns (returncode, combined text)."""
    os.makedirs(outdir, exist_ok=True)
    path = os.path.join(outdir, "launcher.out")
    with open(path, "w") as f:
        p = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=f, stderr=subprocess.STDOUT, start_new_session=True)
        try:
            rc = p.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, 9)
            rc = -9
    return rc, open(path, errors="ignore").read()


def _leftover_processes(marker):
    """Processes whose command line mentions `marker` (the run's private directory)."""
    import psutil

    out = []
    for pr in psutil.process_iter(["pid", "cmdline"]):
        try:
            if any(marker in a for a in (pr.info["cmdline"] or [])) and pr.pid != os.getpid():
                out.append(pr)
        except (psutil.NoSuchProcess, psutil.AccessDenied):
            pass
    return out


def _assert_nothing_left(marker):
    import time

    deadline = time.time() + 30
    while time.time() < deadline and _leftover_processes(marker):
        time.sleep(0.5)
    left = _leftover_processes(marker)
    for pr in left:
        try:
            pr.kill()
        except Exception:
            pass
    assert not left, [pr.info["cmdline"] for pr in left]

_RAY_INIT = r'''
import multiprocessing as mp
import os, pickle, signal, time
from . import util  # noqa: F401

_procs = []

def init(*a, **k):
    return None

def shutdown():
    for p in _procs:
        if p.is_alive():
            p.terminate()

class ObjectRef:
    def __init__(self, proc=None, value=None):
        self.proc, self.value = proc, value
    def done(self):
        return self.proc is None or not self.proc.is_alive()

def _child(env, payload):
    os.environ.update(env)
    try:
        os.setsid()
    except OSError:
        pass
    fn, args = pickle.loads(payload)
    fn(*args)

class _Remote:
    def __init__(self, fn, opts=None):
        self.fn, self.opts = fn, opts or {}
    def options(self, **opts):
        return _Remote(self.fn, {**self.opts, **opts})
    def remote(self, *args):
        env = dict((self.opts.get("runtime_env") or {}).get("env_vars") or {})
        try:
            payload = pickle.dumps((self.fn, args))
        except Exception:  # a closure (e.g. the launcher's node-address probe): run it here
            return ObjectRef(value=self.fn(*args))
        p = mp.get_context("spawn").Process(target=_child, args=(env, payload))
        p.start()
        _procs.append(p)
        with open(os.environ["FAKE_RAY_LOG"], "a") as f:
            f.write(repr(dict(fn=getattr(self.fn, "__name__", "?"), env={k: env[k] for k in sorted(env) if k in
                    ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR")}, num_gpus=self.opts.get("num_gpus"),
                    bundle=getattr(self.opts.get("scheduling_strategy"), "placement_group_bundle_index", None))) + "\n")
        return ObjectRef(proc=p)

def remote(fn=None, **opts):
    return _Remote(fn, opts) if fn is not None else (lambda f: _Remote(f, opts))

def get(ref, timeout=None):
    if isinstance(ref, (list, tuple)):
        return [get(r, timeout) for r in ref]
    if ref.proc is None:
        return ref.value
    ref.proc.join(timeout)
    if ref.proc.is_alive():
        raise TimeoutError("task still running")
    if ref.proc.exitcode != 0:
        raise RuntimeError(f"task exited with {ref.proc.exitcode}")
    return None

def wait(refs, num_returns=1, timeout=None):
    deadline = time.time() + (timeout or 0)
    while True:
        done = [r for r in refs if r.done()][:num_returns]
        if done or time.time() >= deadline:
            return done, [r for r in refs if r not in done]
        time.sleep(0.2)

def cancel(ref, force=False):
    if ref.proc is not None and ref.proc.is_alive():
        try:
            os.killpg(ref.proc.pid, signal.SIGKILL if force else signal.SIGTERM)
        except OSError:
            ref.proc.terminate()
'''

_RAY_UTIL = r'''
from . import placement_group, scheduling_strategies  # noqa: F401

def get_node_ip_address():
    return "127.0.0.1"
'''

_RAY_PG = r'''
class _PG:
    def __init__(self, bundles, strategy):
        self.bundles, self.strategy = bundles, strategy
    def ready(self):
        import ray
        return ray.ObjectRef(value=True)

created, removed = [], []

def placement_group(bundles, strategy="PACK"):
    pg = _PG(bundles, strategy)
    created.append(pg)
    return pg

def remove_placement_group(pg):
    removed.append(pg)
'''

_RAY_SS = r'''
class PlacementGroupSchedulingStrategy:
    def __init__(self, placement_group, placement_group_bundle_index=-1, placement_group_capture_child_tasks=None):
        self.placement_group = placement_group
        self.placement_group_bundle_index = placement_group_bundle_index
        self.placement_group_capture_child_tasks = placement_group_capture_child_tasks
'''


@pytest.mark.slow
def test_decoupled_experiment_runs_as_ray_tasks():
    d = tempfile.mkdtemp(prefix="areal_fake_ray_")
    pkg = os.path.join(d, "site", "ray")
    os.makedirs(os.path.join(pkg, "util"))
    for rel, body in (("__init__.py", _RAY_INIT), ("util/__init__.py", _RAY_UTIL),
                      ("util/placement_group.py", _RAY_PG), ("util/scheduling_strategies.py", _RAY_SS)):
        with open(os.path.join(pkg, rel), "w") as f:
            f.write(body)
    root = os.path.join(d, "files")
    log = os.path.join(d, "tasks.log")
    env = dict(os.environ, PYTHONPATH=os.path.join(d, "site") + os.pathsep + ROOT, CUDA_VISIBLE_DEVICES="",
               HIP_VISIBLE_DEVICES="", GLOO_SOCKET_IFNAME="lo", AREAL_PLATFORM="cpu", FAKE_RAY_LOG=log)
    cmd = [sys.executable, "-m", "areal_vllm_amd.launcher.ray", os.path.join(ROOT, "examples/synthetic/toy_grpo.py"),
           "--config", os.path.join(ROOT, "examples/synthetic/toy_grpo.yaml"), f"cluster.fileroot={root}",
           f"cluster.name_resolve.nfs_record_root={root}/nr", "trial_name=ray", "allocation_mode=native:d1+native:d2",
           "async_training=true", "total_train_epochs=1", "rollout_weight_update=disk", "cluster.n_gpus_per_node=2"]
    rc, text = _run_to_files(cmd, env, d, timeout=300)
    assert rc == 0, text[-3000:]
    _assert_nothing_left(root)  # the cancelled server task took its rollout server with it
    tasks = [eval(line) for line in open(log)]
    assert [t["fn"] for t in tasks] == ["run_func"] * 3
    server, ranks = tasks[0], tasks[1:]
    assert server["num_gpus"] == 1 and server["bundle"] == 0 and "RANK" not in server["env"]
    assert [t["env"]["RANK"] for t in ranks] == ["0", "1"]
    assert all(t["env"]["WORLD_SIZE"] == "2" and t["env"]["MASTER_ADDR"] == "127.0.0.1" and t["num_gpus"] == 1
               for t in ranks)
    assert [t["env"]["LOCAL_RANK"] for t in ranks] == ["0", "1"]
    logs = os.path.join(root, "logs", getpass.getuser(), "toy-grpo", "ray")
    for base, _, files in os.walk(logs) if os.path.isdir(logs) else []:
        for fn in files:
            text += open(os.path.join(base, fn), errors="ignore").read()
    assert "Training completes" in text, text[-3000:]
