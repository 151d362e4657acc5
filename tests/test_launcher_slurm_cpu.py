"""The Slurm launcher (``launcher/slurm.py``) against a stand-in cluster: `sbatch`, `srun`, `squeue`, `sacct`,
`scancel` and `scontrol` are small scripts on PATH that run the submitted batch scripts as local process groups
and keep their states in a directory. What is checked is the launcher's side of the contract: the batch scripts
it writes run as they stand, job states are followed through the queue and the accounting query, a failing step
fails its job, cancelling reaches the whole job, and a decoupled experiment (rollout-server job + trainer job)
runs to completion through `slurm_main`. No real Slurm is available here: scheduling itself is not covered."""
import getpass
import os
import stat
import subprocess
import sys
import tempfile
import time

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

_COMMON = r'''
import os, sys, signal, subprocess, json
D = os.environ["FAKE_SLURM_DIR"]
os.makedirs(D, exist_ok=True)
def alive(pid):
    try:
        os.kill(pid, 0)
        return True
    except OSError:
        return False
def state(jid):
    if os.path.exists(f"{D}/{jid}.cancelled"):
        return "CANCELLED"
    if os.path.exists(f"{D}/{jid}.rc"):
        return "COMPLETED" if open(f"{D}/{jid}.rc").read().strip() == "0" else "FAILED"
    if os.path.exists(f"{D}/{jid}.pid") and alive(int(open(f"{D}/{jid}.pid").read())):
        return "RUNNING"
    return None
'''

_TOOLS = {
    "sbatch": _COMMON + r'''
script = sys.argv[1]
n = int(open(f"{D}/next").read()) + 1 if os.path.exists(f"{D}/next") else 101
open(f"{D}/next", "w").write(str(n))
out = "/dev/null"
for line in open(script):
    if line.startswith("#SBATCH --output="):
        out = line.split("=", 1)[1].strip()
env = dict(os.environ, SLURM_JOB_ID=str(n), SLURM_JOB_NODELIST="localhost", SLURM_NODEID="0")
p = subprocess.Popen(["bash", "-c", f"bash {script} >> {out} 2>&1; echo $? > {D}/{n}.rc"], env=env,
                     start_new_session=True, stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL,
                     stderr=subprocess.DEVNULL)
open(f"{D}/{n}.pid", "w").write(str(p.pid))
print(f"Submitted batch job {n}")
''',
    "srun": r'''
import os, sys
args = sys.argv[1:]
env = dict(os.environ)
while args and args[0].startswith("-"):
    a = args.pop(0)
    if a in ("-w", "--chdir"):
        args.pop(0)
    elif a.startswith("--export="):
        for kv in a[len("--export="):].split(","):
            if "=" in kv:
                k, v = kv.split("=", 1)
                env[k] = v
if args[0] == "hostname":
    print("127.0.0.1")
    sys.exit(0)
os.execvpe(args[0], args, env)
''',
    "scontrol": r'''
print("localhost")
''',
    "squeue": _COMMON + r'''
ids = sys.argv[sys.argv.index("-j") + 1].split(",")
for j in ids:
    if state(j) == "RUNNING":
        print(f"{j:<20}{'RUNNING':<20}localhost")
''',
    "sacct": _COMMON + r'''
j = sys.argv[sys.argv.index("-j") + 1]
s = state(j)
if s:
    print(s)
''',
    "scancel": _COMMON + r'''
sig = getattr(signal, "SIG" + sys.argv[sys.argv.index("-s") + 1])
j = sys.argv[-1]
open(f"{D}/{j}.cancelled", "w").write("1")
try:
    os.killpg(int(open(f"{D}/{j}.pid").read()), sig)
except (OSError, FileNotFoundError):
    pass
''',
}


@pytest.fixture()
def fake_slurm(monkeypatch):
    d = tempfile.mkdtemp(prefix="areal_fake_slurm_")
    bindir = os.path.join(d, "bin")
    os.makedirs(bindir)
    for name, body in _TOOLS.items():
        path = os.path.join(bindir, name)
        with open(path, "w") as f:
            f.write(f"#!{sys.executable}\n" + body)
        os.chmod(path, os.stat(path).st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", bindir + os.pathsep + os.environ["PATH"])
    monkeypatch.setenv("FAKE_SLURM_DIR", os.path.join(d, "state"))
    return d


def test_jobs_are_followed_from_submission_to_their_final_state(fake_slurm):
    from areal_vllm_amd.launcher.slurm import SlurmLauncher
    from areal_vllm_amd.utils.launcher import JobException, JobState

    root = os.path.join(fake_slurm, "files")
    la = SlurmLauncher("exp", "t0", root)
    marker = os.path.join(fake_slurm, "ok.txt")
    la.submit("good", f"sleep 1; echo $GREETING $AREAL_HEAD_NODE_IP > {marker}", nodes=1, gpus_per_node=0,
              cpus_per_task=1, mem_per_node=100, env=dict(GREETING="hello"))
    assert la.find("good").state in (JobState.RUNNING, JobState.PENDING)
    la.wait(timeout=30, poll_interval=0.3)
    assert open(marker).read().split() == ["hello", "127.0.0.1"]  # the exported variable and the head node's address
    assert la.find("good").state == JobState.COMPLETED
    script = open(la.sbatch_path_of("good")).read()
    assert "#SBATCH --job-name=exp_t0:good" in script and "--export=" in script

    lb = SlurmLauncher("exp", "t1", root)
    lb.submit("bad", "exit 3", nodes=1, gpus_per_node=0, cpus_per_task=1, mem_per_node=100)
    with pytest.raises(JobException):
        lb.wait(timeout=30, poll_interval=0.3)  # a failing step fails the job
    assert lb.find("bad").state == JobState.FAILED

    lc = SlurmLauncher("exp", "t2", root)
    jid = lc.submit("long", "sleep 300", nodes=1, gpus_per_node=0, cpus_per_task=1, mem_per_node=100)
    time.sleep(0.5)
    assert lc.find("long").state == JobState.RUNNING
    pid = int(open(os.path.join(fake_slurm, "state", f"{jid}.pid")).read())
    lc.stop("long", "SIGKILL")
    deadline = time.time() + 10
    while time.time() < deadline:
        try:
            os.killpg(pid, 0)
            time.sleep(0.1)
        except OSError:
            break
    else:
        raise AssertionError("the cancelled job's processes are still running")
    assert lc.find("long").state == JobState.NOT_FOUND  # forgotten by the launcher once stopped


@pytest.mark.slow
def test_decoupled_experiment_runs_as_two_slurm_jobs(fake_slurm):
    """`python -m areal_vllm_amd.launcher.slurm` on the toy GRPO recipe: a rollout-server job and a trainer job
    (torchrun started by the batch script on the 'head node'), until the trainer completes."""
    root = os.path.join(fake_slurm, "files")
    env = dict(os.environ, PYTHONPATH=ROOT, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", GLOO_SOCKET_IFNAME="lo",
               AREAL_PLATFORM="cpu")
    cmd = [sys.executable, "-m", "areal_vllm_amd.launcher.slurm", os.path.join(ROOT, "examples/synthetic/toy_grpo.py"),
           "--config", os.path.join(ROOT, "examples/synthetic/toy_grpo.yaml"), f"cluster.fileroot={root}",
           f"cluster.name_resolve.nfs_record_root={root}/nr", "trial_name=slurm", "allocation_mode=native:d1+native:d1",
           "async_training=true", "total_train_epochs=1", "rollout_weight_update=disk", "cluster.n_gpus_per_node=1"]
    rc, out = _run_to_files(cmd, env, fake_slurm, timeout=300)
    logs = os.path.join(root, "logs", getpass.getuser(), "toy-grpo", "slurm")
    tail = ""
    for fn in ("trainer.log", "llm_server.log"):
        if os.path.exists(os.path.join(logs, fn)):
            tail += f"\n--- {fn}\n" + open(os.path.join(logs, fn), errors="ignore").read()[-2000:]
    assert rc == 0, out[-3000:] + tail
    assert "Training completes" in open(os.path.join(logs, "trainer.log"), errors="ignore").read()
    state = os.path.join(fake_slurm, "state")
    assert sorted(f for f in os.listdir(state) if f.endswith(".pid")) == ["101.pid", "102.pid"]
    assert open(os.path.join(state, "102.rc")).read().strip() == "0"  # the trainer job
    assert os.path.exists(os.path.join(state, "101.cancelled"))  # servers are cancelled once the trainer is done
    _assert_nothing_left(root)  # ... and the cancelled job took its rollout server (a session of its own) with it
