"""Slurm launcher: generated sbatch scripts for rollout-server wrappers and torchrun trainers.

``python -m areal_vllm_amd.launcher.slurm <entry.py> --config <yaml> [k=v ...]`` submits one job array
for the rollout servers (one ``server_wrapper`` task per node) and one for the trainer (one
``torch.distributed.run`` task per node, rendezvous on the first trainer node), polls ``squeue`` /
``sacct`` for their states, and applies the same recover policy as the local launcher. Structure and
job contract follow ``areal/launcher/slurm.py:91-572`` (sbatch template, ``srun`` with optional
apptainer container, job-state polling, cancel with a signal); the script text and state machine are
independent.
"""
import getpass
import os
import re
import shlex
import subprocess
import sys
import time
from typing import Dict, List, Optional, Tuple

from ..api.alloc_mode import AllocationMode, AllocationType
from ..api.cli_args import BaseExperimentConfig
from ..utils import logging, name_resolve, names
from ..utils.launcher import (
    JobException,
    JobInfo,
    JobState,
    get_env_vars,
    validate_config_for_distributed_launcher,
    wait_llm_server_addrs,
)
from ..utils.recover import check_if_recover

logger = logging.getLogger("slurm-launcher")

SLURM_STATES = {
    "PENDING": JobState.PENDING, "CONFIGURING": JobState.PENDING, "RUNNING": JobState.RUNNING,
    "COMPLETING": JobState.RUNNING, "COMPLETED": JobState.COMPLETED, "FAILED": JobState.FAILED,
    "CANCELLED": JobState.CANCELLED, "TIMEOUT": JobState.FAILED, "NODE_FAIL": JobState.FAILED,
    "OUT_OF_MEMORY": JobState.FAILED, "PREEMPTED": JobState.FAILED, "BOOT_FAIL": JobState.FAILED,
}

SBATCH_TEMPLATE = """#!/bin/bash
#SBATCH --job-name={job_name}
#SBATCH --output={log_path}
#SBATCH --open-mode=append
#SBATCH --no-requeue
#SBATCH --nodes={nodes}
#SBATCH --ntasks-per-node={ntasks_per_node}
#SBATCH --gres=gpu:{gpus_per_node}
#SBATCH --cpus-per-task={cpus_per_task}
#SBATCH --mem={mem_per_node}M
{extra_directives}
nodes=( $(scontrol show hostnames $SLURM_JOB_NODELIST) )
head_node=${{nodes[0]}}
head_node_ip=$(srun --overlap --nodes=1 --ntasks=1 -w "$head_node" hostname --ip-address | awk '{{print $1}}')
export AREAL_HEAD_NODE_IP=$head_node_ip
pids=()
{srun_lines}
# the job fails when any of its steps fails (a bare `wait` would report success)
rc=0
for p in "${{pids[@]}}"; do wait "$p" || rc=$?; done
exit $rc
"""


def build_srun_line(cmd: str, env: Dict[str, str], *, nodes: int, ntasks: int, container_type: str = "none",
                    image: Optional[str] = None, mount: Optional[str] = None, srun_args: str = "") -> str:
    exports = ",".join(["ALL"] + [f"{k}={v}" for k, v in env.items()])
    if container_type == "apptainer" and image:
        binds = f"--bind {mount}" if mount else ""
        cmd = f"apptainer exec --rocm --no-home --writable-tmpfs {binds} {image} {cmd}"
    return (f"srun {srun_args} --nodes={nodes} --ntasks={ntasks} --export={shlex.quote(exports)} "
            f"bash -c {shlex.quote(cmd)} &\npids+=($!)")


class SlurmLauncher:
    def __init__(self, experiment_name: str, trial_name: str, fileroot: str, container_type: str = "none",
                 mount: Optional[str] = None, srun_args: str = ""):
        self.experiment_name, self.trial_name, self.fileroot = experiment_name, trial_name, fileroot
        self.container_type, self.mount, self.srun_args = container_type, mount, srun_args
        self.jobs: Dict[int, JobInfo] = {}
        self.job_names: Dict[str, int] = {}

    @property
    def run_name(self):
        return f"{self.experiment_name}_{self.trial_name}"

    def slurm_name(self, job_name: str) -> str:
        return f"{self.experiment_name}_{self.trial_name}:{job_name}"

    def log_path_of(self, job_name: str) -> str:
        d = f"{self.fileroot}/logs/{getpass.getuser()}/{self.experiment_name}/{self.trial_name}"
        os.makedirs(d, exist_ok=True)
        return os.path.join(d, f"{job_name}.log")

    def sbatch_path_of(self, job_name: str) -> str:
        d = f"{self.fileroot}/logs/{getpass.getuser()}/{self.experiment_name}/{self.trial_name}/slurm"
        os.makedirs(d, exist_ok=True)
        return os.path.join(d, f"{job_name}.sh")

    def render(self, job_name: str, cmd: str, *, nodes: int, gpus_per_node: int, cpus_per_task: int,
               mem_per_node: int, env: Optional[Dict[str, str]] = None, image: Optional[str] = None,
               nodelist: Optional[str] = None, exclude: Optional[str] = None) -> str:
        extra = []
        if nodelist:
            extra.append(f"#SBATCH --nodelist={nodelist}")
        if exclude:
            extra.append(f"#SBATCH --exclude={exclude}")
        line = build_srun_line(cmd, env or {}, nodes=nodes, ntasks=nodes, container_type=self.container_type,
                               image=image, mount=self.mount, srun_args=self.srun_args)
        return SBATCH_TEMPLATE.format(job_name=self.slurm_name(job_name), log_path=self.log_path_of(job_name),
                                      nodes=nodes, ntasks_per_node=1, gpus_per_node=gpus_per_node,
                                      cpus_per_task=cpus_per_task, mem_per_node=mem_per_node,
                                      extra_directives="\n".join(extra), srun_lines=line)

    def submit(self, job_name: str, cmd: str, **kw) -> int:
        script = self.render(job_name, cmd, **kw)
        path = self.sbatch_path_of(job_name)
        with open(path, "w") as f:
            f.write(script)
        out = subprocess.check_output(["sbatch", path]).decode()
        m = re.search(r"Submitted batch job (\d+)", out)
        if not m:
            raise RuntimeError(f"sbatch failed: {out}")
        jid = int(m.group(1))
        self.jobs[jid] = JobInfo(name=job_name, state=JobState.PENDING, slurm_id=jid)
        self.job_names[job_name] = jid
        logger.info(f"Submitted {job_name} as slurm job {jid}")
        return jid

    @staticmethod
    def parse_squeue(text: str) -> Dict[int, Tuple[JobState, str]]:
        """`squeue -O jobid,state,nodelist -h` output -> {jobid: (state, nodelist)}."""
        out = {}
        for line in text.strip().splitlines():
            parts = line.split()
            if len(parts) >= 2 and parts[0].split("_")[0].isdigit():
                out[int(parts[0].split("_")[0])] = (SLURM_STATES.get(parts[1], JobState.NOT_FOUND),
                                                    parts[2] if len(parts) > 2 else "")
        return out

    def update(self):
        if not self.jobs:
            return
        ids = ",".join(map(str, self.jobs))
        try:
            text = subprocess.check_output(["squeue", "-j", ids, "-h", "-O", "jobid:20,state:20,nodelist:200"],
                                           stderr=subprocess.DEVNULL).decode()
            seen = self.parse_squeue(text)
        except subprocess.CalledProcessError:
            seen = {}
        for jid, info in self.jobs.items():
            if jid in seen:
                info.state, info.host = seen[jid]
                continue
            # left the queue: ask the accounting database for the final state
            try:
                text = subprocess.check_output(["sacct", "-j", str(jid), "-n", "-X", "-o", "state"],
                                               stderr=subprocess.DEVNULL).decode().split()
                info.state = SLURM_STATES.get(text[0].rstrip("+"), JobState.NOT_FOUND) if text else JobState.NOT_FOUND
            except (subprocess.CalledProcessError, FileNotFoundError):
                info.state = JobState.NOT_FOUND

    def find(self, job_name: str) -> JobInfo:
        self.update()
        jid = self.job_names.get(job_name)
        return self.jobs[jid] if jid is not None else JobInfo(name=job_name, state=JobState.NOT_FOUND)

    def find_all(self, job_name_regex: str = ".*") -> List[JobInfo]:
        self.update()
        return [i for i in self.jobs.values() if re.fullmatch(job_name_regex, i.name)]

    def stop(self, job_name: str, signal: str = "SIGKILL"):
        jid = self.job_names.pop(job_name, None)
        if jid is not None:
            subprocess.call(["scancel", "-s", signal.replace("SIG", ""), "--full", str(jid)])
            self.jobs.pop(jid, None)

    def stop_all(self, signal: str = "SIGKILL"):
        for n in list(self.job_names):
            self.stop(n, signal)

    def wait(self, timeout=None, check_status=(JobState.CANCELLED, JobState.FAILED, JobState.NOT_FOUND),
             remove_status=(JobState.COMPLETED,), update=False, poll_interval: float = 5.0):
        deadline = None if timeout is None else time.time() + timeout
        left = set(self.job_names)
        while left:
            if deadline is not None and time.time() > deadline:
                raise TimeoutError(f"Timeout waiting for {self.run_name}: {', '.join(sorted(left))}")
            self.update()
            for name in list(left):
                info = self.jobs[self.job_names[name]]
                if info.state in check_status:
                    raise JobException(self.run_name, name, info.host, info.state)
                if info.state in remove_status:
                    left.discard(name)
                    if update:
                        self.jobs.pop(self.job_names.pop(name))
            time.sleep(poll_interval)


def slurm_main(config: BaseExperimentConfig, run_id: int = 0, entry_argv: Optional[List[str]] = None):
    entry_argv = entry_argv if entry_argv is not None else sys.argv[1:]
    is_recover_run = check_if_recover(config.recover, run_id)
    am = validate_config_for_distributed_launcher(config)
    sl = config.launcher.slurm
    launcher = SlurmLauncher(config.experiment_name, config.trial_name, config.cluster.fileroot, sl.container_type,
                             sl.mount, sl.srun_additional_args)
    name_resolve.reconfigure(config.cluster.name_resolve)
    name_resolve.clear_subtree(names.trial_root(config.experiment_name, config.trial_name))
    per_node = config.cluster.n_gpus_per_node
    lc = config.launcher
    cfg_args = " ".join(shlex.quote(a) for a in entry_argv[1:])
    try:
        n_servers = 0
        if am.gen is not None and am.gen.world_size > 0 and am.type_ != AllocationType.COLOCATE:
            n_nodes = max(1, am.gen.world_size // per_node)
            n_servers = am.gen.dp_size
            launcher.submit("llm_server", f"python3 -m areal_vllm_amd.launcher.server_wrapper {cfg_args}",
                            nodes=n_nodes, gpus_per_node=min(per_node, am.gen.world_size),
                            cpus_per_task=lc.inference_server_cpus_per_gpu * min(per_node, am.gen.world_size),
                            mem_per_node=lc.inference_server_mem_per_gpu * min(per_node, am.gen.world_size),
                            env=get_env_vars(config.cluster.cluster_name, lc.inference_server_env_vars),
                            image=sl.inference_server_image)
        if am.type_ != AllocationType.LLM_SERVER_ONLY:
            def servers_alive():
                info = launcher.find("llm_server")
                if info.state in (JobState.FAILED, JobState.CANCELLED, JobState.NOT_FOUND, JobState.COMPLETED):
                    raise JobException(launcher.run_name, "llm_server", info.host, JobState.FAILED)

            addrs = wait_llm_server_addrs(config.experiment_name, config.trial_name, n_servers,
                                          check=servers_alive) if n_servers else []
            world = 1 if am.type_ == AllocationType.DECOUPLED_EVAL else am.train.world_size
            n_nodes = max(1, world // per_node)
            gpn = min(per_node, world) if am.type_ != AllocationType.DECOUPLED_EVAL else 0
            cmd = (f"python3 -m torch.distributed.run --nnodes {n_nodes} --nproc-per-node {max(1, gpn)} "
                   f"--node-rank $SLURM_NODEID --master-addr $AREAL_HEAD_NODE_IP --master-port 29577 "
                   f"{' '.join(shlex.quote(a) for a in entry_argv)}")
            env = dict(**get_env_vars(config.cluster.cluster_name, lc.trainer_env_vars),
                       AREAL_LLM_SERVER_ADDRS=",".join(addrs), AREAL_RECOVER_RUN=str(int(is_recover_run)))
            launcher.submit("trainer", cmd, nodes=n_nodes, gpus_per_node=gpn,
                            cpus_per_task=lc.trainer_cpus_per_gpu * max(1, gpn),
                            mem_per_node=lc.trainer_mem_per_gpu * max(1, gpn), env=env, image=sl.trainer_image)
        while True:
            infos = {i.name: i for i in launcher.find_all()}
            for i in infos.values():
                if i.state in (JobState.FAILED, JobState.CANCELLED, JobState.NOT_FOUND) or (
                        i.name != "trainer" and i.state == JobState.COMPLETED):
                    raise JobException(launcher.run_name, i.name, i.host, JobState.FAILED)
            if "trainer" in infos and infos["trainer"].state == JobState.COMPLETED:
                break
            time.sleep(10)
        launcher.stop_all("SIGTERM")
    except (KeyboardInterrupt, JobException, TimeoutError) as e:
        launcher.stop_all("SIGKILL")
        if isinstance(e, JobException) and run_id < config.recover.retries and config.recover.mode in ("auto", "fault"):
            time.sleep(10)
            return slurm_main(config, run_id + 1, entry_argv)
        raise


def main():
    from .local import load_launcher_config

    config, _raw, _ = load_launcher_config(sys.argv[2:])
    slurm_main(config, 0)


if __name__ == "__main__":
    main()
