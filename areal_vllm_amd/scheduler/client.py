"""Scheduler clients of the worker runtime: start the controller and the worker processes of an
experiment, on this machine (sub-processes with round-robin device assignment) or through slurm
(generated sbatch scripts), and watch them. Interface of ``realhf/scheduler/client.py:14-160``; the two
implementations delegate to the launchers of this package (``launcher/local.py``, ``launcher/slurm.py``),
which hold the process / job bookkeeping.
"""
import sys
from typing import Dict, List, Optional

from ..utils.launcher import JobException, JobInfo, JobState  # noqa: F401


class SchedulerClient:
    def __init__(self, expr_name: str, trial_name: str):
        self.expr_name, self.trial_name = expr_name, trial_name
        self.run_name = f"{expr_name}_{trial_name}"

    def submit(self, worker_type: str, cmd: str, **kwargs):
        raise NotImplementedError()

    def submit_array(self, worker_type: str, cmd: str, count: int, **kwargs):
        raise NotImplementedError()

    def stop(self, job_name: str):
        raise NotImplementedError()

    def stop_all(self, signal=None):
        raise NotImplementedError()

    def find(self, job_name: str) -> Optional[JobInfo]:
        raise NotImplementedError()

    def find_all(self, job_name_regex: str = ".*") -> List[JobInfo]:
        raise NotImplementedError()

    def wait(self, timeout=None, **kwargs):
        raise NotImplementedError()


def remote_worker_cmd(expr_name: str, trial_name: str, worker_type: str, index_expr: str = "{index}") -> str:
    return (f"{sys.executable} -m areal_vllm_amd.apps.remote worker -w {worker_type} -e {expr_name} "
            f"-f {trial_name} -i {index_expr}")


def control_cmd(expr_name: str, trial_name: str, experiment_pickle: str, ignore_worker_error: bool = False) -> str:
    return (f"{sys.executable} -m areal_vllm_amd.apps.remote controller -e {expr_name} -f {trial_name} "
            f"--experiment {experiment_pickle}" + (" --ignore_worker_error" if ignore_worker_error else ""))


class LocalSchedulerClient(SchedulerClient):
    def __init__(self, expr_name: str, trial_name: str, fileroot: str = "/tmp/areal"):
        super().__init__(expr_name, trial_name)
        from ..launcher.local import LocalLauncher

        self.launcher = LocalLauncher(expr_name, trial_name, fileroot)

    def submit(self, worker_type: str, cmd: str, gpu: int = 0, env_vars: Optional[Dict[str, str]] = None, **_):
        self.launcher.submit(worker_type, cmd, gpu=gpu, env_vars=env_vars)

    def submit_array(self, worker_type: str, cmd: str, count: int, gpu: int = 0,
                     env_vars: Optional[Dict[str, str]] = None, **_):
        """`cmd` may contain ``{index}``: the position of the process in the array."""
        cmds = [cmd.format(index=i) if "{index}" in cmd else cmd for i in range(count)]
        self.launcher.submit_array(worker_type, cmds, count=count, gpu=gpu, env_vars=env_vars)

    def stop(self, job_name: str):
        self.launcher.stop(job_name)

    def stop_all(self, signal=None):
        self.launcher.stop_all(signal=signal)

    def find(self, job_name: str):
        return self.launcher.find(job_name)

    def find_all(self, job_name_regex: str = ".*"):
        return self.launcher.find_all(job_name_regex)

    def wait(self, timeout=None, **kwargs):
        return self.launcher.wait(timeout=timeout, **kwargs)


SBATCH_ARRAY_TEMPLATE = """#!/bin/bash
#SBATCH --job-name={job_name}
#SBATCH --output={log_path}
#SBATCH --open-mode=append
#SBATCH --nodes={nodes}
#SBATCH --ntasks={ntasks}
#SBATCH --ntasks-per-node={ntasks_per_node}
#SBATCH --gres=gpu:{gpus_per_node}
#SBATCH --cpus-per-task={cpus_per_task}
#SBATCH --mem-per-cpu={mem_per_cpu}M
{extra}
{exports}
srun --ntasks={ntasks} --kill-on-bad-exit=1 {container} bash -c '{cmd}'
"""


class SlurmSchedulerClient(SchedulerClient):
    """One sbatch job per worker type; an array of `count` workers becomes `count` srun tasks, each
    deriving its worker index from ``$SLURM_PROCID`` and its device from ``$SLURM_LOCALID`` (the reference
    writes a multiprog file per job, ``realhf/scheduler/slurm/client.py:78-400``)."""

    def __init__(self, expr_name: str, trial_name: str, fileroot: str = "/tmp/areal", container_type: str = "none",
                 container_image: Optional[str] = None, container_mounts: Optional[str] = None,
                 allocate: bool = False, node_resources=None):
        """`allocate`: place the tasks of every array on nodes here (``scheduler/slurm_utils.py``: free resources
        from ``scontrol show node`` or `node_resources`, GPU arrays packed node by node so that neighbouring
        worker indices share a node) and pin the placement with a hostfile and a multi-prog file, instead of
        leaving the distribution to Slurm."""
        super().__init__(expr_name, trial_name)
        self.allocate = allocate
        self._free = node_resources  # {host: NodeResource}; queried on first use when None
        self.placements: Dict[str, object] = {}
        self.fileroot = fileroot
        self.container_type, self.container_image, self.container_mounts = container_type, container_image, \
            container_mounts
        self.job_ids: Dict[str, int] = {}

    def _dir(self) -> str:
        import getpass
        import os

        d = f"{self.fileroot}/logs/{getpass.getuser()}/{self.expr_name}/{self.trial_name}"
        os.makedirs(d, exist_ok=True)
        return d

    def render(self, worker_type: str, cmd: str, count: int, gpu: int = 0, cpu: int = 1, mem: int = 1024,
               env_vars: Optional[Dict[str, str]] = None, nodelist: Optional[str] = None,
               exclude: Optional[str] = None, time_limit: Optional[str] = None, gpus_per_node: int = 8) -> str:
        per_node = max(1, gpus_per_node // int(gpu)) if gpu else count
        per_node = min(per_node, count)
        nodes = (count + per_node - 1) // per_node
        extra = [f"#SBATCH --{k}={v}" for k, v in (("nodelist", nodelist), ("exclude", exclude), ("time", time_limit))
                 if v]
        env = dict(env_vars or {})
        if gpu:  # the task-local id only exists inside the srun step
            cmd = "export LOCAL_RANK=$SLURM_LOCALID; " + cmd
        exports = "\n".join(f"export {k}={v}" for k, v in env.items())
        container = ""
        if self.container_type == "apptainer" and self.container_image:
            mounts = f"--bind {self.container_mounts} " if self.container_mounts else ""
            container = f"apptainer exec --rocm {mounts}{self.container_image}"
        return SBATCH_ARRAY_TEMPLATE.format(
            job_name=f"{self.run_name}:{worker_type}", log_path=f"{self._dir()}/{worker_type}.log", nodes=nodes,
            ntasks=count, ntasks_per_node=per_node, gpus_per_node=int(gpu) * per_node if gpu else 0,
            cpus_per_task=cpu, mem_per_cpu=max(1, mem // max(1, cpu)), extra="\n".join(extra), exports=exports,
            container=container, cmd=cmd.replace("{index}", "$SLURM_PROCID").replace("'", "'\\''"))

    def render_allocated(self, worker_type: str, cmd: str, count: int, gpu: int = 0, cpu: int = 1, mem: int = 1024,
                         env_vars: Optional[Dict[str, str]] = None, nodelist: Optional[str] = None,
                         exclude: Optional[str] = None, time_limit: Optional[str] = None, **_) -> str:
        """sbatch script of an array whose tasks were placed by `allocate_resources`: the job asks for exactly
        the chosen nodes, `SLURM_HOSTFILE` + ``--distribution=arbitrary`` put task i on its host and the
        multi-prog file gives it its worker index and its device ordinal on that host."""
        from . import slurm_utils as su

        if self._free is None:
            self._free = su.get_all_node_resources()
        req = su.LaunchRequest(worker_type, count, su.NodeResource(cpu=cpu, mem=mem, gpu=int(gpu)), nodelist, exclude)
        (pl,), self._free = su.allocate_resources([req], self._free)
        self.placements[worker_type] = pl
        d = self._dir()
        with open(f"{d}/{worker_type}.hostfile", "w") as f:
            f.write(pl.hostfile())
        task = ("export LOCAL_RANK={local}; " if gpu else "") + cmd
        container = ""
        if self.container_type == "apptainer" and self.container_image:
            mounts = f"--bind {self.container_mounts} " if self.container_mounts else ""
            container = f"apptainer exec --rocm {mounts}{self.container_image} "
        with open(f"{d}/{worker_type}.multiprog", "w") as f:
            f.write(pl.multiprog(f"{container}bash -c '" + task.replace("'", "'\\''") + "'"))
        per_node = max(pl.hosts.count(h) for h in set(pl.hosts))
        lines = ["#!/bin/bash", f"#SBATCH --job-name={self.run_name}:{worker_type}",
                 f"#SBATCH --output={d}/{worker_type}.log", "#SBATCH --open-mode=append",
                 f"#SBATCH --nodelist={pl.nodelist()}", f"#SBATCH --nodes={len(set(pl.hosts))}",
                 f"#SBATCH --ntasks={count}", f"#SBATCH --cpus-per-task={cpu}",
                 f"#SBATCH --mem-per-cpu={max(1, mem // max(1, cpu))}M"]
        if gpu:
            lines.append(f"#SBATCH --gres=gpu:{int(gpu) * per_node}")
        if time_limit:
            lines.append(f"#SBATCH --time={time_limit}")
        lines += [f"export {k}={v}" for k, v in (env_vars or {}).items()]
        lines += [f"export SLURM_HOSTFILE={d}/{worker_type}.hostfile",
                  f"srun --ntasks={count} --distribution=arbitrary --kill-on-bad-exit=1 --multi-prog "
                  f"{d}/{worker_type}.multiprog"]
        return "\n".join(lines) + "\n"

    def submit(self, worker_type: str, cmd: str, **kwargs):
        return self.submit_array(worker_type, cmd, 1, **kwargs)

    def submit_array(self, worker_type: str, cmd: str, count: int, **kwargs):
        import re
        import subprocess

        script = (self.render_allocated if self.allocate else self.render)(worker_type, cmd, count, **kwargs)
        path = f"{self._dir()}/{worker_type}.sh"
        with open(path, "w") as f:
            f.write(script)
        out = subprocess.check_output(["sbatch", path]).decode()
        m = re.search(r"Submitted batch job (\d+)", out)
        if not m:
            raise RuntimeError(f"sbatch failed: {out}")
        self.job_ids[worker_type] = int(m.group(1))
        return self.job_ids[worker_type]

    def _states(self) -> Dict[str, JobInfo]:
        import subprocess

        from ..launcher.slurm import SlurmLauncher

        ids = ",".join(str(j) for j in self.job_ids.values())
        if not ids:
            return {}
        txt = subprocess.check_output(["squeue", "-h", "-j", ids, "-o", "%i %T %N"]).decode()
        parsed = SlurmLauncher.parse_squeue(txt)
        out = {}
        for name, jid in self.job_ids.items():
            st, host = parsed.get(jid, (JobState.COMPLETED, ""))
            out[name] = JobInfo(name=name, state=st, host=host, slurm_id=jid)
        return out

    def find(self, job_name: str):
        return self._states().get(job_name)

    def find_all(self, job_name_regex: str = ".*"):
        import re

        return [j for n, j in self._states().items() if re.fullmatch(job_name_regex, n)]

    def stop(self, job_name: str):
        import subprocess

        jid = self.job_ids.pop(job_name, None)
        if jid is not None:
            subprocess.call(["scancel", str(jid)])

    def stop_all(self, signal=None):
        for n in list(self.job_ids):
            self.stop(n)

    def wait(self, timeout=None, check_status=(JobState.CANCELLED, JobState.FAILED, JobState.NOT_FOUND),
             remove_status=(JobState.COMPLETED,), poll_interval: float = 5.0, **_):
        import time

        deadline = None if timeout is None else time.time() + timeout
        left = set(self.job_ids)
        while left:
            for n, j in self._states().items():
                if n in left and j.state in check_status:
                    raise JobException(self.run_name, n, j.host, j.state)
                if n in left and j.state in remove_status:
                    left.discard(n)
            if deadline is not None and time.time() > deadline:
                raise TimeoutError(f"Timeout waiting for {self.run_name}: {sorted(left)}")
            if left:
                time.sleep(poll_interval)


def make(mode: str, expr_name: str, trial_name: str, **kwargs) -> SchedulerClient:
    if mode == "local":
        return LocalSchedulerClient(expr_name, trial_name, **kwargs)
    if mode == "slurm":
        return SlurmSchedulerClient(expr_name, trial_name, **kwargs)
    raise NotImplementedError(f"Scheduler {mode} not found")
