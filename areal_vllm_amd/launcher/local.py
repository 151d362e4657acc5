"""Single-node launcher: rollout servers + a torchrun trainer, with a recover loop.

CLI and environment contract follow the reference (``areal/launcher/local.py:73-363``):
``python -m areal_vllm_amd.launcher.local <entry.py> --config <yaml> [k=v ...]`` starts one native
rollout server per generation data-parallel replica (GPUs handed out round-robin), then the trainer
under ``torch.distributed.run`` with ``AREAL_LLM_SERVER_ADDRS`` and ``AREAL_RECOVER_RUN`` set, watches
all children, and on a failure re-launches the whole experiment when ``recover.mode`` allows it.
Children are plain ``Popen`` objects with their own environment dict and log file (no shell pipeline),
so exit codes are exact: a trainer that exits 0 ends the experiment, anything else is a failure.

Colocated allocation (``native:d8+native:d8`` style, or a trainer-only mode with
``rollout.colocated``) starts no server processes: every trainer rank embeds its rollout engine.
"""
import getpass
import os
import re
import signal as signal_module
import subprocess
import sys
import time
from collections import defaultdict
from typing import Dict, List, Optional, Tuple, Union

import psutil

from ..api.alloc_mode import AllocationMode, AllocationType
from ..api.cli_args import (
    BaseExperimentConfig,
    InferenceServerConfig,
    parse_cli_args,
    resolve_interpolations,
    to_structured_cfg,
)
from ..utils import logging, name_resolve, names
from ..utils.launcher import JobException, JobInfo, JobState, device_env, get_env_vars, visible_devices
from ..utils.network import find_free_ports, gethostip
from ..utils.recover import check_if_recover

logger = logging.getLogger("local-launcher")
RECOVER_TIME_INTERVAL = 10  # seconds


def _as_signal(sig: Optional[Union[str, int]]) -> int:
    """None -> SIGKILL, "SIGTERM" / "TERM" -> its number, numbers pass through."""
    if sig is None:
        return int(signal_module.SIGKILL)
    if isinstance(sig, str):
        name = sig if sig.startswith("SIG") else "SIG" + sig
        return int(getattr(signal_module, name))
    return int(sig)


def terminate_process_and_children(pid: int, signal: Optional[Union[str, int]] = None):
    """Signal a process and everything it has started, deepest descendants first (a server wrapper's servers run in
    sessions of their own, so the process group is not enough). Processes that are already gone are skipped."""
    signum = _as_signal(signal)
    try:
        root = psutil.Process(pid)
        family = root.children(recursive=True)[::-1] + [root]
    except psutil.NoSuchProcess:
        return
    for proc in family:
        try:
            proc.send_signal(signum)
        except psutil.NoSuchProcess:
            continue


class LocalLauncher:
    def __init__(self, experiment_name: str, trial_name: str, fileroot: str, devices: Optional[List[str]] = None):
        self.experiment_name, self.trial_name, self.fileroot = experiment_name, trial_name, fileroot
        self._jobs: Dict[str, subprocess.Popen] = {}
        self._logs: Dict[str, object] = {}
        self._job_counter: Dict[str, int] = defaultdict(int)
        self._job_states: Dict[str, JobState] = {}
        self._gpu_counter = 0
        self._devices = devices if devices is not None else visible_devices()

    @property
    def run_name(self):
        return f"{self.experiment_name}_{self.trial_name}"

    def log_path_of(self, job_name: str) -> str:
        d = f"{self.fileroot}/logs/{getpass.getuser()}/{self.experiment_name}/{self.trial_name}"
        os.makedirs(d, exist_ok=True)
        return os.path.join(d, f"{job_name}.log")

    def submit_array(self, job_name: str, cmd: Union[str, List[str]], count: int = 1, gpu: int = 0,
                     env_vars: Optional[Dict[str, str]] = None):
        cmds = cmd if isinstance(cmd, list) else [cmd] * count
        offset = self._job_counter[job_name]
        for i in range(count):
            env = dict(os.environ)
            env.update(env_vars or {})
            if gpu > 0:
                if not self._devices:
                    logger.warning("no GPU visible to the local launcher: children run on the CPU platform")
                    # many ranks share the host's cores: one math thread each (what torchrun does for its
                    # workers), or every process spins a full OpenMP team
                    env.setdefault("OMP_NUM_THREADS", "1")
                else:
                    ids = []
                    for _ in range(gpu):
                        ids.append(self._devices[self._gpu_counter % len(self._devices)])
                        self._gpu_counter += 1
                    env.update(device_env(ids))
            log = open(self.log_path_of(job_name), "ab", buffering=0)
            argv = cmds[i] if isinstance(cmds[i], list) else ["bash", "-c", cmds[i]]
            logger.info(f"Starting local process {job_name}/{offset + i}: {cmds[i]}")
            p = subprocess.Popen(argv, env=env, stdout=log, stderr=subprocess.STDOUT, start_new_session=True)
            key = f"{job_name}/{offset + i}"
            self._jobs[key] = p
            self._logs[key] = log
            self._job_counter[job_name] += 1

    def submit(self, job_name: str, cmd, gpu: int = 0, env_vars: Optional[Dict[str, str]] = None):
        self.submit_array(job_name, [cmd], 1, gpu, env_vars)

    def stop(self, job_name: str, signal=None):
        keys = [k for k in self._jobs if k.split("/")[0] == job_name]
        for k in keys:
            terminate_process_and_children(self._jobs[k].pid, signal=signal)
        deadline = time.time() + 20
        for k in keys:
            p = self._jobs.pop(k)
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                terminate_process_and_children(p.pid)
                p.wait()
            self._logs.pop(k).close()
        self._job_counter.pop(job_name, None)

    def stop_all(self, signal=None):
        for name in list(self._job_counter):
            self.stop(name, signal=signal)

    def _state_of(self, key: str) -> JobState:
        p = self._jobs.get(key)
        if p is None:
            return JobState.NOT_FOUND
        rc = p.poll()
        if rc is None:
            return JobState.RUNNING
        return JobState.COMPLETED if rc == 0 else JobState.FAILED

    def find(self, job_name: str) -> JobInfo:
        st = self._state_of(job_name)
        p = self._jobs.get(job_name)
        return JobInfo(name=job_name, state=st, host="localhost", exit_code=None if p is None else p.poll())

    def find_all(self, job_name_regex: str = ".*") -> List[JobInfo]:
        return [self.find(n) for n in self._jobs if re.fullmatch(job_name_regex, n)]

    def wait(self, timeout: Optional[float] = None,
             check_status: Tuple[JobState, ...] = (JobState.CANCELLED, JobState.FAILED, JobState.NOT_FOUND),
             remove_status: Tuple[JobState, ...] = (JobState.COMPLETED,), update: bool = False,
             poll_interval: float = 1.0):
        deadline = None if timeout is None else time.time() + timeout
        left = set(self._jobs)
        while left:
            if deadline is not None and time.time() > deadline:
                raise TimeoutError(f"Timeout waiting for {self.run_name}: {', '.join(sorted(left))}")
            for key in list(left):
                st = self._state_of(key)
                self._job_states[key] = st
                if st in check_status:
                    raise JobException(self.run_name, key.split("/")[0], "local", st)
                if st in remove_status:
                    logger.info(f"Job {key} is {st}. (Removed)")
                    left.discard(key)
                    if update:
                        self._jobs.pop(key, None)
            if left:
                time.sleep(poll_interval)


def load_launcher_config(argv: List[str]):
    """Parse `--config x.yaml k=v ...` into (BaseExperimentConfig view, raw dict, config path)."""
    import dataclasses

    raw, config_file = parse_cli_args(argv)
    raw = resolve_interpolations(raw)
    known = {f.name for f in dataclasses.fields(BaseExperimentConfig)}
    view = to_structured_cfg({k: v for k, v in raw.items() if k in known}, BaseExperimentConfig)
    return view, raw, str(config_file)


def main():
    if len(sys.argv) < 2:
        print(__doc__)
        sys.exit(2)
    config, raw, _ = load_launcher_config(sys.argv[2:])
    local_main(config, raw, run_id=0)


def local_main(config: BaseExperimentConfig, raw: Optional[dict] = None, run_id: int = 0,
               entry_argv: Optional[List[str]] = None):
    raw = raw or {}
    entry_argv = entry_argv if entry_argv is not None else sys.argv[1:]
    is_recover_run = check_if_recover(config.recover, run_id)
    launcher = LocalLauncher(config.experiment_name, config.trial_name, config.cluster.fileroot)
    name_resolve.reconfigure(config.cluster.name_resolve)
    name_resolve.clear_subtree(names.trial_root(config.experiment_name, config.trial_name))
    alloc_mode = AllocationMode.from_str(config.allocation_mode)
    logger.info(f"LocalLauncher: experiment_name={config.experiment_name}, trial_name={config.trial_name}, "
                f"fileroot={config.cluster.fileroot}, run_id={run_id}, is_recover_run={is_recover_run}")

    server_addrs: List[str] = []
    colocated = alloc_mode.type_ == AllocationType.COLOCATE
    if alloc_mode.gen is not None and alloc_mode.gen.world_size > 0 and not colocated:
        scfg: InferenceServerConfig = config.sglang
        base_seed = scfg.random_seed
        n = alloc_mode.gen.dp_size
        ports = find_free_ports(n * 2, port_range=(10000, 50000))
        host = "127.0.0.1" if not scfg.enable_metrics else gethostip()
        cmds = []
        for i in range(n):
            scfg.random_seed = base_seed + i
            cmds.append(InferenceServerConfig.build_cmd(scfg, host=host, tp_size=alloc_mode.gen.tp_size, base_gpu_id=0,
                                                        port=ports[2 * i], dist_init_addr=f"127.0.0.1:{ports[2 * i + 1]}"))
            server_addrs.append(f"{host}:{ports[2 * i]}")
        scfg.random_seed = base_seed
        launcher.submit_array("llm_server", cmds, count=n, gpu=alloc_mode.gen.pp_size * alloc_mode.gen.tp_size,
                              env_vars=get_env_vars(config.cluster.cluster_name,
                                                    config.launcher.inference_server_env_vars))
        logger.info(f"LLM inference server launched at: AREAL_LLM_SERVER_ADDRS={','.join(server_addrs)}")

    if alloc_mode.type_ != AllocationType.LLM_SERVER_ONLY:
        if alloc_mode.type_ == AllocationType.DECOUPLED_EVAL:
            gpu, nprocs = 0, 1
        else:
            gpu = nprocs = alloc_mode.train.world_size
        port = find_free_ports(1, (10000, 50000))[0]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes", "1", "--nproc-per-node", str(nprocs),
               "--master-addr", "127.0.0.1", "--master-port", str(port), *entry_argv]
        launcher.submit("trainer", cmd, gpu=gpu, env_vars=dict(
            **get_env_vars(config.cluster.cluster_name, config.launcher.trainer_env_vars),
            AREAL_LLM_SERVER_ADDRS=",".join(server_addrs), AREAL_RECOVER_RUN=str(int(is_recover_run))))

    has_trainer = alloc_mode.type_ != AllocationType.LLM_SERVER_ONLY
    try:
        while True:
            infos = {i.name: i for i in launcher.find_all()}
            for i in infos.values():
                worker = i.name.split("/")[0]
                # a rollout server never finishes on its own: any exit is a failure
                if i.state == JobState.FAILED or (worker != "trainer" and i.state == JobState.COMPLETED):
                    raise JobException(launcher.run_name, worker, "local", JobState.FAILED)
            if has_trainer and infos["trainer/0"].state == JobState.COMPLETED:
                logger.info("trainer completed; stopping rollout servers")
                break
            time.sleep(1.0)
        launcher.stop_all("SIGTERM")
    except (KeyboardInterrupt, JobException) as e:
        launcher.stop_all("SIGTERM")
        _maybe_recover(e, config, raw, run_id, entry_argv)


def _maybe_recover(e, config, raw, run_id, entry_argv):
    if isinstance(e, JobException):
        recover_this = (e.reason in (JobState.FAILED, JobState.NOT_FOUND) and run_id < config.recover.retries
                        and config.recover.mode in ("auto", "fault"))
        if recover_this:
            logger.warning(f"{e}; relaunching (run {run_id + 1} of at most {config.recover.retries})")
            time.sleep(RECOVER_TIME_INTERVAL)
            return local_main(config, raw, run_id=run_id + 1, entry_argv=entry_argv)
    raise e


if __name__ == "__main__":
    main()
