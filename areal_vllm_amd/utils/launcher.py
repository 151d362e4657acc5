"""Shared launcher vocabulary: job states, environment assembly, server discovery, config checks.

Same contracts as the reference (``areal/utils/launcher.py:1-137``): a ``JobException`` carries the
failed worker type, ``get_env_vars`` merges the framework's base environment with a user string of
``K=V,K2=V2`` pairs, ``wait_llm_server_addrs`` blocks until the expected number of rollout servers has
registered under ``gen_servers`` in name_resolve.
"""
import dataclasses
import enum
import getpass
import os
import pathlib
import time
from typing import Dict, List, Optional

from . import logging, name_resolve, names

logger = logging.getLogger("launcher-utils")

LOCAL_CACHE_DIR = "/tmp/areal"
PYTORCH_KERNEL_CACHE_PATH = f"{LOCAL_CACHE_DIR}/.cache/{getpass.getuser()}/torch/kernels/"
os.makedirs(PYTORCH_KERNEL_CACHE_PATH, exist_ok=True)

BASE_ENVIRONS = {
    "TOKENIZERS_PARALLELISM": "true",
    "PYTORCH_KERNEL_CACHE_PATH": PYTORCH_KERNEL_CACHE_PATH,
    "PYTHONPATH": str(pathlib.Path(__file__).resolve().parent.parent.parent),
    # dmabuf IPC between processes sharing a GPU buffer (RCCL, tensor sharing)
    "HSA_ENABLE_IPC_MODE_LEGACY": "0",
}
# RCCL knobs for one xGMI-connected node; harmless elsewhere.
RCCL_ENVIRONS = {
    "NCCL_DEBUG": "WARN",
    "TORCH_NCCL_ASYNC_ERROR_HANDLING": "1",
}


class JobState(enum.Enum):
    NOT_FOUND = 0
    PENDING = 1
    RUNNING = 2
    COMPLETED = 3
    FAILED = 4
    CANCELLED = 5

    def active(self):
        return self in (JobState.PENDING, JobState.RUNNING)


@dataclasses.dataclass
class JobInfo:
    name: str
    state: JobState
    host: Optional[str] = None
    submit_time: Optional[str] = None
    start_time: Optional[str] = None
    slurm_id: Optional[int] = None
    exit_code: Optional[int] = None


class JobException(Exception):
    def __init__(self, run_name, worker_type, host, reason: JobState):
        super().__init__(f"Job {run_name}:{worker_type} {reason} at node {host}")
        self.run_name = run_name
        self.worker_type = worker_type
        self.host = host
        self.reason = reason


def get_env_vars(cluster_name: str = "local", additional_env_vars: Optional[str] = None) -> Dict[str, str]:
    env: Dict[str, str] = {}
    if additional_env_vars:
        for item in additional_env_vars.split(","):
            if not item.strip():
                continue
            k, _, v = item.partition("=")
            env[k.strip()] = v.strip()
    merged = {**BASE_ENVIRONS, **RCCL_ENVIRONS, **env}
    if os.environ.get("PYTHONPATH"):
        merged["PYTHONPATH"] = merged["PYTHONPATH"] + os.pathsep + os.environ["PYTHONPATH"]
    return merged


def device_env(device_ids: List[str]) -> Dict[str, str]:
    """Environment that restricts a child process to the given GPUs."""
    ids = ",".join(str(d) for d in device_ids)
    return {"HIP_VISIBLE_DEVICES": ids, "CUDA_VISIBLE_DEVICES": ids}


def visible_devices() -> List[str]:
    for k in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        if os.environ.get(k):
            return [x for x in os.environ[k].split(",") if x != ""]
    import torch

    return [str(i) for i in range(torch.cuda.device_count())]


def wait_llm_server_addrs(experiment_name: str, trial_name: str, n_rollout_servers: int,
                          timeout: float = 600.0, check=None, check_interval: float = 5.0) -> List[str]:
    """Addresses of the rollout servers once all of them have registered. `check()` runs every
    `check_interval` seconds while waiting and should raise when the servers can no longer come up (their job
    failed): the caller then fails at once instead of after `timeout`."""
    key = names.gen_servers(experiment_name, trial_name)
    start = time.perf_counter()
    last_check = start
    while True:
        if check is not None and time.perf_counter() - last_check >= check_interval:
            last_check = time.perf_counter()
            check()
        addrs = name_resolve.get_subtree(key)
        if len(addrs) >= n_rollout_servers:
            logger.info(f"Found {len(addrs)} rollout servers: {', '.join(sorted(addrs))}")
            return sorted(addrs)
        if time.perf_counter() - start > timeout:
            raise TimeoutError(f"Timeout waiting for rollout servers: {len(addrs)}/{n_rollout_servers} registered")
        time.sleep(1)


def validate_config_for_distributed_launcher(config):
    """Multi-node launchers need every rollout replica to fit in whole nodes or divide one."""
    from ..api.alloc_mode import AllocationMode

    am = AllocationMode.from_str(config.allocation_mode)
    per_node = config.cluster.n_gpus_per_node
    if am.gen is not None and am.gen.world_size > 0:
        inst = am.gen.tp_size * am.gen.pp_size
        assert inst <= per_node and per_node % inst == 0 or inst % per_node == 0, (
            f"a rollout replica of {inst} GPUs neither divides nor is a multiple of a {per_node}-GPU node")
    return am
