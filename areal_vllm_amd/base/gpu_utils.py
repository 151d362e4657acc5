"""Device visibility helpers for processes that are started without a launcher-provided device list.

Workers of one node agree on who takes which GPU through name_resolve: each registers under the node's
key, sorts the registrations and takes the device with its own position (``isolate_device``), then narrows
``HIP_VISIBLE_DEVICES`` to it so that device 0 is "its" GPU in every library. Role of
``realhf/base/gpu_utils.py`` (CUDA_VISIBLE_DEVICES there)."""
import os
import socket
import time
from typing import List, Optional

from ..utils import name_resolve, names

VISIBLE_ENV = ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")


def gpu_count() -> int:
    """GPUs visible to this process without initialising the runtime on them."""
    for k in VISIBLE_ENV:
        v = os.environ.get(k)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    try:
        import torch

        return torch.cuda.device_count()
    except Exception:
        return 0


def set_visible_devices(devices: List[int]):
    v = ",".join(str(d) for d in devices)
    for k in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        os.environ[k] = v


def isolate_device(experiment_name: str, trial_name: str, worker_type: str, worker_index: int, n_workers_on_node: int,
                   n_gpus_per_node: int = 8, timeout: float = 300.0) -> Optional[int]:
    """Pick this worker's GPU among the workers of its host and hide the others. Returns the physical
    device index (None when the launcher already narrowed visibility to one device)."""
    if gpu_count() == 1:
        return None
    host = socket.gethostname()
    root = names.trial_root(experiment_name, trial_name) + f"/gpu_isolation/{host}/{worker_type}"
    name_resolve.add_subentry(root, str(worker_index))
    deadline = time.monotonic() + timeout
    while True:
        peers = sorted(int(x) for x in name_resolve.get_subtree(root))
        if len(peers) >= n_workers_on_node:
            break
        if time.monotonic() > deadline:
            raise TimeoutError(f"only {len(peers)} of {n_workers_on_node} workers registered on {host}")
        time.sleep(0.1)
    local_rank = peers.index(worker_index)
    assert local_rank < n_gpus_per_node, f"{n_workers_on_node} workers on {host} but {n_gpus_per_node} GPUs"
    set_visible_devices([local_rank])
    return local_rank
