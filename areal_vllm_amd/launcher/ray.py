"""Ray launcher: placement groups for rollout servers and trainer ranks on a Ray cluster.

``python -m areal_vllm_amd.launcher.ray <entry.py> --config <yaml> [k=v ...]``. Each rollout node gets
a ``server_wrapper`` task pinned to a placement-group bundle; each trainer rank is a remote task that
sets the torch.distributed environment (``RANK``, ``WORLD_SIZE``, ``MASTER_ADDR``, ...) and then runs
the entry module's ``main(argv)`` in-process. Contract of ``areal/launcher/ray.py:44-523``
(``run_func`` + env hook, ``submit_array`` over placement groups, wait / recover loop). Ray itself is an
optional dependency: importing this module without it works, constructing the launcher raises.
"""
import importlib.util
import os
import sys
import time
from typing import Callable, Dict, List, Optional

from ..api.alloc_mode import AllocationMode, AllocationType
from ..api.cli_args import BaseExperimentConfig
from ..utils import logging, name_resolve, names
from ..utils.launcher import (
    JobException,
    JobState,
    get_env_vars,
    validate_config_for_distributed_launcher,
    wait_llm_server_addrs,
)
from ..utils.recover import check_if_recover

logger = logging.getLogger("ray-launcher")

try:  # optional
    import ray
    from ray.util.placement_group import placement_group, remove_placement_group
    from ray.util.scheduling_strategies import PlacementGroupSchedulingStrategy
except ImportError:  # pragma: no cover
    ray = None


def run_func(file_path: str, function_name: str, env: Optional[Dict[str, str]], *args, **kwargs):
    """Import `file_path` - a script path, or the dotted name of an installed module (modules of this package
    use relative imports and cannot be loaded from their file path) - and call `function_name` (executed inside a
    Ray worker)."""
    if env:
        os.environ.update(env)
    if not file_path.endswith(".py"):
        module = importlib.import_module(file_path)
        return getattr(module, function_name)(*args, **kwargs)
    module_name = os.path.splitext(os.path.basename(file_path))[0]
    spec = importlib.util.spec_from_file_location(module_name, file_path)
    module = importlib.util.module_from_spec(spec)
    sys.modules[module_name] = module
    spec.loader.exec_module(module)
    return getattr(module, function_name)(*args, **kwargs)


def torch_env_hook(n_tasks: int, master_addr: str, master_port: int, gpus_per_node: int) -> List[Dict[str, str]]:
    """Per-task torch.distributed environment for `n_tasks` ranks laid out node by node."""
    return [dict(RANK=str(i), WORLD_SIZE=str(n_tasks), LOCAL_RANK=str(i % gpus_per_node),
                 MASTER_ADDR=master_addr, MASTER_PORT=str(master_port)) for i in range(n_tasks)]


class RayLauncher:
    def __init__(self, experiment_name: str, trial_name: str, fileroot: str):
        if ray is None:
            raise RuntimeError("the Ray launcher needs the `ray` package, which is not installed")
        self.experiment_name, self.trial_name, self.fileroot = experiment_name, trial_name, fileroot
        self.jobs: Dict[str, "ray.ObjectRef"] = {}
        self.placement_groups: Dict[str, object] = {}

    @property
    def run_name(self):
        return f"{self.experiment_name}_{self.trial_name}"

    def submit_array(self, job_name: str, file_path: str, func_name: str, count: int, nodes: int, gpus_per_task: int,
                     cpus_per_task: int, mem_per_task: int, list_args: List[List], env_vars: Optional[Dict] = None,
                     env_hook: Optional[Callable[[object], List[Dict]]] = None):
        tasks_per_node = count // nodes
        bundles = [{"CPU": cpus_per_task * tasks_per_node, "GPU": gpus_per_task * tasks_per_node,
                    "memory": mem_per_task * tasks_per_node * 1024 * 1024}] * nodes
        pg = placement_group(bundles, strategy="STRICT_SPREAD" if nodes > 1 else "PACK")
        ray.get(pg.ready(), timeout=600)
        self.placement_groups[job_name] = pg
        extra = env_hook(pg) if env_hook else [{} for _ in range(count)]
        remote = ray.remote(run_func)
        for i in range(count):
            env = {**(env_vars or {}), **extra[i]}
            ref = remote.options(
                num_cpus=cpus_per_task, num_gpus=gpus_per_task, memory=mem_per_task * 1024 * 1024,
                scheduling_strategy=PlacementGroupSchedulingStrategy(pg, placement_group_bundle_index=i // tasks_per_node,
                                                                    placement_group_capture_child_tasks=True),
                runtime_env={"env_vars": {k: str(v) for k, v in env.items()}},
            ).remote(file_path, func_name, env, *list_args[i])
            self.jobs[f"{job_name}:{i}"] = ref

    def check_running(self, job_name: str):
        """Raise if a task of `job_name` has ended (servers are meant to run until they are cancelled)."""
        refs = [v for k, v in self.jobs.items() if k.split(":")[0] == job_name]
        done, _ = ray.wait(refs, num_returns=1, timeout=0) if refs else ([], [])
        if done:
            raise JobException(self.run_name, job_name, "ray", JobState.FAILED)

    def stop(self, job_name: str, force: bool = False):
        for k in [k for k in self.jobs if k.split(":")[0] == job_name]:
            ray.cancel(self.jobs.pop(k), force=force)
        pg = self.placement_groups.pop(job_name, None)
        if pg is not None:
            remove_placement_group(pg)

    def stop_all(self, force: bool = False):
        for n in {k.split(":")[0] for k in self.jobs}:
            self.stop(n, force)

    def wait(self, check_status=(JobState.FAILED,), remove_status=(JobState.COMPLETED,), until: Optional[str] = None):
        """Block until every job (or every job of worker type `until`) finished; raise on a failure."""
        while self.jobs:
            done, _ = ray.wait(list(self.jobs.values()), num_returns=1, timeout=5)
            for ref in done:
                name = next(k for k, v in self.jobs.items() if v == ref)
                try:
                    ray.get(ref)
                except Exception as e:  # ray.exceptions.RayTaskError and friends
                    logger.error(f"job {name} failed: {e}")
                    raise JobException(self.run_name, name.split(":")[0], "ray", JobState.FAILED)
                self.jobs.pop(name)
            if until is not None and not any(k.split(":")[0] == until for k in self.jobs):
                return


def ray_main(config: BaseExperimentConfig, run_id: int = 0, entry_argv: Optional[List[str]] = None):
    entry_argv = entry_argv if entry_argv is not None else sys.argv[1:]
    is_recover_run = check_if_recover(config.recover, run_id)
    am = validate_config_for_distributed_launcher(config)
    ray.init(ignore_reinit_error=True)
    launcher = RayLauncher(config.experiment_name, config.trial_name, config.cluster.fileroot)
    name_resolve.reconfigure(config.cluster.name_resolve)
    name_resolve.clear_subtree(names.trial_root(config.experiment_name, config.trial_name))
    per_node = config.cluster.n_gpus_per_node
    lc = config.launcher
    try:
        n_servers = 0
        if am.gen is not None and am.gen.world_size > 0 and am.type_ != AllocationType.COLOCATE:
            nodes = max(1, am.gen.world_size // per_node)
            gpn = min(per_node, am.gen.world_size)
            n_servers = am.gen.dp_size
            launcher.submit_array("llm_server", "areal_vllm_amd.launcher.server_wrapper", "main", count=nodes, nodes=nodes, gpus_per_task=gpn,
                                  cpus_per_task=lc.inference_server_cpus_per_gpu * gpn,
                                  mem_per_task=lc.inference_server_mem_per_gpu * gpn,
                                  list_args=[[entry_argv[1:]] for _ in range(nodes)],
                                  env_vars=get_env_vars(config.cluster.cluster_name, lc.inference_server_env_vars))
        if am.type_ != AllocationType.LLM_SERVER_ONLY:
            addrs = wait_llm_server_addrs(config.experiment_name, config.trial_name, n_servers,
                                          check=lambda: launcher.check_running("llm_server")) if n_servers else []
            world = 1 if am.type_ == AllocationType.DECOUPLED_EVAL else am.train.world_size
            nodes = max(1, world // per_node)

            def hook(pg):
                from ..utils.network import find_free_ports

                head = ray.get(ray.remote(lambda: ray.util.get_node_ip_address()).options(
                    scheduling_strategy=PlacementGroupSchedulingStrategy(pg, placement_group_bundle_index=0)).remote())
                return torch_env_hook(world, head, find_free_ports(1)[0], per_node)

            env = dict(**get_env_vars(config.cluster.cluster_name, lc.trainer_env_vars),
                       AREAL_LLM_SERVER_ADDRS=",".join(addrs), AREAL_RECOVER_RUN=str(int(is_recover_run)))
            launcher.submit_array("trainer", entry_argv[0], "main", count=world, nodes=nodes,
                                  gpus_per_task=0 if am.type_ == AllocationType.DECOUPLED_EVAL else 1,
                                  cpus_per_task=lc.trainer_cpus_per_gpu, mem_per_task=lc.trainer_mem_per_gpu,
                                  list_args=[[entry_argv[1:]] for _ in range(world)], env_vars=env, env_hook=hook)
            launcher.wait(until="trainer")
        else:
            launcher.wait()
        launcher.stop_all()
    except (KeyboardInterrupt, JobException) as e:
        launcher.stop_all(force=True)
        if isinstance(e, JobException) and run_id < config.recover.retries and config.recover.mode in ("auto", "fault"):
            time.sleep(10)
            return ray_main(config, run_id + 1, entry_argv)
        raise


def main():
    from .local import load_launcher_config

    if ray is None:
        raise RuntimeError("the Ray launcher needs the `ray` package, which is not installed")
    config, _raw, _ = load_launcher_config(sys.argv[2:])
    ray_main(config, 0)


if __name__ == "__main__":
    main()
