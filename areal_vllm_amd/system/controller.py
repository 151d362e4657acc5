"""Experiment controller: configures the worker processes, starts them and supervises the run.

The scheduler has already started one process per worker; each registered its control address in
name_resolve. The controller builds the experiment's configuration, connects to every worker, sends
``configure`` (worker information + its configuration object), then ``start``, and afterwards watches the
worker statuses: the run ends when the master worker completes; a worker in ERROR / LOST state fails the
run (the launcher's recover loop may then relaunch it); on the way out every worker is told to exit.
Behaviour of ``realhf/system/controller.py:98-422``.
"""
import re
import time
from typing import Any, Dict, List, Optional, Tuple

from ..utils import logging, name_resolve, names
from .system_api import WORKER_FIELDS, Experiment, ExperimentConfig
from .worker_base import ExpStatus, WorkerControlPanel, WorkerException, WorkerServerStatus

logger = logging.getLogger("controller")
CONNECTION_RETRY_AFTER_SECONDS = 360


class Controller:
    def __init__(self, experiment_name: str, trial_name: str, panel: Optional[WorkerControlPanel] = None):
        self.experiment_name, self.trial_name = experiment_name, trial_name
        self.panel = panel or WorkerControlPanel(experiment_name, trial_name)
        self.worker_names: Dict[str, List[str]] = {}

    def _check_consistency(self, setup: ExperimentConfig, scheduling):
        """The scheduling (what was started) and the configuration (what will be configured) must agree."""
        for wt in WORKER_FIELDS:
            n_cfg = len(getattr(setup, wt))
            tg = getattr(scheduling, wt, None)
            n_sched = 0 if tg is None else tg.count
            if n_cfg != n_sched:
                raise ValueError(f"scheduling starts {n_sched} {wt} process(es) but the experiment configures {n_cfg}")

    def start(self, experiment: Experiment, ignore_worker_error: bool = False, poll_interval: float = 1.0,
              timeout: Optional[float] = None):
        scheduling = experiment.scheduling_setup()
        setup = experiment.initial_setup()
        setup.lazy_init()
        self._check_consistency(setup, scheduling)
        setup.set_worker_information(self.experiment_name, self.trial_name)
        name_resolve.add(names.experiment_status(self.experiment_name, self.trial_name), ExpStatus.RUNNING.value,
                         replace=True)
        all_names: List[str] = []
        for wt in WORKER_FIELDS:
            self.worker_names[wt] = [f"{wt}/{i}" for i in range(len(getattr(setup, wt)))]
            all_names += self.worker_names[wt]
        logger.info(f"Connecting to {len(all_names)} workers...")
        self.panel.connect(all_names, timeout=CONNECTION_RETRY_AFTER_SECONDS)
        try:
            for wt in WORKER_FIELDS:
                for w, cfg in zip(self.worker_names[wt], getattr(setup, wt)):
                    self.panel.request(w, "configure", worker_info=cfg.worker_info, config=cfg)
            logger.info("Configuration of all workers has finished; starting them.")
            self.panel.group_request("start")
            self.wait(poll_interval=poll_interval, ignore_worker_error=ignore_worker_error, timeout=timeout)
        except Exception:
            self.interrupt()
            raise
        self.stop()

    def wait(self, poll_interval: float = 1.0, ignore_worker_error: bool = False, timeout: Optional[float] = None):
        deadline = None if timeout is None else time.time() + timeout
        masters = self.worker_names["master_worker"]
        while True:
            st = self.panel.pulse()
            if all(st[m] in (WorkerServerStatus.COMPLETED, WorkerServerStatus.PAUSED) for m in masters):
                logger.info("The master worker has finished: experiment complete.")
                return
            for w, s in st.items():
                if s in (WorkerServerStatus.ERROR, WorkerServerStatus.LOST) and not ignore_worker_error:
                    raise WorkerException(w, s, "experiment is running")
            if deadline is not None and time.time() > deadline:
                raise TimeoutError("experiment did not finish in time")
            time.sleep(poll_interval)

    def stop(self):
        name_resolve.add(names.experiment_status(self.experiment_name, self.trial_name), ExpStatus.COMPLETE.value,
                         replace=True)
        for w in list(self.panel.socks):
            try:
                self.panel.request(w, "exit")
            except Exception:  # the worker may have exited on its own (status watch)
                pass
        self.panel.close()

    def interrupt(self):
        name_resolve.add(names.experiment_status(self.experiment_name, self.trial_name), ExpStatus.ABORTED.value,
                         replace=True)
        for w in list(self.panel.socks):
            try:
                self.panel.request(w, "exit")
            except Exception:
                pass
        self.panel.close()


# ------------------------------------------------------------------------------------- Ray
def run_ray_worker(worker_type: str, index: int, experiment_name: str, trial_name: str, env: Dict[str, str]):
    """Body of the Ray task that hosts one worker: same loop as the process entry of the other launchers
    (``apps/remote.py``); the worker registers its control address in name_resolve, where the controller finds it."""
    import os

    os.environ.update({k: str(v) for k, v in env.items()})
    if env.get("AREAL_PLATFORM") == "cpu":
        os.environ["HIP_VISIBLE_DEVICES"] = os.environ["CUDA_VISIBLE_DEVICES"] = ""
    from .worker_base import run_worker

    run_worker(worker_type, experiment_name, trial_name, f"{worker_type}/{index}")


def plan_ray_placement(worker_counts: List[Tuple[str, int, Any]], available: Dict[str, float]
                       ) -> List[Tuple[str, int, Dict[str, Any]]]:
    """-> [(worker type, index, options of ``ray.remote``)] for every worker, given ``ray.available_resources()``.

    CPU-only workers take Ray's default placement. GPU workers (one GPU each) are packed: worker i of a type goes
    to node i // gpus_per_node, pinned through the node's ``node:<ip>`` resource, so that consecutive model
    workers - tensor / pipeline neighbours of the rank grid - share a node and its xGMI links. Raises when the
    cluster cannot hold the request (the experiment would otherwise wait forever for placement).
    Placement rule of ``realhf/system/controller.py:515-640``."""
    need_cpu = sum(tg.scheduling.cpu * n for _, n, tg in worker_counts)
    need_gpu = sum(tg.scheduling.gpu * n for _, n, tg in worker_counts)
    need_mem = sum(tg.scheduling.mem * n for _, n, tg in worker_counts) / 1024  # GB
    have_cpu, have_gpu = available.get("CPU", 0), available.get("GPU", 0)
    have_mem = available.get("memory", 0) / 1024 ** 3
    if have_cpu < need_cpu or have_gpu < need_gpu or have_mem < need_mem:
        raise RuntimeError(f"Ray does not have the resources for this experiment: required {need_cpu} CPU, {need_gpu} "
                           f"GPU, {need_mem:.1f} GB memory; available {have_cpu} CPU, {have_gpu} GPU, {have_mem:.1f} GB")
    nodes = sorted(k for k in available if re.fullmatch(r"node:(?:\d{1,3}\.){3}\d{1,3}", k))
    plan: List[Tuple[str, int, Dict[str, Any]]] = []
    next_gpu_slot = 0  # GPU workers of all types share the nodes: slots are handed out consecutively
    for wt, n, tg in worker_counts:
        sch = tg.scheduling
        base = dict(num_cpus=sch.cpu, num_gpus=sch.gpu, memory=int(sch.mem) * 1024 ** 2)
        if sch.gpu == 0:
            plan += [(wt, i, dict(base, name=f"{wt}/{i}")) for i in range(n)]
            continue
        if sch.gpu != 1:
            raise ValueError("GPU workers are scheduled one GPU each")
        if not nodes or have_gpu % len(nodes) != 0:
            raise ValueError("cannot pack GPU workers: the Ray nodes hold different numbers of GPUs")
        per_node = int(have_gpu // len(nodes))
        for i in range(n):
            node = nodes[(next_gpu_slot + i) // per_node]
            plan.append((wt, i, dict(base, name=f"{wt}/{i}", resources={node: 1.0 / per_node})))
        next_gpu_slot += n
    return plan


class RayController:
    """Runs the worker runtime on a Ray cluster: every worker is a Ray task with its CPU / GPU / memory request,
    GPU workers packed node by node, and the ordinary `Controller` then configures and supervises them through
    their control sockets. The reference's counterpart is ``realhf/system/controller.py:424-689``."""

    def __init__(self, experiment_name: str, trial_name: str, ray_module=None):
        self.experiment_name, self.trial_name = experiment_name, trial_name
        if ray_module is None:
            import ray as ray_module  # noqa: F811  (absent from images without Ray: the import error says so)
        self.ray = ray_module
        self.refs: Dict[str, Any] = {}
        self.base: Optional[Controller] = None

    def _launch(self, experiment: Experiment, env: Dict[str, str]):
        scheduling = experiment.scheduling_setup()
        setup = experiment.initial_setup()
        counts = [(wt, len(getattr(setup, wt)), getattr(scheduling, wt)) for wt in WORKER_FIELDS
                  if len(getattr(setup, wt)) > 0]
        for wt, idx, opts in plan_ray_placement(counts, self.ray.available_resources()):
            job = self.ray.remote(**opts)(run_ray_worker).remote(wt, idx, self.experiment_name, self.trial_name, env)
            self.refs[f"{wt}/{idx}"] = job
        try:  # one poll so that import / start-up errors of a task surface here and not as a connection timeout
            self.ray.get(list(self.refs.values()), timeout=1)
        except self.ray.exceptions.GetTimeoutError:
            pass
        logger.info(f"launched {len(self.refs)} workers as Ray tasks")

    def start(self, experiment: Experiment, ignore_worker_error: bool = False, env: Optional[Dict[str, str]] = None,
              timeout: Optional[float] = None):
        import os

        env = dict(env or {})
        for k in ("AREAL_PLATFORM", "AREAL_RECOVER_RUN", "HSA_ENABLE_IPC_MODE_LEGACY"):
            if k in os.environ:
                env.setdefault(k, os.environ[k])
        if not self.ray.is_initialized():
            self.ray.init(runtime_env={"env_vars": env, "working_dir": os.getcwd()})
        try:
            self._launch(experiment, env)
            self.base = Controller(self.experiment_name, self.trial_name)
            self.base.start(experiment, ignore_worker_error=ignore_worker_error, timeout=timeout)
            self.ray.get(list(self.refs.values()), timeout=60)  # the workers were told to exit
        finally:
            self.ray.shutdown()
