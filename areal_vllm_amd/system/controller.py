"""Experiment controller: configures the worker processes, starts them and supervises the run.

The scheduler has already started one process per worker; each registered its control address in
name_resolve. The controller builds the experiment's configuration, connects to every worker, sends
``configure`` (worker information + its configuration object), then ``start``, and afterwards watches the
worker statuses: the run ends when the master worker completes; a worker in ERROR / LOST state fails the
run (the launcher's recover loop may then relaunch it); on the way out every worker is told to exit.
Behaviour of ``realhf/system/controller.py:98-422``.
"""
import time
from typing import Dict, List, Optional

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
