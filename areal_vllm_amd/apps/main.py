"""Launch an experiment of the worker runtime and keep it alive.

``main_start`` builds the experiment, clears stale name_resolve entries, submits the controller and one
process per worker through a scheduler client, waits for the controller, and - when recovery is enabled -
relaunches the whole experiment with ``AREAL_RECOVER_RUN=1`` after a failure, up to `recover_retries`
times. ``main_stop`` cancels the jobs of a trial. Behaviour of ``realhf/apps/main.py`` (main_start's
recover loop, 426 lines).
"""
import argparse
import getpass
import os
import pickle
import time
from typing import Optional

from .. import scheduler as sched
from ..scheduler.client import control_cmd, remote_worker_cmd
from ..system.system_api import WORKER_FIELDS, Experiment, make_experiment
from ..utils import logging, name_resolve, names

logger = logging.getLogger("main")
RECOVER_TIME_INTERVAL = 10.0


def _submit_workers(client, experiment: Experiment, expr_name: str, trial_name: str, env: dict, platform: str):
    scheduling = experiment.scheduling_setup()
    for wt in WORKER_FIELDS:
        tg = getattr(scheduling, wt, None)
        if tg is None or tg.count == 0:
            continue
        s = tg.scheduling
        cmd = remote_worker_cmd(expr_name, trial_name, wt) + (f" --platform {platform}" if platform else "")
        gpu = int(s.gpu) if platform != "cpu" else 0
        client.submit_array(wt, cmd, count=tg.count, gpu=gpu, cpu=s.cpu, mem=s.mem,
                            env_vars=dict(env, **s.env_vars), nodelist=s.nodelist, exclude=s.exclude,
                            time_limit=s.time_limit)


def main_start(experiment: Experiment, expr_name: str, trial_name: str, mode: str = "local",
               fileroot: str = "/tmp/areal", recover_mode: str = "disabled", recover_retries: int = 1,
               platform: str = "", ignore_worker_error: bool = False, timeout: Optional[float] = None,
               recover_run: bool = False):
    """Returns when the experiment has completed; raises the last failure otherwise."""
    log_dir = os.path.join(fileroot, "logs", getpass.getuser(), expr_name, trial_name)
    os.makedirs(log_dir, exist_ok=True)
    exp_pkl = os.path.join(log_dir, "experiment.pkl")
    with open(exp_pkl, "wb") as f:
        pickle.dump(experiment, f)
    if mode == "ray":
        # the controller launches the workers itself, as Ray tasks (no scheduler client, no controller process)
        from ..system.controller import RayController

        name_resolve.clear_subtree(names.trial_root(expr_name, trial_name))
        RayController(expr_name, trial_name).start(
            experiment, ignore_worker_error=ignore_worker_error, timeout=timeout,
            env=dict(AREAL_RECOVER_RUN="1" if recover_run else "0", AREAL_PLATFORM=platform))
        return
    attempt = 0
    while True:
        name_resolve.clear_subtree(names.trial_root(expr_name, trial_name))
        env = dict(AREAL_RECOVER_RUN="1" if (recover_run or attempt > 0) else "0", AREAL_PLATFORM=platform,
                   PYTHONPATH=os.pathsep.join([os.path.dirname(os.path.dirname(os.path.dirname(
                       os.path.abspath(__file__))))] + os.environ.get("PYTHONPATH", "").split(os.pathsep)))
        client = sched.make(mode, expr_name, trial_name, fileroot=fileroot)
        try:
            client.submit("controller", control_cmd(expr_name, trial_name, exp_pkl, ignore_worker_error), env_vars=env)
            _submit_workers(client, experiment, expr_name, trial_name, env, platform)
            client.wait(timeout=timeout, check_status=(sched.JobState.CANCELLED, sched.JobState.FAILED,
                                                      sched.JobState.NOT_FOUND),
                        remove_status=(sched.JobState.COMPLETED,))
            logger.info(f"experiment {expr_name}/{trial_name} finished")
            return
        except (sched.JobException, TimeoutError) as e:
            client.stop_all()
            recoverable = isinstance(e, sched.JobException) and recover_mode in ("auto", "fault")
            if not recoverable or attempt >= recover_retries:
                raise
            attempt += 1
            logger.warning(f"experiment failed ({e}); recovering, attempt {attempt}/{recover_retries}")
            time.sleep(RECOVER_TIME_INTERVAL)
        except KeyboardInterrupt:
            client.stop_all()
            raise
        finally:
            try:
                client.stop_all()
            except Exception:
                pass


def main_stop(expr_name: str, trial_name: str, mode: str = "local", fileroot: str = "/tmp/areal"):
    sched.make(mode, expr_name, trial_name, fileroot=fileroot).stop_all()


def main(argv=None):
    p = argparse.ArgumentParser(prog="areal_vllm_amd.apps.main")
    sub = p.add_subparsers(dest="cmd", required=True)
    s = sub.add_parser("start")
    s.add_argument("-e", "--experiment_name", required=True, help="a registered experiment name")
    s.add_argument("-f", "--trial_name", default="trial")
    s.add_argument("--mode", default="local", choices=["local", "slurm", "ray"])
    s.add_argument("--fileroot", default="/tmp/areal")
    s.add_argument("--recover_mode", default="disabled", choices=["disabled", "auto", "fault", "resume"])
    s.add_argument("--recover_retries", type=int, default=1)
    s.add_argument("--platform", default="")
    k = sub.add_parser("stop")
    k.add_argument("-e", "--experiment_name", required=True)
    k.add_argument("-f", "--trial_name", default="trial")
    k.add_argument("--mode", default="local")
    args = p.parse_args(argv)
    if args.cmd == "start":
        main_start(make_experiment(args.experiment_name), args.experiment_name, args.trial_name, args.mode,
                   args.fileroot, args.recover_mode, args.recover_retries, args.platform,
                   recover_run=args.recover_mode == "resume")
    else:
        main_stop(args.experiment_name, args.trial_name, args.mode)


if __name__ == "__main__":
    main()
