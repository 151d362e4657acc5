"""Process entry points of the worker runtime.

    python -m areal_vllm_amd.apps.remote worker -w model_worker -e EXP -f TRIAL -i 3
    python -m areal_vllm_amd.apps.remote controller -e EXP -f TRIAL --experiment /path/experiment.pkl

The experiment object travels as a pickle the launcher itself wrote (``apps/main.py``), so controller and
workers see exactly the configuration the user's command line produced. Role of ``realhf/apps/remote.py``.
"""
import argparse
import os
import pickle
import sys

from ..utils import logging

logger = logging.getLogger("remote")


def main_worker(args):
    from ..system.worker_base import run_worker

    if args.platform == "cpu":
        os.environ["AREAL_PLATFORM"] = "cpu"
        os.environ["HIP_VISIBLE_DEVICES"] = os.environ["CUDA_VISIBLE_DEVICES"] = ""
    logger.info(f"Run {args.worker_type} worker with index {args.index}")
    run_worker(args.worker_type, args.experiment_name, args.trial_name, f"{args.worker_type}/{args.index}")


def main_controller(args):
    from ..system.controller import Controller

    with open(args.experiment, "rb") as f:
        experiment = pickle.load(f)
    Controller(args.experiment_name, args.trial_name).start(experiment, ignore_worker_error=args.ignore_worker_error)


def main(argv=None):
    p = argparse.ArgumentParser(prog="areal_vllm_amd.apps.remote")
    sub = p.add_subparsers(dest="cmd", required=True)
    w = sub.add_parser("worker")
    w.add_argument("-w", "--worker_type", required=True)
    w.add_argument("-e", "--experiment_name", required=True)
    w.add_argument("-f", "--trial_name", required=True)
    w.add_argument("-i", "--index", type=int, required=True)
    w.add_argument("--platform", default=os.environ.get("AREAL_PLATFORM", ""))
    w.set_defaults(fn=main_worker)
    c = sub.add_parser("controller")
    c.add_argument("-e", "--experiment_name", required=True)
    c.add_argument("-f", "--trial_name", required=True)
    c.add_argument("--experiment", required=True)
    c.add_argument("--ignore_worker_error", action="store_true")
    c.set_defaults(fn=main_controller)
    args = p.parse_args(argv)
    args.fn(args)


if __name__ == "__main__":
    main()
