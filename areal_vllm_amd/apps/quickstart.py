"""Command-line entry of the worker runtime:

    python -m areal_vllm_amd.apps.quickstart async-ppo-math --config exp.yaml actor.path=/ckpt/qwen2 \
        layout=d4 n_generation_servers=4 max_head_offpolicyness=2 mode=local recover_mode=auto

The first argument names a registered experiment; its dataclass fields are filled from the YAML file and
``key=value`` overrides (same override syntax and validation as the SPMD launchers). ``mode``,
``recover_mode``, ``recover_retries`` and ``platform`` address the launcher itself. Role of
``realhf/apps/quickstart.py`` + ``realhf/api/quickstart/*``.
"""
import argparse
import dataclasses
import sys
from typing import List

import yaml

from ..api.cli_args import apply_overrides, resolve_interpolations, to_structured_cfg
from ..system import system_api
from ..utils import logging
from .main import main_start

logger = logging.getLogger("quickstart")
LAUNCHER_KEYS = dict(mode="local", recover_mode="disabled", recover_retries=1, platform="")


def build_experiment(name: str, config_file: str = None, overrides: List[str] = ()):
    from .. import experiments as _e  # noqa: F401

    if name not in system_api.ALL_EXPERIMENT_CLASSES:
        raise KeyError(f"unknown experiment '{name}'; registered: {sorted(system_api.ALL_EXPERIMENT_CLASSES)}")
    cls = system_api.ALL_EXPERIMENT_CLASSES[name]
    cfg = {}
    if config_file:
        with open(config_file) as f:
            cfg = yaml.safe_load(f) or {}
    cfg = apply_overrides(cfg, list(overrides))
    launcher = {k: cfg.pop(k, d) for k, d in LAUNCHER_KEYS.items()}
    try:
        defaults = dataclasses.asdict(cls())
    except TypeError:
        defaults = {}
    cfg = resolve_interpolations(cfg, defaults)
    return to_structured_cfg(cfg, cls), launcher


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    p = argparse.ArgumentParser(prog="areal_vllm_amd.apps.quickstart")
    p.add_argument("experiment")
    p.add_argument("--config", default=None)
    p.add_argument("--show", action="store_true", help="print the resolved configuration and exit")
    args, overrides = p.parse_known_args(argv)
    exp, launcher = build_experiment(args.experiment, args.config, overrides)
    if args.show:
        print(yaml.safe_dump(dataclasses.asdict(exp), sort_keys=False))
        return exp
    main_start(exp, exp.experiment_name, exp.trial_name, mode=launcher["mode"], fileroot=exp.fileroot,
               recover_mode=launcher["recover_mode"], recover_retries=int(launcher["recover_retries"]),
               platform=launcher["platform"] or ("cpu" if exp.device == "cpu" else ""),
               recover_run=launcher["recover_mode"] == "resume")
    return exp


if __name__ == "__main__":
    main()
