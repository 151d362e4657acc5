"""Recover information of the worker runtime: where to resume and which samples of the interrupted
epoch were already consumed.

Written by the master after every checkpoint, read by the master, the model workers (dataset position)
and the rollout workers (samples to skip) of the relaunched experiment. Plain JSON on the shared file
system (the reference pickles the same fields, ``realhf/base/recover.py``).
"""
import dataclasses
import json
import os
from typing import Dict, List, Tuple

from ..api.io_struct import StepInfo

RECOVER_FILE = "recover_info.json"


@dataclasses.dataclass
class RecoverInfo:
    recover_start: StepInfo
    last_step_info: StepInfo
    save_ctl_info: Dict = dataclasses.field(default_factory=dict)
    ckpt_ctl_info: Dict = dataclasses.field(default_factory=dict)
    eval_ctl_info: Dict = dataclasses.field(default_factory=dict)
    hash_vals_to_ignore: List[str] = dataclasses.field(default_factory=list)


def dump_recover_info(recover_dir: str, info: RecoverInfo):
    os.makedirs(recover_dir, exist_ok=True)
    tmp = os.path.join(recover_dir, RECOVER_FILE + ".tmp")
    with open(tmp, "w") as f:
        json.dump(dataclasses.asdict(info), f)
    os.replace(tmp, os.path.join(recover_dir, RECOVER_FILE))


def load_recover_info(recover_dir: str) -> Tuple[bool, RecoverInfo]:
    """-> (is this a recover run, info). A run recovers when ``AREAL_RECOVER_RUN=1`` and a dump exists."""
    p = os.path.join(recover_dir, RECOVER_FILE)
    if os.environ.get("AREAL_RECOVER_RUN", "0") != "1" or not os.path.exists(p):
        zero = StepInfo(epoch=0, epoch_step=0, global_step=0, steps_per_epoch=1)
        return False, RecoverInfo(recover_start=zero, last_step_info=zero)
    with open(p) as f:
        d = json.load(f)
    d["recover_start"] = StepInfo(**d["recover_start"])
    d["last_step_info"] = StepInfo(**d["last_step_info"])
    return True, RecoverInfo(**d)
