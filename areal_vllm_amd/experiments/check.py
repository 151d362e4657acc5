"""Sanity checks of an experiment description before anything is launched: model paths, optimizer
settings, batch sizes against the data-parallel degree, rollout-server memory. Role of
``realhf/experiments/common/check.py``; raising early beats failing after minutes of start-up."""
import os
from typing import Dict

from ..api.alloc_mode import ParallelStrategy
from ..models.config import PRESETS
from .common import ModelTrainEvalConfig


def check_valid_optimizer(role: str, model: ModelTrainEvalConfig):
    o = model.optimizer
    if model.backend in ("inference", "null"):
        return
    if o.lr <= 0:
        raise ValueError(f"{role}: learning rate must be positive, got {o.lr}")
    if not (0.0 <= o.warmup_steps_proportion < 1.0):
        raise ValueError(f"{role}: warmup_steps_proportion must be in [0, 1), got {o.warmup_steps_proportion}")
    if o.lr_scheduler_type not in ("constant", "linear", "cosine"):
        raise ValueError(f"{role}: unknown lr scheduler {o.lr_scheduler_type}")
    if not (0.0 <= o.min_lr_ratio <= 1.0):
        raise ValueError(f"{role}: min_lr_ratio must be in [0, 1], got {o.min_lr_ratio}")


def check_valid_model_and_path(role: str, model: ModelTrainEvalConfig):
    if model.backend == "null":
        return
    p = model.path
    if os.path.isdir(p):
        if not os.path.exists(os.path.join(p, "config.json")):
            raise FileNotFoundError(f"{role}: {p} has no config.json (not a HF checkpoint directory)")
        if not model.init_from_scratch and not any(f.endswith(".safetensors") for f in os.listdir(p)):
            raise FileNotFoundError(f"{role}: {p} has no *.safetensors weights and init_from_scratch is off")
        return
    if p.lower() in PRESETS:
        if not model.init_from_scratch:
            raise ValueError(f"{role}: '{p}' is an architecture preset without weights; set init_from_scratch")
        return
    raise FileNotFoundError(f"{role}: model path '{p}' is neither a checkpoint directory nor a known preset")


def check_valid_parallel_batch_size(name: str, n_seqs: int, layout: ParallelStrategy, pp_min_mbs: int = 2):
    """Every data-parallel rank needs at least one sequence, pipeline stages at least `pp_min_mbs * pp`
    micro-batches' worth (the 1F1B schedule runs 2*pp micro-batches by default)."""
    need = layout.dp_size * (pp_min_mbs * layout.pp_size if layout.pp_size > 1 else 1)
    if n_seqs < need:
        raise ValueError(f"{name}: batch of {n_seqs} sequences is too small for layout {layout} (needs >= {need})")


def check_valid_rollout_server(mem_fraction_static: float, colocated_with_trainer: bool):
    if not (0.0 < mem_fraction_static <= 1.0):
        raise ValueError(f"mem_fraction_static must be in (0, 1], got {mem_fraction_static}")
    if colocated_with_trainer and mem_fraction_static > 0.8:
        raise ValueError("a rollout engine that shares its GPU with a trainer must leave room for it: "
                         "mem_fraction_static <= 0.8 (or give the KV cache an absolute size)")


def check_experiment(models: Dict[str, ModelTrainEvalConfig], layout: ParallelStrategy, train_bs_n_seqs: int):
    for role, m in models.items():
        check_valid_model_and_path(role, m)
        check_valid_optimizer(role, m)
    check_valid_parallel_batch_size("train batch", train_bs_n_seqs, layout)
