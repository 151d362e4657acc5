"""Process-wide context of a model worker: which model (and therefore which parallel grid) is "current".

A worker may host shards of several models (actor, critic, reference) with different topologies; code that
needs "the" data-parallel group asks this module, and the worker switches the context with
``with model_scope(name):`` around each model function call. Role of ``realhf/base/constants.py``
(model_scope, grid accessors, path helpers); environment knobs keep the ``REAL_*`` / ``AREAL_*`` names.
"""
import contextlib
import getpass
import os
from typing import Dict, Optional

_experiment_name: Optional[str] = None
_trial_name: Optional[str] = None
_model_name: Optional[str] = None
_grids: Dict[str, object] = {}
_rank_mapping: Dict[str, Dict[int, int]] = {}

QUICKSTART_EXPR_CACHE_PATH = os.getenv("AREAL_EXPR_CACHE", "/tmp/areal/quickstart_cache")
DATASET_CACHE_PATH = os.getenv("AREAL_DATASET_CACHE", "/tmp/areal/datasets_cache")
SAVE_MAX_SHARD_SIZE_BYTE = int(os.getenv("REAL_SAVE_MAX_SHARD_SIZE_BYTE", str(int(1e10))))
PPO_DATA_RECORD_N_SEQS = int(os.getenv("REAL_RECORD_PERFORMANCE", "0"))


def set_experiment_trial_names(experiment_name: str, trial_name: str):
    global _experiment_name, _trial_name
    _experiment_name, _trial_name = experiment_name, trial_name


def experiment_name() -> str:
    assert _experiment_name is not None, "experiment name not set"
    return _experiment_name


def trial_name() -> str:
    assert _trial_name is not None, "trial name not set"
    return _trial_name


def set_grid(model_name: str, grid) -> None:
    _grids[str(model_name)] = grid


def has_model_name(model_name: str) -> bool:
    return str(model_name) in _grids


def set_rank_mapping(model_name: str, mapping: Dict[int, int]):
    _rank_mapping[str(model_name)] = mapping


def rank_mapping_of_model(model_name: str) -> Dict[int, int]:
    return _rank_mapping[str(model_name)]


@contextlib.contextmanager
def model_scope(model_name: str):
    global _model_name
    assert str(model_name) in _grids, f"model {model_name} has no grid on this worker"
    prev, _model_name = _model_name, str(model_name)
    try:
        yield
    finally:
        _model_name = prev


@contextlib.contextmanager
def model_scope_disabled():
    global _model_name
    prev, _model_name = _model_name, None
    try:
        yield
    finally:
        _model_name = prev


def model_name() -> str:
    assert _model_name is not None, "not inside a model_scope"
    return _model_name


def grid():
    return _grids[model_name()]


def data_parallel_group():
    return grid().get_data_parallel_group()


def tensor_parallel_group():
    return grid().get_tensor_model_parallel_group()


def pipe_parallel_group():
    return grid().get_pipe_parallel_group()


def data_parallel_rank() -> int:
    return grid().coord().data


def tensor_parallel_rank() -> int:
    return grid().coord().tensor


def pipe_parallel_rank() -> int:
    return grid().coord().pipe


def data_parallel_world_size() -> int:
    return grid().topo.get_dim("data")


def tensor_parallel_world_size() -> int:
    return grid().topo.get_dim("tensor")


def pipe_parallel_world_size() -> int:
    return grid().topo.get_dim("pipe")


def log_root(fileroot: str) -> str:
    return os.path.join(fileroot, "logs", getpass.getuser())


def model_save_root(fileroot: str) -> str:
    return os.path.join(fileroot, "checkpoints", getpass.getuser())


def get_log_path(fileroot: str, experiment: Optional[str] = None, trial: Optional[str] = None) -> str:
    p = os.path.join(log_root(fileroot), experiment or experiment_name(), trial or trial_name())
    os.makedirs(p, exist_ok=True)
    return p


def get_save_path(fileroot: str, experiment: Optional[str] = None, trial: Optional[str] = None) -> str:
    p = os.path.join(model_save_root(fileroot), experiment or experiment_name(), trial or trial_name())
    os.makedirs(p, exist_ok=True)
    return p
