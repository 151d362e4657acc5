"""Worker configurations, experiment description and experiment registry of the worker runtime.

An `Experiment` produces an `ExperimentConfig`: the list of model function calls plus one configuration
object per worker (master, model workers, generation servers, their manager, rollout workers) and an
`ExperimentScheduling` (how many processes of each type, on which devices). `ExperimentConfig.lazy_init`
derives the routing tables every worker needs: model topologies, the (model shard -> model worker) map,
which pairs of models exchange data and which MFC publishes new weights. Contract of
``realhf/api/core/system_api.py:30-470`` (ExpStatus, worker dataclasses, ExperimentConfig, registry).
"""
import dataclasses
from typing import Any, Dict, List, Optional, Tuple

from ..api.alloc_mode import ParallelStrategy
from ..api.cli_args import InferenceServerConfig
from .agent_api import AgentAbstraction, EnvServiceAbstraction
from .dfg import MFCDef, ModelInterfaceType, ModelName, ParamReallocHook, build_graph
from .generation_server import GenerationServerConfig
from .gserver_manager import GserverManagerConfig
from .model_api import ModelAbstraction, ModelBackendAbstraction, ModelInterfaceAbstraction
from .worker_base import WorkerInformation


@dataclasses.dataclass
class DatasetAbstraction:
    type_: str = "prompt"
    args: Dict[str, Any] = dataclasses.field(default_factory=dict)


@dataclasses.dataclass(frozen=True)
class ModelShardID:
    model_name: ModelName
    dp_rank: int
    tp_rank: int = 0
    pp_rank: int = 0

    def __str__(self):
        return f"{self.model_name}/d{self.dp_rank}t{self.tp_rank}p{self.pp_rank}"


@dataclasses.dataclass
class StandaloneModelShard:
    id: ModelShardID
    model: ModelAbstraction
    backend: ModelBackendAbstraction
    eval_dataset: Optional[DatasetAbstraction] = None
    eval_bs: int = 128
    should_instantiate: bool = True


@dataclasses.dataclass
class ExperimentSaveEvalControl:
    total_train_epochs: int = 1
    save_freq_epochs: Optional[int] = None
    save_freq_steps: Optional[int] = None
    save_freq_secs: Optional[int] = None
    ckpt_freq_epochs: Optional[int] = None
    ckpt_freq_steps: Optional[int] = None
    ckpt_freq_secs: Optional[int] = None
    eval_freq_epochs: Optional[int] = None
    eval_freq_steps: Optional[int] = None
    eval_freq_secs: Optional[int] = None
    benchmark_steps: Optional[int] = None
    benchmark_n_seqs: Optional[int] = None


@dataclasses.dataclass
class ModelWorkerConfig:
    base_seed: int = 1
    shards: List[StandaloneModelShard] = dataclasses.field(default_factory=list)
    datasets: Optional[List[DatasetAbstraction]] = None
    tokenizer_name_or_path: Optional[str] = None
    shuffle_dataset: bool = True
    # "cpu" runs the worker on gloo / host tensors (tests, plumbing runs); "cuda" binds one MI355X
    device: str = "cuda"
    fileroot: str = "/tmp/areal"
    realloc_root: str = "/tmp/areal/param_realloc"
    # filled by ExperimentConfig.lazy_init
    model_rpcs: List[MFCDef] = dataclasses.field(default_factory=list)
    model_topos: Dict[ModelName, ParallelStrategy] = dataclasses.field(default_factory=dict)
    msid2mwid: Dict[ModelShardID, int] = dataclasses.field(default_factory=dict)
    n_model_workers: int = 1
    train_batch_size: int = 1
    total_train_epochs: int = 1
    profile_mode: bool = False
    worker_info: Optional[WorkerInformation] = None


@dataclasses.dataclass
class MasterWorkerConfig:
    exp_ctrl: ExperimentSaveEvalControl = dataclasses.field(default_factory=ExperimentSaveEvalControl)
    n_model_workers: int = 1
    buffer_size: int = 16384
    fileroot: str = "/tmp/areal"
    # the master polls this long for new data before it re-checks control messages
    fetch_timeout: float = 1.0
    model_rpcs: List[MFCDef] = dataclasses.field(default_factory=list)
    model_topos: Dict[ModelName, ParallelStrategy] = dataclasses.field(default_factory=dict)
    msid2mwid: Dict[ModelShardID, int] = dataclasses.field(default_factory=dict)
    worker_info: Optional[WorkerInformation] = None


@dataclasses.dataclass
class RolloutWorkerConfig:
    base_seed: int = 1
    model_name: str = "actor"
    tokenizer_path: Optional[str] = None
    new_tokens_per_chunk: int = 1 << 30
    rollout_request_timeout: float = 3600.0
    env: EnvServiceAbstraction = dataclasses.field(default_factory=EnvServiceAbstraction)
    agent: AgentAbstraction = dataclasses.field(default_factory=AgentAbstraction)
    datasets: List[DatasetAbstraction] = dataclasses.field(default_factory=list)
    stop_token_ids: List[int] = dataclasses.field(default_factory=list)
    fileroot: str = "/tmp/areal"
    worker_info: Optional[WorkerInformation] = None


@dataclasses.dataclass
class Scheduling:
    """Resources of one process: used by the local scheduler for device assignment and by the slurm
    scheduler for the resource request."""
    cpu: int = 1
    gpu: float = 0
    mem: int = 1024
    container_image: Optional[str] = None
    env_vars: Dict[str, str] = dataclasses.field(default_factory=dict)
    nodelist: Optional[str] = None
    exclude: Optional[str] = None
    time_limit: Optional[str] = None


@dataclasses.dataclass
class TasksGroup:
    count: int
    scheduling: Scheduling = dataclasses.field(default_factory=Scheduling)


@dataclasses.dataclass
class ExperimentScheduling:
    master_worker: TasksGroup = dataclasses.field(default_factory=lambda: TasksGroup(1))
    model_worker: TasksGroup = dataclasses.field(default_factory=lambda: TasksGroup(1))
    generation_server: Optional[TasksGroup] = None
    gserver_manager: Optional[TasksGroup] = None
    rollout_worker: Optional[TasksGroup] = None
    controller_image: Optional[str] = None


WORKER_FIELDS = ("master_worker", "model_worker", "generation_server", "gserver_manager", "rollout_worker")


@dataclasses.dataclass
class ExperimentConfig:
    exp_ctrl: ExperimentSaveEvalControl
    model_rpcs: List[MFCDef]
    model_worker: List[ModelWorkerConfig]
    generation_server: List[GenerationServerConfig] = dataclasses.field(default_factory=list)
    gserver_manager: List[GserverManagerConfig] = dataclasses.field(default_factory=list)
    rollout_worker: List[RolloutWorkerConfig] = dataclasses.field(default_factory=list)
    master_worker: List[MasterWorkerConfig] = dataclasses.field(default_factory=list)
    fileroot: str = "/tmp/areal"

    def __post_init__(self):
        self._validate_model_names()
        if not self.master_worker:
            self.master_worker = [MasterWorkerConfig(exp_ctrl=self.exp_ctrl, n_model_workers=len(self.model_worker),
                                                     fileroot=self.fileroot)]

    def _all_shards(self) -> List[Tuple[int, StandaloneModelShard]]:
        return [(i, s) for i, mw in enumerate(self.model_worker) for s in mw.shards]

    def _validate_model_names(self):
        names = {s.id.model_name for _, s in self._all_shards()}
        for r in self.model_rpcs:
            if r.model_name not in names:
                raise ValueError(f"MFC {r.name} uses model {r.model_name}, which no model worker hosts")

    def lazy_init(self):
        """Derive the routing tables and hand them to every worker configuration."""
        G = build_graph(self.model_rpcs)
        topos: Dict[ModelName, ParallelStrategy] = {}
        msid2mwid: Dict[ModelShardID, int] = {}
        extents: Dict[ModelName, List[int]] = {}
        for wid, s in self._all_shards():
            assert s.id not in msid2mwid, f"model shard {s.id} is hosted twice"
            msid2mwid[s.id] = wid
            e = extents.setdefault(s.id.model_name, [0, 0, 0])
            e[0], e[1], e[2] = max(e[0], s.id.dp_rank + 1), max(e[1], s.id.tp_rank + 1), max(e[2], s.id.pp_rank + 1)
        for name, (dp, tp, pp) in extents.items():
            topos[name] = ParallelStrategy(data_parallel_size=dp, tensor_parallel_size=tp, pipeline_parallel_size=pp)
            n = sum(1 for sid in msid2mwid if sid.model_name == name)
            assert n == dp * tp * pp, f"model {name}: {n} shards do not fill a d{dp}t{tp}p{pp} grid"
        # publishing weights: the last train step on a role that generation servers serve
        served = {g.model_name for g in self.gserver_manager}
        for r in self.model_rpcs:
            if r.interface_type == ModelInterfaceType.TRAIN_STEP and r.role in served and \
                    r.is_dst_of_model_role(r.role) and not any(isinstance(h, ParamReallocHook) for h in r._post_hooks):
                r.add_post_hook(ParamReallocHook(source=r.model_name, target=None))
        # replicas of a trained role that live in the worker world under another layout (a generation copy of a
        # tensor-parallel actor): the last train step of the role hands them its weights, shard to shard
        for r in self.model_rpcs:
            if r.interface_type == ModelInterfaceType.TRAIN_STEP and r.is_dst_of_model_role(r.role):
                for other in sorted((m for m in topos if m.role == r.role and m != r.model_name), key=str):
                    if not any(isinstance(h, ParamReallocHook) and h.target == other for h in r._post_hooks):
                        r.add_post_hook(ParamReallocHook(source=r.model_name, target=other))
        src = [r for r in self.model_rpcs if r.is_src]
        train_bs = src[0].n_seqs if src else 1
        for mw in self.model_worker:
            mw.model_rpcs, mw.model_topos, mw.msid2mwid = self.model_rpcs, topos, msid2mwid
            mw.n_model_workers = len(self.model_worker)
            mw.train_batch_size = train_bs
            mw.total_train_epochs = self.exp_ctrl.total_train_epochs
        for m in self.master_worker:
            m.model_rpcs, m.model_topos, m.msid2mwid = self.model_rpcs, topos, msid2mwid
            m.n_model_workers = len(self.model_worker)
        for g in self.gserver_manager:
            g.train_batch_size = train_bs
        self._graph = G
        return self

    def set_worker_information(self, experiment_name: str, trial_name: str):
        for wt in WORKER_FIELDS:
            cfgs = getattr(self, wt)
            for i, c in enumerate(cfgs):
                info = WorkerInformation(experiment_name=experiment_name, trial_name=trial_name, worker_type=wt,
                                         worker_index=i, worker_count=len(cfgs))
                if dataclasses.is_dataclass(c) and hasattr(c, "worker_info"):
                    c.worker_info = info
                else:
                    setattr(c, "worker_info", info)

    def get_worker_infos(self, experiment_name: str, trial_name: str) -> Dict[str, List[WorkerInformation]]:
        self.set_worker_information(experiment_name, trial_name)
        return {wt: [c.worker_info for c in getattr(self, wt)] for wt in WORKER_FIELDS}


class Experiment:
    """User-facing experiment: declares scheduling and builds the full configuration."""

    def scheduling_setup(self) -> ExperimentScheduling:
        raise NotImplementedError()

    def initial_setup(self) -> ExperimentConfig:
        raise NotImplementedError()


ALL_EXPERIMENT_CLASSES: Dict[str, Any] = {}


def register_experiment(name: str, cls):
    assert name not in ALL_EXPERIMENT_CLASSES, name
    ALL_EXPERIMENT_CLASSES[name] = cls


def make_experiment(name: str) -> Experiment:
    from .. import experiments as _e  # noqa: F401  (registers the built-in experiments)

    cls = ALL_EXPERIMENT_CLASSES[name]
    return cls() if isinstance(cls, type) else cls()
