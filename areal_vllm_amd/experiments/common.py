"""Pieces shared by the experiment builders: model / optimizer descriptions, the symmetric placement of
models on model workers and the parsing of a train layout string. Role of
``realhf/experiments/common/{common,utils}.py`` and the model blocks of ``realhf/api/cli_args.py``."""
import dataclasses
import re
from typing import Any, Dict, List, Optional, Tuple

from ..api.alloc_mode import ParallelStrategy
from ..api.cli_args import MicroBatchSpec, OptimizerConfig
from ..system.dfg import ModelName
from ..system.model_api import ModelAbstraction, ModelBackendAbstraction
from ..system.system_api import DatasetAbstraction, ModelShardID, ModelWorkerConfig, StandaloneModelShard


@dataclasses.dataclass
class ModelTrainEvalConfig:
    path: str = "tiny"
    init_from_scratch: bool = False
    is_critic: bool = False
    init_critic_from_actor: bool = False
    dtype: str = "bfloat16"
    gradient_checkpointing: bool = False
    backend: str = "native_train"  # native_train | inference | native_generation (vllm, sglang) | mock_train | null
    optimizer: OptimizerConfig = dataclasses.field(default_factory=OptimizerConfig)
    tokenizer_path: Optional[str] = None
    # park the model's weights and optimizer state in pinned host memory whenever no function call runs on it
    # (off by default: 288 GB per GPU hold an actor, a reference copy and a critic side by side)
    offload: bool = False


def attach_offload_hooks(rpcs, models: Dict[str, "ModelTrainEvalConfig"]) -> None:
    """Every model function call on a role configured with `offload` gets the offload post-hook."""
    from ..system.dfg import OffloadHook

    for rpc in rpcs:
        m = models.get(rpc.model_name.role)
        if m is not None and m.offload and rpc.model_name.replica_id == 0:
            rpc.add_post_hook(OffloadHook())


def parse_layout(s: str) -> ParallelStrategy:
    """"d4t2p1" (any order, missing letters = 1) -> ParallelStrategy."""
    dims = dict(d=1, t=1, p=1)
    for k, v in re.findall(r"([dtp])(\d+)", s):
        dims[k] = int(v)
    return ParallelStrategy(data_parallel_size=dims["d"], tensor_parallel_size=dims["t"],
                            pipeline_parallel_size=dims["p"])


def resolve_allocation_mode(exp) -> None:
    """`allocation_mode` (the reference's spelling of the device split, e.g. ``sglang.d4p1m1+d2p2m1`` or
    ``d4p2m1``) overrides `layout` and, for experiments with generation servers, their count and
    tensor-parallel size."""
    mode = getattr(exp, "allocation_mode", None)
    if not mode:
        return
    from ..api.alloc_mode import AllocationMode

    a = AllocationMode.from_str(mode)
    t = a.train if a.train is not None else a.gen
    exp.layout = f"d{t.dp_size}t{t.tp_size}p{t.pp_size}"
    if a.gen is not None and a.gen_backend is not None and hasattr(exp, "n_generation_servers"):
        exp.n_generation_servers = a.gen.dp_size
        exp.gen_tp_size = a.gen.tp_size


def model_abstraction(m: ModelTrainEvalConfig, experiment_name: str, trial_name: str) -> ModelAbstraction:
    if m.backend == "null":
        return ModelAbstraction("null")
    return ModelAbstraction("native", args=dict(
        path=m.path, is_critic=m.is_critic, init_from_scratch=m.init_from_scratch,
        init_critic_from_actor=m.init_critic_from_actor, dtype=m.dtype,
        gradient_checkpointing=m.gradient_checkpointing, tokenizer_path=m.tokenizer_path,
        experiment_name=experiment_name, trial_name=trial_name))


def backend_abstraction(m: ModelTrainEvalConfig) -> ModelBackendAbstraction:
    if m.backend in ("native_train", "mock_train", "megatron"):
        return ModelBackendAbstraction(m.backend, args=dict(optimizer=dataclasses.asdict(m.optimizer)))
    return ModelBackendAbstraction(m.backend)


def symmetric_model_workers(models: Dict[str, ModelTrainEvalConfig], layout: ParallelStrategy, experiment_name: str,
                            trial_name: str, device: str, fileroot: str, realloc_root: str, base_seed: int,
                            datasets: Optional[List[DatasetAbstraction]], tokenizer_path: Optional[str],
                            eval_dataset: Optional[DatasetAbstraction] = None, eval_role: Optional[str] = None,
                            shuffle_dataset: bool = True,
                            replicas: Optional[Dict[ModelName, Tuple[ModelTrainEvalConfig, ParallelStrategy]]] = None,
                            ) -> List[ModelWorkerConfig]:
    """Every model uses the same (dp, tp, pp) grid over all model workers: worker w hosts shard w of each
    model (the reference calls this the symmetric allocation; with 288 GB per GPU an actor, a reference
    copy and a critic of up to ~30B parameters each fit side by side, so nothing is offloaded)."""
    n = layout.world_size
    workers = []
    for w in range(n):
        tp = w % layout.tp_size
        dp = (w // layout.tp_size) % layout.dp_size
        pp = w // (layout.tp_size * layout.dp_size)
        shards = []
        for role, m in models.items():
            sid = ModelShardID(ModelName(role, 0), dp_rank=dp, tp_rank=tp, pp_rank=pp)
            shards.append(StandaloneModelShard(
                id=sid, model=model_abstraction(m, experiment_name, trial_name), backend=backend_abstraction(m),
                eval_dataset=eval_dataset if role == eval_role else None))
        # further replicas of a role with a layout of their own over the same workers (e.g. a dp-only copy of a
        # tensor-parallel actor that generates): worker w hosts shard w of that layout as well
        for rname, (m, lay) in (replicas or {}).items():
            assert lay.world_size == n, f"replica {rname} must cover all {n} model workers"
            sid = ModelShardID(rname, dp_rank=(w // lay.tp_size) % lay.dp_size, tp_rank=w % lay.tp_size,
                               pp_rank=w // (lay.tp_size * lay.dp_size))
            shards.append(StandaloneModelShard(id=sid, model=model_abstraction(m, experiment_name, trial_name),
                                               backend=backend_abstraction(m)))
        workers.append(ModelWorkerConfig(base_seed=base_seed, shards=shards, datasets=datasets,
                                         tokenizer_name_or_path=tokenizer_path, device=device, fileroot=fileroot,
                                         realloc_root=realloc_root, shuffle_dataset=shuffle_dataset))
    return workers
