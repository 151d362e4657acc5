"""Experiment configuration: dataclass tree + YAML loader + ``key=value`` overrides.

The field names, defaults and YAML layout follow the reference's ``areal/api/cli_args.py`` so its
example YAMLs load unchanged (``experiment_name``, ``allocation_mode``, ``rollout.*``, ``gconfig.*``,
``actor.*``, ``sglang.*`` ...). Hydra/OmegaConf are not dependencies: ``load_expr_config`` reads the
YAML, applies dotted CLI overrides, resolves ``${a.b}`` interpolations and merges the result into the
dataclass defaults with type checking (``to_structured_cfg``).

The inference-server section keeps the reference's name ``sglang`` in YAML for compatibility but
configures the NATIVE engine (``areal_vllm_amd.inference``); ``InferenceServerConfig.build_cmd``
produces the native server command line.
"""
import argparse
import dataclasses
import getpass
import os
import re
import types
import typing
from dataclasses import asdict, dataclass, field
from pathlib import Path
from typing import Any, Dict, List, Optional, Union

import yaml

MISSING = "???"


def get_user_tmp() -> str:
    user = getpass.getuser() if hasattr(getpass, "getuser") else "user"
    p = os.path.join("/tmp", "areal", user)
    return p


@dataclass
class NormConfig:
    """Base class of normalisation configs."""


@dataclass
class AdvNormConfig(NormConfig):
    mean_level: str = field(default="batch", metadata={"help": "batch | group | none"})
    std_level: str = field(default="batch", metadata={"help": "batch | group | none"})
    group_size: int = field(default=1, metadata={"help": "group size for group-level statistics"})


@dataclass
class RewardNormConfig(NormConfig):
    pass


@dataclass
class MicroBatchSpec:
    """n_mbs = (minimum) number of micro-batches; max_tokens_per_mb = token budget per micro-batch."""

    n_mbs: Optional[int] = 1
    max_tokens_per_mb: Optional[int] = None

    @classmethod
    def new(cls, mb_spec: "MicroBatchSpec", **kwargs):
        f = dict(n_mbs=mb_spec.n_mbs, max_tokens_per_mb=mb_spec.max_tokens_per_mb)
        f.update(kwargs)
        return cls(**f)


@dataclass
class GenerationHyperparameters:
    n_samples: int = 1
    max_new_tokens: int = 16384
    min_new_tokens: int = 0
    greedy: bool = False
    top_p: float = 1.0
    top_k: int = int(1e8)
    temperature: float = 1.0
    stop_token_ids: List[int] = field(default_factory=list)
    stop: Optional[List[str]] = None
    frequency_penalty: float = 0.0

    def new(self, **kwargs):
        args = asdict(self)
        args.update(kwargs)
        return GenerationHyperparameters(**args)


@dataclass
class OptimizerConfig:
    type: str = field(default="adam", metadata={"choices": ["adam", "empty"]})
    lr: float = 2e-5
    weight_decay: float = 0.05
    beta1: float = 0.9
    beta2: float = 0.95
    eps: float = 1e-5
    min_lr_ratio: float = 0.0
    lr_scheduler_type: str = field(default="constant", metadata={"choices": ["linear", "cosine", "constant"]})
    warmup_steps_proportion: float = 0.001
    offload: bool = False
    initial_loss_scale: float = 2**32
    min_loss_scale: float = 1.0
    loss_scale_window: float = 5
    hysteresis: int = 2
    gradient_clipping: float = 1.0


@dataclass
class FSDPWrapPolicy:
    transformer_layer_cls_to_wrap: Optional[List[str]] = None


@dataclass
class FSDPEngineConfig:
    wrap_policy: Optional[FSDPWrapPolicy] = None
    offload_params: bool = False


@dataclass
class DeepSpeedAutoTPEngineConfig:
    autotp_size: Optional[int] = 1


@dataclass
class NativeEngineConfig:
    """Knobs of the native ZeRO/TP/PP trainer (no counterpart in the reference)."""

    keep_logits: bool = field(default=True, metadata={"help": "keep bf16 logits for backward instead of "
                                                      "recomputing the LM head"})
    logprob_chunk: int = 8192
    sequence_parallel: bool = False
    overlap_grad_reduce: bool = field(default=True, metadata={
        "help": "ZeRO-1: reduce gradient buckets to their owners during the last backward pass of a step instead "
                "of one reduce-scatter after it"})
    zero_stage: int = field(default=1, metadata={
        "help": "1: bf16 weights and fp32 gradients replicated, optimizer state sharded over dp (two large "
                "collectives per step; the default with 288 GB per GPU). 3: weights, gradients and optimizer state "
                "all sharded, decoder blocks gathered on demand (FSDP semantics; data / context parallel layouts)",
        "choices": [1, 3]})
    reshard_after_forward: bool = field(default=True, metadata={
        "help": "ZeRO-3: drop a block's gathered weights after its forward and gather them again for the backward"})
    offload_optimizer: bool = field(default=False, metadata={
        "help": "ZeRO-3: keep fp32 master weights and Adam moments in pinned host memory"})
    peer_collectives: bool = field(default=False, metadata={
        "help": "groups of 2..8 ranks on one node run the Ulysses exchange, the sequence-parallel all-gather / "
                "reduce-scatter and the ZeRO-1 bucket collectives as kernels over peer-mapped memory instead of RCCL "
                "(also AREAL_PEER_COLLECTIVES=1); a group that fails the start-up check stays on RCCL"})
    context_parallel_mode: str = field(default="ulysses", metadata={
        "help": "what the `c` dimension of the allocation mode does: 'ulysses' (all-to-all between token and "
                "head partitions, the reference's FSDP engine) or 'zigzag' (load-balanced token placement with "
                "all-gathered K/V, the role of Megatron's context parallelism)",
        "choices": ["ulysses", "zigzag"]})


@dataclass
class TrainEngineConfig:
    experiment_name: str = MISSING
    trial_name: str = MISSING
    path: str = field(default="", metadata={"help": "HF checkpoint dir, or a preset name with init_from_scratch"})
    attn_impl: str = "flash_attention_2"
    init_from_scratch: bool = False
    init_critic_from_actor: bool = False
    mb_spec: MicroBatchSpec = field(default_factory=MicroBatchSpec)
    pad_to_maximum: bool = False
    disable_dropout: bool = False
    gradient_checkpointing: bool = True
    dtype: str = "bfloat16"
    grad_reduce_dtype: str = "float32"
    optimizer: Optional[OptimizerConfig] = None
    backend: str = ""
    fsdp: FSDPEngineConfig = field(default_factory=FSDPEngineConfig)
    ds_auto_tp: DeepSpeedAutoTPEngineConfig = field(default_factory=DeepSpeedAutoTPEngineConfig)
    native: NativeEngineConfig = field(default_factory=NativeEngineConfig)


@dataclass
class PPOActorConfig(TrainEngineConfig):
    group_size: int = 1
    ppo_n_minibatches: int = 4
    eps_clip: float = 0.2
    eps_clip_higher: Optional[float] = None
    c_clip: Optional[float] = None
    temperature: float = 1.0
    group_reward_norm: bool = False
    reward_scaling: float = 1.0
    reward_bias: float = 0.0
    reward_clip: float = 20.0
    overlong_reward_penalty: bool = False
    overlong_tokens: Optional[int] = None
    overlong_penalty_factor: Optional[float] = None
    mask_no_eos_with_zero: bool = False
    discount: float = 1.0
    gae_lambda: float = 1.0
    kl_ctl: float = 0.1
    recompute_logprob: bool = False
    fuse_proximal_logp: bool = field(default=False, metadata={
        "help": "decoupled loss with ppo_n_minibatches == 1 only: take the proximal log-probs from the training "
                "forward pass itself (same weights, so the same values as a separate recompute pass) instead of "
                "running an extra forward over the batch"})
    use_decoupled_loss: bool = False
    behav_imp_weight_cap: Optional[float] = None
    dynamic_sampling: bool = False
    log_agent_stats: bool = False
    log_agent_stats_keys: List[str] = field(default_factory=list)
    max_new_tokens: int = 1024
    adv_norm: Optional[AdvNormConfig] = None


@dataclass
class PPOCriticConfig(TrainEngineConfig):
    ppo_n_minibatches: int = 4
    value_eps_clip: float = 0.2
    value_norm: bool = False
    value_norm_type: str = "exp"
    value_norm_beta: float = 0.99995
    value_norm_eps: float = 1e-5
    value_loss_type: str = "mse"
    mask_no_eos_with_zero: bool = False


@dataclass
class InferenceServerConfig:
    """Native rollout-engine server options. The names match the reference's ``SGLangConfig``
    (``areal/api/cli_args.py:398-454``); options that only select third-party kernels are accepted and
    ignored (this engine has one HIP implementation)."""

    model_path: str = ""
    random_seed: int = 1
    skip_tokenizer_init: bool = False
    disable_cuda_graph: bool = False
    disable_radix_cache: bool = False
    disable_cuda_graph_padding: bool = False
    enable_nccl_nvls: bool = False
    disable_outlines_disk_cache: bool = False
    disable_custom_all_reduce: bool = False
    disable_overlap_schedule: bool = False
    enable_mixed_chunk: bool = False
    enable_dp_attention: bool = False
    enable_ep_moe: bool = False
    enable_torch_compile: bool = False
    torch_compile_max_bs: int = 32
    cuda_graph_max_bs: Optional[int] = None
    cuda_graph_bs: Optional[List[int]] = None
    torchao_config: str = ""
    enable_nan_detection: bool = False
    enable_p2p_check: bool = False
    triton_attention_reduce_in_fp32: bool = False
    triton_attention_num_kv_splits: int = 8
    num_continuous_decode_steps: int = 1
    enable_memory_saver: bool = False
    allow_auto_truncate: bool = False
    attention_backend: Optional[str] = "hip"
    sampling_backend: Optional[str] = None
    context_length: Optional[int] = 32768
    mem_fraction_static: Optional[float] = 0.9
    max_running_requests: Optional[int] = None
    chunked_prefill_size: Optional[int] = -1
    max_prefill_tokens: int = 32768
    schedule_policy: str = "lpm"
    schedule_conservativeness: float = 1.0
    cpu_offload_gb: int = 0
    dtype: str = "bfloat16"
    kv_cache_dtype: str = "auto"
    dp_size: int = 1
    ep_size: int = 1
    log_level: str = "warning"
    log_level_http: Optional[str] = "warning"
    log_requests: bool = False
    log_requests_level: int = 0
    show_time_cost: bool = False
    enable_metrics: bool = True
    decode_log_interval: int = 1

    def reject_unsupported(self) -> None:
        """Options that change what the server computes or where its memory lives, and that this engine
        does not implement, are errors - never silently ignored. (Options that only pick between
        third-party kernel back ends stay accepted: there is one HIP implementation.)"""
        if str(self.kv_cache_dtype) not in ("auto", "bfloat16", "bf16", "fp8", "fp8_e4m3"):
            raise ValueError(f"kv_cache_dtype={self.kv_cache_dtype!r}: the paged KV cache is stored in the model "
                             "dtype (auto / bf16) or as e4m3 bytes with scale 1 (fp8 / fp8_e4m3)")
        if self.cpu_offload_gb and float(self.cpu_offload_gb) > 0:
            raise ValueError(f"cpu_offload_gb={self.cpu_offload_gb}: weights are resident in HBM (288 GB per GPU); "
                             "host offload of server weights is not implemented - use tensor parallelism instead")
        if self.enable_memory_saver:
            raise ValueError("enable_memory_saver: releasing and re-mapping the server's KV cache and weights is not "
                             "implemented; colocated runs size the KV cache with `kv_cache_gb` / mem_fraction_static")
        if str(self.dtype) not in ("bfloat16", "float32"):
            raise ValueError(f"dtype={self.dtype!r}: the server computes in bfloat16 (or float32 on the CPU platform)")
        if self.torchao_config:
            raise ValueError(f"torchao_config={self.torchao_config!r}: weight quantisation is not implemented")

    @staticmethod
    def build_args(config: "InferenceServerConfig", tp_size, base_gpu_id, host, port,
                   dist_init_addr: Optional[str] = None, n_nodes: int = 1, node_rank: int = 0) -> Dict[str, Any]:
        config.reject_unsupported()
        args = dict(
            host=host, port=port, tokenizer_path=config.model_path, tp_size=tp_size, base_gpu_id=base_gpu_id,
            nnodes=n_nodes, node_rank=node_rank, dist_init_addr=dist_init_addr,
        )
        args.update(conf_as_dict(config))
        return args

    @staticmethod
    def build_cmd(config: "InferenceServerConfig", tp_size, base_gpu_id, host, port,
                  dist_init_addr: Optional[str] = None, n_nodes: int = 1, node_rank: int = 0) -> str:
        args = InferenceServerConfig.build_args(config, tp_size, base_gpu_id, host, port, dist_init_addr, n_nodes,
                                                node_rank)
        flags = []
        for k, v in args.items():
            if v is None or v is False or v == "":
                continue
            name = "--" + k.replace("_", "-")
            if v is True:
                flags.append(name)
            elif isinstance(v, (list, tuple)):
                flags.append(f"{name} {' '.join(map(str, v))}")
            else:
                flags.append(f"{name} {v}")
        return f"python3 -m areal_vllm_amd.inference.server {' '.join(flags)}"


SGLangConfig = InferenceServerConfig  # name used by the reference's YAMLs and scripts


@dataclass
class InferenceEngineConfig:
    experiment_name: Optional[str] = None
    trial_name: Optional[str] = None
    max_concurrent_rollouts: Optional[int] = None
    queue_size: Optional[int] = None
    consumer_batch_size: int = 1
    max_head_offpolicyness: int = 0
    enable_rollout_tracing: bool = False
    schedule_policy: str = field(default="round_robin", metadata={"choices": ["round_robin"]})
    setup_timeout: float = 120.0
    request_timeout: float = 3600
    request_retries: int = 3


@dataclass
class _Timer:
    experiment_name: str = MISSING
    trial_name: str = MISSING
    fileroot: str = MISSING
    freq_epochs: Optional[int] = None
    freq_steps: Optional[int] = None
    freq_secs: Optional[int] = None


@dataclass
class EvaluatorConfig(_Timer):
    pass


@dataclass
class SaverConfig(_Timer):
    pass


@dataclass
class RecoverConfig(_Timer):
    mode: str = field(default="disabled", metadata={"help": "disabled | auto | fault | resume"})
    retries: int = 3


@dataclass
class WandBConfig:
    mode: str = "disabled"
    entity: Optional[str] = None
    project: Optional[str] = None
    name: Optional[str] = None
    job_type: Optional[str] = None
    group: Optional[str] = None
    notes: Optional[str] = None
    tags: Optional[List[str]] = None
    config: Optional[Dict] = None


@dataclass
class SwanlabConfig:
    project: Optional[str] = None
    name: Optional[str] = None
    config: Optional[Dict] = None
    logdir: Optional[str] = None
    mode: Optional[str] = "local"
    api_key: Optional[str] = os.getenv("SWANLAB_API_KEY", None)


@dataclass
class TensorBoardConfig:
    path: Optional[str] = None


@dataclass
class StatsLoggerConfig:
    experiment_name: str = MISSING
    trial_name: str = MISSING
    fileroot: str = MISSING
    wandb: WandBConfig = field(default_factory=WandBConfig)
    swanlab: SwanlabConfig = field(default_factory=SwanlabConfig)
    tensorboard: TensorBoardConfig = field(default_factory=TensorBoardConfig)


@dataclass
class NameResolveConfig:
    type: str = field(default="nfs", metadata={"choices": ["nfs", "etcd3", "ray", "memory"]})
    nfs_record_root: str = "/tmp/areal/name_resolve"
    etcd3_addr: str = "localhost:2379"
    ray_actor_name: str = "ray_kv_store"


@dataclass
class ClusterSpecConfig:
    name_resolve: NameResolveConfig = field(default_factory=NameResolveConfig)
    cluster_name: str = "local"
    fileroot: str = field(default_factory=get_user_tmp)
    n_nodes: int = 32
    n_gpus_per_node: int = 8


@dataclass
class DatasetConfig:
    path: str = MISSING
    type: Optional[str] = None
    batch_size: int = 1
    shuffle: bool = True
    pin_memory: bool = False
    num_workers: int = 0
    drop_last: bool = True
    reward_fn: Optional[str] = None
    max_length: Optional[int] = None


@dataclass
class SlurmLauncherConfig:
    srun_additional_args: str = "--mpi=pmi2 -K --chdir $PWD"
    container_type: str = field(default="apptainer", metadata={"choices": ["apptainer", "none"]})
    mount: str = "/storage:/storage"
    trainer_image: Optional[str] = None
    inference_server_image: Optional[str] = None


@dataclass
class LauncherConfig:
    inference_server_cpus_per_gpu: int = 4
    inference_server_mem_per_gpu: int = 32 * 1024
    trainer_cpus_per_gpu: int = 4
    trainer_mem_per_gpu: int = 32 * 1024
    inference_server_env_vars: str = ""
    trainer_env_vars: str = ""
    slurm: SlurmLauncherConfig = field(default_factory=SlurmLauncherConfig)


@dataclass
class BaseExperimentConfig:
    experiment_name: str = MISSING
    trial_name: str = MISSING
    cluster: ClusterSpecConfig = field(default_factory=ClusterSpecConfig)
    allocation_mode: str = ""
    seed: int = 1
    total_train_epochs: int = 1
    total_train_steps: Optional[int] = None
    total_train_n_seqs: Optional[int] = None
    tokenizer_path: str = ""
    train_dataset: DatasetConfig = field(default_factory=DatasetConfig)
    valid_dataset: Optional[DatasetConfig] = None
    saver: SaverConfig = field(default_factory=SaverConfig)
    evaluator: EvaluatorConfig = field(default_factory=EvaluatorConfig)
    stats_logger: StatsLoggerConfig = field(default_factory=StatsLoggerConfig)
    recover: RecoverConfig = field(default_factory=RecoverConfig)
    sglang: InferenceServerConfig = field(default_factory=InferenceServerConfig)
    launcher: LauncherConfig = field(default_factory=LauncherConfig)


@dataclass
class SFTConfig(BaseExperimentConfig):
    model: TrainEngineConfig = field(default_factory=TrainEngineConfig)


@dataclass
class GRPOConfig(BaseExperimentConfig):
    async_training: bool = True
    rollout_weight_update: str = field(default="xgmi", metadata={
        "choices": ["nccl", "xgmi", "ipc", "disk"],
        "help": "transport to remote rollout servers: xgmi = collective push with every trainer rank sending a share "
                "of the parameter chunks, nccl = all chunks broadcast by trainer rank 0, ipc = servers on the same "
                "node map the trainer's flat parameter buffer and pull it with one device-to-device copy, "
                "disk = checkpoint files"})
    colocated_kv_cache_gb: Optional[float] = field(default=None, metadata={
        "help": "KV-cache budget of the in-process rollout engine (colocated mode); default: mem_fraction_static"})
    gconfig: GenerationHyperparameters = field(default_factory=GenerationHyperparameters)
    rollout: InferenceEngineConfig = field(default_factory=InferenceEngineConfig)
    actor: PPOActorConfig = field(default_factory=PPOActorConfig)
    ref: PPOActorConfig = field(default_factory=PPOActorConfig)


@dataclass
class PPOConfig(GRPOConfig):
    critic: PPOCriticConfig = field(default_factory=PPOCriticConfig)


# ------------------------------------------------------------------------------ YAML machinery
_INTERP = re.compile(r"\$\{([^${}]+)\}")


def _parse_scalar(text: str) -> Any:
    try:
        return yaml.safe_load(text)
    except yaml.YAMLError:
        return text


def _set_dotted(d: Dict, key: str, value: Any):
    parts = key.split(".")
    cur = d
    for p in parts[:-1]:
        if p not in cur or not isinstance(cur[p], dict):
            cur[p] = {}
        cur = cur[p]
    cur[parts[-1]] = value


def _get_dotted(d: Dict, key: str):
    cur = d
    for p in key.split("."):
        if not isinstance(cur, dict) or p not in cur:
            raise KeyError(key)
        cur = cur[p]
    return cur


def apply_overrides(cfg: Dict, overrides: List[str]) -> Dict:
    """Apply hydra-style `a.b=c` overrides (`+a.b=c` adds, `~a.b` deletes)."""
    for ov in overrides:
        if ov.startswith("~"):
            parts = ov[1:].split(".")
            cur = cfg
            for p in parts[:-1]:
                cur = cur.get(p, {})
            cur.pop(parts[-1], None)
            continue
        if "=" not in ov:
            raise ValueError(f"override must look like key=value, got {ov!r}")
        k, v = ov.split("=", 1)
        k = k.lstrip("+")
        _set_dotted(cfg, k, _parse_scalar(v))
    return cfg


def resolve_interpolations(cfg: Dict, defaults: Optional[Dict] = None) -> Dict:
    """Resolve `${dotted.key}` references (against the YAML first, then the dataclass defaults)."""

    def lookup(key: str, stack):
        if key.startswith("oc.env:"):
            name, _, dflt = key[len("oc.env:"):].partition(",")
            return os.environ.get(name, dflt or None)
        if key in stack:
            raise ValueError(f"interpolation cycle at ${{{key}}}")
        try:
            val = _get_dotted(cfg, key)
        except KeyError:
            if defaults is None:
                raise KeyError(f"interpolation key '{key}' not found")
            val = _get_dotted(defaults, key)
        return resolve(val, stack + [key])

    def resolve(v, stack):
        if isinstance(v, str):
            m = _INTERP.fullmatch(v.strip())
            if m:
                return lookup(m.group(1).strip(), stack)
            return _INTERP.sub(lambda mm: str(lookup(mm.group(1).strip(), stack)), v)
        if isinstance(v, dict):
            return {k: resolve(x, stack) for k, x in v.items()}
        if isinstance(v, list):
            return [resolve(x, stack) for x in v]
        return v

    return resolve(cfg, [])


def _strip_optional(tp):
    origin = typing.get_origin(tp)
    if origin is Union or origin is getattr(types, "UnionType", None):
        args = [a for a in typing.get_args(tp) if a is not type(None)]
        return (args[0] if len(args) == 1 else tp), True
    return tp, False


def _coerce(value: Any, tp, path: str):
    tp, optional = _strip_optional(tp)
    if value is None:
        return None
    if value == MISSING:
        return value
    if dataclasses.is_dataclass(tp):
        if isinstance(value, tp):
            return value
        if not isinstance(value, dict):
            raise TypeError(f"{path}: expected a mapping for {tp.__name__}, got {type(value).__name__}")
        return to_structured_cfg(value, tp, path)
    origin = typing.get_origin(tp)
    if origin in (list, List):
        (et,) = typing.get_args(tp) or (Any,)
        if not isinstance(value, (list, tuple)):
            raise TypeError(f"{path}: expected a list")
        return [_coerce(v, et, f"{path}[{i}]") for i, v in enumerate(value)]
    if origin in (dict, Dict):
        return dict(value)
    if tp is bool:
        if isinstance(value, bool):
            return value
        if isinstance(value, str) and value.lower() in ("true", "false"):
            return value.lower() == "true"
        raise TypeError(f"{path}: expected bool, got {value!r}")
    if tp is int:
        if isinstance(value, bool):
            raise TypeError(f"{path}: expected int, got bool")
        if isinstance(value, int):
            return value
        if isinstance(value, float) and value.is_integer():
            return int(value)
        if isinstance(value, str):
            return int(float(value))
        raise TypeError(f"{path}: expected int, got {value!r}")
    if tp is float:
        if isinstance(value, (int, float)) and not isinstance(value, bool):
            return float(value)
        if isinstance(value, str):
            return float(value)
        raise TypeError(f"{path}: expected float, got {value!r}")
    if tp is str:
        return str(value)
    return value


def to_structured_cfg(cfg: Dict, config_cls, path: str = ""):
    """Merge a plain dict into the dataclass defaults, validating keys and coercing types."""
    if cfg is None:
        cfg = {}
    hints = typing.get_type_hints(config_cls)
    names = {f.name for f in dataclasses.fields(config_cls)}
    unknown = set(cfg) - names
    if unknown:
        raise KeyError(f"unknown config key(s) {sorted(unknown)} under '{path or config_cls.__name__}'")
    kwargs = {}
    for f in dataclasses.fields(config_cls):
        p = f"{path}.{f.name}" if path else f.name
        if f.name in cfg:
            val = cfg[f.name]
            ftp, _ = _strip_optional(hints[f.name])
            if isinstance(val, dict) and dataclasses.is_dataclass(ftp):
                kwargs[f.name] = to_structured_cfg(val, ftp, p)
            else:
                kwargs[f.name] = _coerce(val, hints[f.name], p)
    return config_cls(**kwargs)


def conf_as_dict(cfg) -> Dict[str, Any]:
    if dataclasses.is_dataclass(cfg):
        return asdict(cfg)
    return dict(cfg)


def _check_missing(obj, path=""):
    if dataclasses.is_dataclass(obj):
        for f in dataclasses.fields(obj):
            _check_missing(getattr(obj, f.name), f"{path}.{f.name}" if path else f.name)
    elif obj == MISSING:
        raise ValueError(f"missing mandatory config value: {path}")


def parse_cli_args(argv: List[str]):
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", help="path of the main YAML configuration file", required=True)
    args, overrides = parser.parse_known_args(argv)
    config_file = Path(args.config).absolute()
    assert config_file.exists(), f"Config file {config_file} does not exist."
    with open(config_file) as f:
        cfg = yaml.safe_load(f) or {}
    cfg = apply_overrides(cfg, overrides)
    return cfg, config_file


def load_expr_config(argv: List[str], config_cls, check_missing: bool = False):
    """YAML + overrides -> validated dataclass instance; also configures name_resolve."""
    cfg, config_file = parse_cli_args(argv)
    defaults = _defaults_dict(config_cls)
    cfg = resolve_interpolations(cfg, defaults)
    obj = to_structured_cfg(cfg, config_cls)
    assert isinstance(obj, BaseExperimentConfig)
    if check_missing:
        _check_missing(obj)
    from ..utils import name_resolve

    name_resolve.reconfigure(obj.cluster.name_resolve)
    return obj, str(config_file)


def _defaults_dict(config_cls) -> Dict:
    try:
        return asdict(config_cls())
    except TypeError:
        return {}


def save_config(cfg, path: str):
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "w") as f:
        yaml.safe_dump(conf_as_dict(cfg), f, sort_keys=False)
