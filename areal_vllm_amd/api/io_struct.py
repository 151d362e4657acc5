"""Plain data structures exchanged between workflows, inference engines and train engines.

Field-for-field counterparts of ``areal/api/io_struct.py:22-179`` in the reference, plus the extra
weight-update transport this framework adds (``xgmi``: sharded peer-to-peer copies, see
``engine/weight_update.py``). Unlike the reference, ``output_versions`` is always populated by the native
engine with the weight version that produced each token.
"""
import os
import uuid
from dataclasses import dataclass, field
from typing import TYPE_CHECKING, Any, Dict, List, Literal, Optional, Tuple

import numpy as np
import torch

from .alloc_mode import AllocationMode
from .cli_args import GenerationHyperparameters

if TYPE_CHECKING:
    from .engine_api import TrainEngine


@dataclass
class ModelRequest:
    rid: str = field(default_factory=lambda: str(uuid.uuid4()))
    input_ids: List[int] = field(default_factory=list)
    gconfig: GenerationHyperparameters = field(default_factory=GenerationHyperparameters)
    metadata: Dict[str, Any] = field(default_factory=dict)
    tokenizer: Optional[Any] = None
    image_data: Optional[List[Any]] = field(default_factory=list)
    processor: Optional[Any] = None

    def copy(self):
        return ModelRequest(rid=self.rid, input_ids=list(self.input_ids), gconfig=self.gconfig.new(),
                            metadata=dict(self.metadata), tokenizer=self.tokenizer,
                            image_data=list(self.image_data or []), processor=self.processor)


@dataclass
class ModelResponse:
    input_tokens: List[int] = field(default_factory=list)
    output_tokens: List[int] = field(default_factory=list)
    output_logprobs: List[float] = field(default_factory=list)
    output_versions: List[int] = field(default_factory=list)
    stop_reason: Literal["length", "stop", "interrupt", "tool_calls"] = "stop"
    tokenizer: Optional[Any] = None
    input_images: List[Any] = field(default_factory=list)
    processor: Optional[Any] = None
    latency: float = float("inf")
    ttft: float = float("inf")
    itl: List[float] = field(default_factory=list)

    @property
    def input_len(self) -> int:
        return len(self.input_tokens)

    @property
    def output_len(self) -> int:
        return len(self.output_tokens)


@dataclass
class FinetuneSpec:
    total_train_epochs: int
    dataset_size: int
    train_batch_size: int

    @property
    def total_train_steps(self):
        return self.total_train_epochs * (self.dataset_size // self.train_batch_size)

    @property
    def steps_per_epoch(self):
        return self.dataset_size // self.train_batch_size


@dataclass
class ParamSpec:
    name: str
    shape: Tuple
    dtype: str

    @property
    def size(self) -> int:
        return int(getattr(torch, self.dtype).itemsize * np.prod(self.shape))


@dataclass
class WeightUpdateMeta:
    type: Literal["disk", "nccl", "xgmi", "inproc", "ipc"]
    path: Optional[str] = None
    alloc_mode: Optional[AllocationMode] = None
    nccl_master_address: str = "127.0.0.1"
    nccl_master_port: int = 29500
    nccl_param_specs: List[List[ParamSpec]] = field(default_factory=list)
    nccl_group_name: str = "update_weight_group"
    # Sharded push: trainer ranks 0 .. n-1 (all hold the full replica under ZeRO-1 data parallelism) each own a
    # broadcast group with the rollout servers and send every n-th chunk, so an update leaves through n
    # devices' links at once instead of funnelling through rank 0 (the reference's scheme, n = 1).
    nccl_n_senders: int = 1
    nccl_master_ports: List[int] = field(default_factory=list)

    # Peer-memory push ("ipc"): one exported flat parameter buffer per data-parallel trainer rank (all of them
    # hold the full replica); rollout server i pulls from buffer i % n with a device-to-device copy.
    ipc_buffers: List[Dict[str, Any]] = field(default_factory=list)
    ipc_layout: str = ""  # fingerprint of the flat parameter layout the buffers use
    ipc_used_bytes: int = 0  # bytes of parameters at the head of every buffer (the tail is shard padding)
    # parallel layout of the exporting trainer: {dp, tp, pp, vpp, vocab_parallel, zero3}; `ipc_buffers[r]` belongs to
    # trainer rank r. Servers derive from it which ranges of which buffers they pull.
    ipc_src: Dict[str, Any] = field(default_factory=dict)

    def sender_port(self, s: int) -> int:
        return self.nccl_master_ports[s] if self.nccl_master_ports else self.nccl_master_port + s

    def sender_group_name(self, s: int) -> str:
        return self.nccl_group_name if s == 0 else f"{self.nccl_group_name}_{s}"

    @classmethod
    def from_disk(cls, experiment_name: str, trial_name: str, file_root: str, name: str = "default"):
        from ..utils.saver import Saver

        path = os.path.join(Saver.get_model_save_root(experiment_name, trial_name, file_root, name), "weight_update")
        return cls(type="disk", path=path)

    @classmethod
    def from_fsdp_nccl(cls, allocation_mode: AllocationMode, fsdp_engine: "TrainEngine",
                       nccl_group_name: str = "update_weight_group", weight_chunked_mem_mb: int = 1024,
                       n_senders: int = 1):
        from ..platforms import current_platform
        from ..utils.network import find_free_ports, gethostip

        n_senders = max(1, min(n_senders, len(fsdp_engine.get_param_specs(weight_chunked_mem_mb))))
        ports = find_free_ports(n_senders)
        return cls(
            type="nccl",  # on CPU the same broadcast protocol runs over gloo
            alloc_mode=allocation_mode,
            nccl_master_address=gethostip(),
            nccl_master_port=ports[0],
            nccl_param_specs=fsdp_engine.get_param_specs(weight_chunked_mem_mb),
            nccl_group_name=nccl_group_name,
            nccl_n_senders=max(1, n_senders),
            nccl_master_ports=list(ports),
        )

    from_nccl = from_fsdp_nccl

    @classmethod
    def from_ipc(cls, engine: "TrainEngine"):
        """Collective over the trainer's world: every rank exports the memory behind its flat parameter buffer."""
        return cls(type="ipc", **engine.export_weight_buffers())

    @classmethod
    def from_inproc(cls):
        """Trainer and rollout engine live in the same process (colocated): device-to-device copy."""
        return cls(type="inproc")


@dataclass
class SaveLoadMeta:
    path: str
    weight_format: str  # hf | dcp | naive_distributed
    with_optim: bool
    tokenizer: Optional[Any] = None
    processor: Optional[Any] = None
    base_model_path: Optional[str] = None
    naive_distributed: bool = False


@dataclass
class RolloutStat:
    submitted: int = 0
    accepted: int = 0
    running: int = 0


@dataclass
class StepInfo:
    epoch: int
    epoch_step: int
    global_step: int
    steps_per_epoch: int

    def next(self):
        last = self.epoch_step == self.steps_per_epoch - 1
        return StepInfo(epoch=self.epoch + int(last), epoch_step=0 if last else self.epoch_step + 1,
                        global_step=self.global_step + 1, steps_per_epoch=self.steps_per_epoch)
