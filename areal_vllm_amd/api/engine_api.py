"""Abstract engine interfaces (``TrainEngine`` / ``InferenceEngine``) and the scheduling descriptor.

Same method surface as ``areal/api/engine_api.py:23-227`` in the reference so training scripts written
against that API run on the native engines.
"""
import abc
from concurrent.futures import Future
from dataclasses import dataclass, field
from typing import TYPE_CHECKING, Any, Callable, Dict, List, Optional

import torch

from .io_struct import FinetuneSpec, ModelRequest, ModelResponse, ParamSpec, SaveLoadMeta, WeightUpdateMeta

if TYPE_CHECKING:
    from .alloc_mode import ParallelStrategy
    from .workflow_api import RolloutWorkflow


@dataclass
class Scheduling:
    cpu: int
    gpu: int
    mem: int
    nodelist: Optional[str] = None
    exclude: Optional[str] = None
    partition: Optional[str] = None
    container_image: Optional[str] = None
    type: Optional[str] = None
    env_vars: Dict[str, str] = field(default_factory=dict)
    time_limit: Optional[str] = None
    begin: Optional[str] = None
    deadline: Optional[str] = None


class TrainEngine(abc.ABC):
    def create_process_group(self, parallel_strategy: Optional["ParallelStrategy"] = None):
        raise NotImplementedError()

    def initialize(self, addr: Optional[str], ft_spec: Optional[FinetuneSpec]):
        raise NotImplementedError()

    @property
    def data_parallel_group(self):
        raise NotImplementedError()

    @property
    def data_parallel_rank(self) -> int:
        raise NotImplementedError()

    @property
    def data_parallel_world_size(self) -> int:
        raise NotImplementedError()

    def current_data_parallel_head(self) -> int:
        raise NotImplementedError()

    def is_data_parallel_head(self) -> bool:
        raise NotImplementedError()

    @property
    def context_and_model_parallel_group(self):
        raise NotImplementedError()

    def get_scheduling_config(self) -> Scheduling:
        raise NotImplementedError()

    def destroy(self):
        pass

    def train(self, mode: bool = True):
        raise NotImplementedError()

    def eval(self):
        return self.train(False)

    def upload_weights(self, meta: WeightUpdateMeta):
        raise NotImplementedError()

    def get_param_specs(self, weight_chunked_mem_mb: int = 1024) -> List[List[ParamSpec]]:
        raise NotImplementedError()

    def set_version(self, version: int):
        raise NotImplementedError()

    def get_version(self) -> int:
        raise NotImplementedError()

    def save(self, meta: SaveLoadMeta):
        raise NotImplementedError()

    def load(self, meta: SaveLoadMeta):
        raise NotImplementedError()

    def step_lr_scheduler(self):
        raise NotImplementedError()

    def train_batch(self, input_: Dict[str, Any], loss_fn: Callable, loss_weight_fn: Callable) -> Dict[str, float]:
        raise NotImplementedError()

    @torch.no_grad()
    def eval_batch(self, input_: Dict[str, Any], loss_fn: Callable, loss_weight_fn: Callable) -> Optional[torch.Tensor]:
        raise NotImplementedError()

    @torch.no_grad()
    def forward(self, input_: Dict[str, Any], output_seqlens: Optional[List[int]] = None,
                post_hook: Optional[Callable] = None, aggregate_fn: Callable = torch.cat) -> Any:
        raise NotImplementedError()


class InferenceEngine(abc.ABC):
    def initialize(self, addr: Optional[str] = None, ft_spec: Optional[FinetuneSpec] = None, **kwargs):
        raise NotImplementedError()

    def destroy(self):
        pass

    async def agenerate(self, req: ModelRequest) -> ModelResponse:
        raise NotImplementedError()

    def update_weights(self, meta: WeightUpdateMeta) -> Future:
        raise NotImplementedError()

    def set_version(self, version: int) -> None:
        raise NotImplementedError()

    def get_version(self) -> int:
        raise NotImplementedError()

    def submit(self, data: Dict[str, Any], workflow: Optional["RolloutWorkflow"] = None,
               workflow_builder: Optional[Callable] = None) -> None:
        raise NotImplementedError()

    def wait(self, count: int, timeout: Optional[float] = None, should_accept: Optional[Callable] = None):
        raise NotImplementedError()

    def rollout_batch(self, data: List[Dict[str, Any]], workflow: Optional["RolloutWorkflow"] = None,
                      workflow_builder: Optional[Callable] = None, should_accept: Optional[Callable] = None):
        raise NotImplementedError()

    def prepare_batch(self, dataloader, workflow: Optional["RolloutWorkflow"] = None,
                      workflow_builder: Optional[Callable] = None, should_accept: Optional[Callable] = None):
        raise NotImplementedError()

    def pause(self):
        raise NotImplementedError()

    def resume(self):
        raise NotImplementedError()
