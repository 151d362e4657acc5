"""CPU stand-in train backend for system tests and dry runs.

The worker runtime is exercised without a GPU by hosting the *same* native engine on the host: fp32
parameters, gloo collectives, PyTorch reference paths of the ops. Unlike the reference's mock
(``realhf/impl/model/backend/mock_train.py``, a hand-rolled engine with manual gradient all-reduce) nothing
is mocked except the device, so ZeRO sharding, micro-batching and the loss path are the production code.
"""
from typing import Any, Dict, Optional

from ...api.io_struct import FinetuneSpec
from ...system.model_api import Model, register_backend
from .native import NativeTrainBackend


class MockTrainBackend(NativeTrainBackend):
    def __init__(self, optimizer: Optional[Dict[str, Any]] = None, **kw):
        super().__init__(optimizer or dict(lr=1e-3, weight_decay=0.0, lr_scheduler_type="constant",
                                           warmup_steps_proportion=0.0))

    def _initialize(self, model: Model, spec: FinetuneSpec) -> Model:
        model.module.config.dtype = "float32"
        model = super()._initialize(model, spec)
        model.backend_name = "mock_train"
        return model


register_backend("mock_train", MockTrainBackend)
