"""Model builder and backends that host the native engine inside a model worker.

* model type ``native``: an un-initialised ``NativeTrainEngine`` for a model path / preset (critics get a
  value head);
* backend ``native_train``: ZeRO-sharded AdamW training (role of the reference's Megatron backend,
  ``realhf/impl/model/backend/megatron.py:561-761``);
* backend ``inference``: forward-only hosting for reference / reward / frozen models
  (``realhf/impl/model/backend/inference.py``).

The worker attaches the model's rank grid (``model.grid``) before the backend initialises the engine, so
several models with different layouts can live in one world of model workers.
"""
import dataclasses
from typing import Any, Dict, Optional

import torch

from ...api.cli_args import MicroBatchSpec, NativeEngineConfig, OptimizerConfig, TrainEngineConfig
from ...api.io_struct import FinetuneSpec, SaveLoadMeta
from ...engine.train_engine import NativeTrainEngine
from ...system.model_api import Model, ModelBackend, register_backend, register_model


def make_native_model(name, device, path: str, is_critic: bool = False, init_from_scratch: bool = False,
                      init_critic_from_actor: bool = False, dtype: str = "bfloat16",
                      gradient_checkpointing: bool = False, tokenizer_path: Optional[str] = None,
                      experiment_name: str = "exp", trial_name: str = "trial",
                      native: Optional[Dict[str, Any]] = None) -> Model:
    cfg = TrainEngineConfig(experiment_name=experiment_name, trial_name=trial_name, path=path,
                            init_from_scratch=init_from_scratch, init_critic_from_actor=init_critic_from_actor,
                            dtype=dtype, gradient_checkpointing=gradient_checkpointing,
                            native=NativeEngineConfig(**(native or {})), optimizer=None)
    eng = NativeTrainEngine(cfg)
    eng.is_critic = is_critic
    tok = None
    if tokenizer_path:
        from ...utils.hf_utils import load_hf_tokenizer

        tok = load_hf_tokenizer(tokenizer_path)
    return Model(name=name, module=eng, tokenizer=tok, device=device, dtype=getattr(torch, dtype),
                 config=dict(path=path, is_critic=is_critic))


class NativeTrainBackend(ModelBackend):
    def __init__(self, optimizer: Optional[Dict[str, Any]] = None, **_):
        oc = optimizer or {}
        self.optimizer = oc if isinstance(oc, OptimizerConfig) else OptimizerConfig(**oc)

    def _initialize(self, model: Model, spec: FinetuneSpec) -> Model:
        eng: NativeTrainEngine = model.module
        eng.optimizer_config = self.optimizer
        eng.config.optimizer = self.optimizer
        eng.create_process_group(grid=getattr(model, "grid", None))
        eng.device = torch.device(model.device)
        eng.initialize(None, spec)
        model.backend_name = "native_train"
        return model

    def save(self, model: Model, save_dir: str):
        model.module.save(SaveLoadMeta(path=save_dir, weight_format="hf", with_optim=True, tokenizer=model.tokenizer))

    def load(self, model: Model, load_dir: str):
        model.module.load(SaveLoadMeta(path=load_dir, weight_format="hf", with_optim=True))


class InferenceBackend(ModelBackend):
    def __init__(self, **_):
        pass

    def _initialize(self, model: Model, spec: FinetuneSpec) -> Model:
        eng: NativeTrainEngine = model.module
        eng.optimizer_config = None
        eng.config.optimizer = None
        eng.create_process_group(grid=getattr(model, "grid", None))
        eng.device = torch.device(model.device)
        eng.initialize(None, spec)
        eng.train(False)
        model.backend_name = "inference"
        return model

    def load(self, model: Model, load_dir: str):
        model.module.load(SaveLoadMeta(path=load_dir, weight_format="hf", with_optim=False))


register_model("native", make_native_model)
register_model("null", lambda name, device, **kw: Model(name=name, module=None, device=device))
register_backend("native_train", NativeTrainBackend)
register_backend("megatron", NativeTrainBackend)  # name used by configurations written for the reference
register_backend("inference", InferenceBackend)
