"""Model builder and backends that host the native engine inside a model worker.

* model type ``native``: an un-initialised ``NativeTrainEngine`` for a model path / preset (critics get a
  value head);
* backend ``native_train``: ZeRO-sharded AdamW training (role of the reference's Megatron backend,
  ``realhf/impl/model/backend/megatron.py:561-761``);
* backend ``inference``: forward-only hosting for reference / reward / frozen models
  (``realhf/impl/model/backend/inference.py``);
* backend ``native_generation`` (also registered as ``vllm`` and ``sglang``, the names configurations written for the
  reference use): a generation-only model worker around the paged-KV rollout engine
  (``realhf/impl/model/backend/vllm.py:36-247``, ``realhf/impl/model/backend/sglang.py``).

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


class GenerationEngineModule:
    """What a generation-only model hosts: the rollout engine plus the calls the worker runtime and the PPO
    interface make on it (the role of ``vLLMGenerationEngine``, ``realhf/impl/model/backend/vllm.py:36-163``)."""

    def __init__(self, engine, model_config, hybrid_train: bool = False):
        self.engine = engine
        self.model_config = model_config
        self.hybrid_train = hybrid_train
        self.device = engine.device

    def train(self, mode: bool = True):
        return self

    def eval(self):
        return self

    def onload(self):
        pass

    def update_weights_from_disk(self, path: str, version: Optional[int] = None):
        self.engine.load_weights_from_disk(path, version=version)

    def generate_tokens(self, prompts, n: int, g, stop_token_ids):
        """`n` answers per prompt, in prompt order -> list of GenerationResult."""
        res = self.engine.generate_blocking(
            [p for p in prompts for _ in range(n)], max_new_tokens=g.max_new_tokens, min_new_tokens=g.min_new_tokens,
            temperature=g.temperature, top_p=g.top_p, top_k=g.top_k, greedy=g.greedy, stop_token_ids=stop_token_ids)
        if self.hybrid_train:  # the pages go back to the pool between calls (a trainer shares the device)
            self.engine.flush_cache()
        return res

    def generate(self, input_, mb_spec, tokenizer, gconfig):
        """The reference's calling convention: (generated ids [B, L] padded, their log-probs [B, L], None)."""
        prompts = [d.data["packed_input_ids"].tolist() if "packed_input_ids" in d.data else
                   d.data["packed_prompts"].tolist() for d in input_.unpack()]
        stops = [int(t) for t in (getattr(tokenizer, "eos_token_id", None), getattr(tokenizer, "pad_token_id", None))
                 if t is not None]
        res = self.generate_tokens(prompts, gconfig.n_samples, gconfig, stops)
        L = max(1, max(len(r.output_tokens) for r in res))
        pad = getattr(tokenizer, "pad_token_id", None) or 0
        ids = torch.full((len(res), L), int(pad), dtype=torch.long)
        lps = torch.zeros(len(res), L, dtype=torch.float32)
        for i, r in enumerate(res):
            ids[i, :len(r.output_tokens)] = torch.tensor(r.output_tokens, dtype=torch.long)
            lps[i, :len(r.output_tokens)] = torch.tensor(r.output_logprobs, dtype=torch.float32)
        return ids.to(self.device), lps.to(self.device), None

    def destroy(self):
        self.engine.destroy()


class GenerationBackend(ModelBackend):
    """Turns a ``native`` model into a generation-only one: the (un-initialised) trainer object the model builder
    made is replaced by a rollout engine for the same path / preset. Options follow the reference's
    ``vLLMConfig`` / ``SGLangConfig`` where this engine has the knob (``max_num_seqs``, ``max_model_len`` /
    ``context_length``, ``gpu_memory_utilization`` / ``mem_fraction_static``, ``enable_prefix_caching``,
    ``num_scheduler_steps``, ``hybrid_train``); others are accepted and have no effect here."""

    def __init__(self, model_path: str = "", max_num_seqs: int = 256, max_model_len: Optional[int] = None,
                 context_length: Optional[int] = None, gpu_memory_utilization: Optional[float] = None,
                 mem_fraction_static: Optional[float] = None, kv_cache_gb: Optional[float] = None,
                 enable_prefix_caching: bool = True, num_scheduler_steps: int = 8, hybrid_train: bool = False,
                 enforce_eager: bool = False, dtype: Optional[str] = None, seed: int = 1, **_):
        self.model_path = model_path
        self.max_num_seqs = max_num_seqs
        self.context_length = context_length or max_model_len or 32768
        self.mem_fraction = mem_fraction_static or gpu_memory_utilization or 0.9
        self.kv_cache_gb = kv_cache_gb
        self.prefix_caching = enable_prefix_caching
        self.decode_steps = num_scheduler_steps
        self.hybrid_train = hybrid_train
        self.eager = enforce_eager
        self.dtype = dtype
        self.seed = seed

    def _initialize(self, model: Model, spec: FinetuneSpec) -> Model:
        from ...api.cli_args import InferenceServerConfig
        from ...inference.engine import NativeInferenceEngine

        tr: NativeTrainEngine = model.module
        path = self.model_path or tr.config.path
        dev = torch.device(model.device)
        cfg = InferenceServerConfig(
            model_path=path, dtype=self.dtype or tr.config.dtype, random_seed=self.seed,
            context_length=self.context_length, max_running_requests=self.max_num_seqs,
            mem_fraction_static=self.mem_fraction, disable_radix_cache=not self.prefix_caching,
            num_continuous_decode_steps=self.decode_steps, disable_cuda_graph=self.eager or dev.type != "cuda")
        on_gpu = dev.type == "cuda"
        eng = NativeInferenceEngine(cfg, device=dev, num_pages=None if on_gpu else 1024,
                                    kv_cache_gb=self.kv_cache_gb if on_gpu else None)
        eng.initialize(load_weights=not tr.config.init_from_scratch)
        model.module = GenerationEngineModule(eng, eng.model_config, self.hybrid_train)
        model.backend_name = "native_generation"
        return model

    def load(self, model: Model, load_dir: str):
        model.module.update_weights_from_disk(load_dir)


register_model("native", make_native_model)
register_model("null", lambda name, device, **kw: Model(name=name, module=None, device=device))
register_backend("native_train", NativeTrainBackend)
register_backend("megatron", NativeTrainBackend)  # name used by configurations written for the reference
register_backend("inference", InferenceBackend)
for _name in ("native_generation", "vllm", "sglang"):  # the latter two: names of the reference's generation backends
    register_backend(_name, GenerationBackend)
