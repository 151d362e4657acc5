"""Value types exchanged between rollout workers, the generation-server manager and the servers.

``GenReqMeta`` / ``GenRespMeta`` / ``AllocateRolloutInput`` are the bodies of the manager's routing
endpoints, ``APIGenerateInput`` / ``APIGenerateOutput`` one (possibly chunked) generation call and
``BundledGenerationOutputs`` the finished group of answers of one prompt. Contract of
``realhf/api/core/model_api.py:40-250`` (GenReqMeta ... BundledGenerationOutputs).
"""
import dataclasses
from typing import Any, Dict, Hashable, List, Optional

from ..api.cli_args import GenerationHyperparameters


@dataclasses.dataclass
class GenReqMeta:
    qid: Hashable
    prompt_len: int
    group_size: int
    new_token_budget: int
    predicted_new_tokens: Optional[int] = None
    previous_server_url: str = ""
    previous_version: int = -1


@dataclasses.dataclass
class GenRespMeta:
    qid: Hashable
    accepted: bool
    n_tokens: int = 0


@dataclasses.dataclass
class ModelVersionReq:
    server_url: str = ""


@dataclasses.dataclass
class AllocateRolloutInput:
    qid: Hashable


@dataclasses.dataclass
class APIGenerateInput:
    qid: Hashable
    prompt_ids: List[int]
    input_ids: List[int]
    gconfig: GenerationHyperparameters
    stop_token_ids: List[int] = dataclasses.field(default_factory=list)
    return_logprob: bool = True
    version_start: int = -1
    prev_logprobs: List[float] = dataclasses.field(default_factory=list)
    metadata: Dict[str, Any] = dataclasses.field(default_factory=dict)


@dataclasses.dataclass
class APIGenerateOutput:
    qid: Hashable
    prompt_ids: List[int]
    input_ids: List[int]
    output_ids: List[List[int]] = dataclasses.field(default_factory=list)
    output_logprobs: List[List[float]] = dataclasses.field(default_factory=list)
    no_eos: List[bool] = dataclasses.field(default_factory=list)
    version_start: int = -1
    version_end: List[int] = dataclasses.field(default_factory=list)
    prev_logprobs: List[float] = dataclasses.field(default_factory=list)
    metadata: Dict[str, Any] = dataclasses.field(default_factory=dict)

    @classmethod
    def from_input(cls, inp: APIGenerateInput) -> "APIGenerateOutput":
        return cls(qid=inp.qid, prompt_ids=inp.prompt_ids, input_ids=inp.input_ids, version_start=inp.version_start,
                   prev_logprobs=list(inp.prev_logprobs), metadata=inp.metadata)

    @property
    def group_size(self) -> int:
        return len(self.output_ids)

    @property
    def prompt_len(self) -> int:
        return len(self.prompt_ids)

    @property
    def input_len(self) -> int:
        return len(self.input_ids)

    @property
    def gen_lens(self) -> List[int]:
        """Tokens generated so far, including those of earlier chunks."""
        return [len(o) + self.input_len - self.prompt_len for o in self.output_ids]


@dataclasses.dataclass
class BundledGenerationOutputs:
    qid: Hashable
    prompt_ids: List[int]
    output_ids: List[List[int]]
    seqs: List[List[int]]
    logprobs: List[List[float]]
    no_eos: List[bool]
    version_start: List[int]
    version_end: List[int]

    @classmethod
    def from_api_outputs(cls, outputs: List[APIGenerateOutput]) -> "BundledGenerationOutputs":
        assert len({o.qid for o in outputs}) == 1
        prompt = outputs[0].prompt_ids
        seqs, lps, outs, no_eos, vs, ve = [], [], [], [], [], []
        for o in outputs:
            for i in range(o.group_size):
                seq = o.input_ids + o.output_ids[i]
                seqs.append(seq)
                outs.append(seq[len(prompt):])
                # prompt positions carry a placeholder log-prob (length = len(seq) - 1, "short" layout)
                lp = [0.0] * (len(prompt) - 1) + list(o.prev_logprobs) + list(o.output_logprobs[i])
                lps.append(lp)
                no_eos.append(o.no_eos[i])
                vs.append(o.version_start)
                ve.append(o.version_end[i])
        return cls(qid=outputs[0].qid, prompt_ids=prompt, output_ids=outs, seqs=seqs, logprobs=lps, no_eos=no_eos,
                   version_start=vs, version_end=ve)

    @property
    def output_logprobs(self) -> List[List[float]]:
        return [lp[self.prompt_len - 1:] for lp in self.logprobs]

    @property
    def prompt_len(self) -> int:
        return len(self.prompt_ids)

    @property
    def output_lens(self) -> List[int]:
        return [len(o) for o in self.output_ids]

    @property
    def seqlens(self) -> List[int]:
        return [len(s) for s in self.seqs]


# ------------------------------------------------------------------------------------------------
# Model / backend / interface abstractions of the worker runtime (contract of
# ``realhf/api/core/model_api.py:340-967``: Model, ModelBackend, ModelInterface and their registries).
# ------------------------------------------------------------------------------------------------
import abc  # noqa: E402

import torch  # noqa: E402

from ..api.io_struct import FinetuneSpec  # noqa: E402
from .dfg import ModelInterfaceAbstraction, ModelInterfaceType, ModelName  # noqa: E402,F401


@dataclasses.dataclass
class ModelVersion:
    epoch: int = 0
    epoch_step: int = 0
    global_step: int = 0


@dataclasses.dataclass
class ModelAbstraction:
    type_: str = "native"
    args: Dict[str, Any] = dataclasses.field(default_factory=dict)


@dataclasses.dataclass
class ModelBackendAbstraction:
    type_: str = "native_train"
    args: Dict[str, Any] = dataclasses.field(default_factory=dict)


@dataclasses.dataclass
class Model:
    """One model shard hosted by a model worker. `module` is an engine exposing ``train_batch`` /
    ``eval_batch`` / ``forward`` on padded batches (``engine.train_engine.NativeTrainEngine`` or the CPU
    mock); the backend fills it in."""
    name: ModelName
    module: Any
    tokenizer: Any = None
    device: Any = "cpu"
    dtype: Optional[torch.dtype] = None
    version: ModelVersion = dataclasses.field(default_factory=ModelVersion)
    ft_spec: Optional[FinetuneSpec] = None
    backend_name: Optional[str] = None
    config: Any = None  # the abstraction args the model was built from

    def inc_version(self):
        self.version.global_step += 1
        self.version.epoch_step += 1
        if self.ft_spec is not None and self.ft_spec.steps_per_epoch > 0 and \
                self.version.epoch_step >= self.ft_spec.steps_per_epoch:
            self.version.epoch += 1
            self.version.epoch_step = 0


class ModelBackend(abc.ABC):
    @abc.abstractmethod
    def _initialize(self, model: Model, spec: FinetuneSpec) -> Model:
        raise NotImplementedError()

    def initialize(self, model: Model, spec: FinetuneSpec) -> Model:
        model.ft_spec = spec
        return self._initialize(model, spec)

    def destroy(self, model: Model):
        pass

    def save(self, model: Model, save_dir: str):
        pass

    def load(self, model: Model, load_dir: str):
        pass


class NullBackend(ModelBackend):
    def _initialize(self, model, spec):
        return model


class ModelInterface(abc.ABC):
    """Algorithm-side handlers of the four model function call types; every method receives the hosted
    `Model`, a `SequenceSample` and the micro-batch spec of the call."""

    def save(self, model: Model, save_dir: str):
        pass

    def evaluate(self, model: Model, eval_dataloader) -> Dict:
        return {}

    def inference(self, model: Model, data, mb_spec):
        raise NotImplementedError()

    def generate(self, model: Model, data, mb_spec):
        raise NotImplementedError()

    def train_step(self, model: Model, data, mb_spec):
        raise NotImplementedError()

    def mock(self, type_: str, model: Model, data):
        """Synthetic input of the right keys for profiling / dry runs of one call type."""
        raise NotImplementedError()


class NullInterface(ModelInterface):
    """Returns nothing for inference, fixed scalars for training: used by system tests and as a stand-in
    for roles that need no model (e.g. a rule-based reward)."""

    def inference(self, model, data, mb_spec):
        from .data_api import SequenceSample

        n = data.bs
        return SequenceSample(keys={"rewards"}, ids=list(data.ids), seqlens=dict(rewards=[[1]] * n),
                              data=dict(rewards=torch.zeros(n)))

    def train_step(self, model, data, mb_spec):
        key = next((k for k in ("packed_input_ids", "packed_prompts") if k in data.keys), None)
        n_tokens = float(sum(sum(x) for x in data.seqlens[key])) if key else 0.0
        model.inc_version()
        return dict(loss=0.0, n_seqs=float(data.bs), n_tokens=n_tokens)

    def generate(self, model, data, mb_spec):
        return data

    def save(self, model, save_dir):
        pass


ALL_MODEL_CLASSES: Dict[str, Any] = {}
ALL_INTERFACE_CLASSES: Dict[str, Any] = {"null": NullInterface}
ALL_BACKEND_CLASSES: Dict[str, Any] = {"null": NullBackend}


def register_model(name: str, fn):
    assert name not in ALL_MODEL_CLASSES, name
    ALL_MODEL_CLASSES[name] = fn


def register_interface(name: str, cls):
    assert name not in ALL_INTERFACE_CLASSES, name
    ALL_INTERFACE_CLASSES[name] = cls


def register_backend(name: str, cls):
    assert name not in ALL_BACKEND_CLASSES, name
    ALL_BACKEND_CLASSES[name] = cls


def _load_impl():
    from ..impl import backend as _b  # noqa: F401  (registers the built-in backends and model builders)
    from ..impl import interface as _i  # noqa: F401


def make_model(cfg: ModelAbstraction, name: ModelName, device) -> Model:
    _load_impl()
    return ALL_MODEL_CLASSES[cfg.type_](name=name, device=device, **cfg.args)


def make_interface(cfg: ModelInterfaceAbstraction) -> ModelInterface:
    _load_impl()
    return ALL_INTERFACE_CLASSES[cfg.type_](**cfg.args)


def make_backend(cfg: ModelBackendAbstraction) -> ModelBackend:
    _load_impl()
    return ALL_BACKEND_CLASSES[cfg.type_](**cfg.args)
