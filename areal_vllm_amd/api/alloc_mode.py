"""Allocation-mode DSL: how GPUs are split between rollout engines and trainers.

Accepted expressions (same language as the reference's Lark grammar, ``areal/api/alloc_mode.py:324-366``;
parsed here by a small hand-written recursive-descent parser, no parser generator needed):

    sglang:d4t2+fsdp:d8           decoupled: 4x(tp=2) rollout replicas + 8-way DP trainer
    vllm:d2t4|megatron:d2t4p2     colocated (same world size on both sides)
    native:d4+native:d4           the native engine/trainer names are accepted as well
    sglang:d4t2+eval / +cpu       rollout + evaluation processes
    sglang.d4p1t1+d4p1t1          legacy dotted form
    d2p2t1, fsdp:d4c2, megatron:d2t4p2e2      training only
    megatron:(attn:d4t2c1|ffn:d2e4t1)         MoE hybrid: attention and expert layouts differ

Dimensions: d data, t tensor, p pipeline, c context (Ulysses in this framework), e expert.
Backend auto-selection: pipeline or expert parallelism selects the "megatron"-class backend
(``areal/api/alloc_mode.py:421-431``); both names map to the native trainer here.
"""
import enum
import math
import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple


class AllocationType(enum.Enum):
    COLOCATE = 0
    DECOUPLED_TRAIN = 1
    LLM_SERVER_ONLY = 2
    DECOUPLED_EVAL = 3


class AllocationValidationError(Exception):
    """A syntactically valid expression violates a consistency rule."""


class InvalidAllocationModeError(Exception):
    """Kept for callers that catch the reference's legacy exception type."""


INFER_BACKENDS = ("sglang", "vllm", "native")
TRAIN_BACKENDS = ("fsdp", "megatron", "native")


@dataclass
class ParallelStrategy:
    tensor_parallel_size: int = 1
    pipeline_parallel_size: int = 1
    data_parallel_size: int = 1
    context_parallel_size: int = 1
    expert_parallel_size: int = 1
    expert_tensor_parallel_size: Optional[int] = None

    def __post_init__(self):
        if self.expert_parallel_size > 1:
            if self.expert_tensor_parallel_size is None:
                self.expert_tensor_parallel_size = self.tensor_parallel_size
            self.expert_model_parallel_size = (
                self.pipeline_parallel_size * self.expert_tensor_parallel_size * self.expert_parallel_size
            )
            assert self.world_size % self.expert_model_parallel_size == 0, (
                f"Expert model parallel size {self.expert_model_parallel_size} cannot divide "
                f"world size {self.world_size}."
            )

    @property
    def expert_data_parallel_size(self) -> int:
        if not hasattr(self, "expert_model_parallel_size"):
            return self.data_parallel_size
        return self.world_size // self.expert_model_parallel_size

    tp_size = property(lambda self: self.tensor_parallel_size)
    pp_size = property(lambda self: self.pipeline_parallel_size)
    dp_size = property(lambda self: self.data_parallel_size)
    cp_size = property(lambda self: self.context_parallel_size)
    ep_size = property(lambda self: self.expert_parallel_size)
    etp_size = property(lambda self: self.expert_tensor_parallel_size)
    edp_size = property(lambda self: self.expert_data_parallel_size)

    @property
    def world_size(self) -> int:
        return (self.data_parallel_size * self.context_parallel_size * self.tensor_parallel_size
                * self.pipeline_parallel_size)

    def __str__(self):
        parts = [f"tp={self.tp_size}", f"pp={self.pp_size}", f"dp={self.dp_size}"]
        if self.cp_size > 1:
            parts.append(f"cp={self.cp_size}")
        if self.ep_size > 1:
            parts.append(f"ep={self.ep_size}")
            if self.etp_size != self.tp_size:
                parts.append(f"ep_tp={self.etp_size}")
        return f"Parallel({','.join(parts)})"

    @staticmethod
    def parallelism_eq(this, other) -> bool:
        return (this.tp_size == other.tp_size and this.pp_size == other.pp_size and this.dp_size == other.dp_size
                and this.cp_size == other.cp_size and this.ep_size == other.ep_size
                and this.etp_size == other.etp_size)


@dataclass
class FSDPParallelStrategy(ParallelStrategy):
    @staticmethod
    def parallelism_eq(this, other):
        return ParallelStrategy.parallelism_eq(this, other)


@dataclass
class MegatronParallelStrategy(ParallelStrategy):
    virtual_pipeline_parallel_size: Optional[int] = None
    use_sequence_parallel: bool = False

    @staticmethod
    def parallelism_eq(this, other):
        return ParallelStrategy.parallelism_eq(this, other) and (
            this.virtual_pipeline_parallel_size == other.virtual_pipeline_parallel_size)


_DIM_FIELD = {"d": "data_parallel_size", "t": "tensor_parallel_size", "p": "pipeline_parallel_size",
              "c": "context_parallel_size", "e": "expert_parallel_size"}


class _Parser:
    """Recursive descent over the allocation grammar. Whitespace is insignificant."""

    def __init__(self, text: str):
        self.s = re.sub(r"\s+", "", text)
        self.i = 0

    def error(self, msg):
        raise ValueError(f"Parsing error: {msg} at position {self.i} of '{self.s}'")

    def peek(self, n=1):
        return self.s[self.i:self.i + n]

    def eat(self, lit: str) -> bool:
        if self.s.startswith(lit, self.i):
            self.i += len(lit)
            return True
        return False

    def word(self, choices) -> Optional[str]:
        for c in sorted(choices, key=len, reverse=True):
            if self.s.startswith(c, self.i):
                self.i += len(c)
                return c
        return None

    def dims(self, allowed: str) -> Dict[str, int]:
        out: Dict[str, int] = {}
        # "m" (model parallel) is the worker runtime's older spelling of "t": "sglang.d4p1m1+d2p2m1"
        # (``realhf/experiments/common/utils.py:281-314``)
        while self.i < len(self.s) and (self.peek() in allowed or (self.peek() == "m" and "t" in allowed)):
            d = "t" if self.peek() == "m" else self.peek()
            save = self.i
            self.i += 1
            m = re.match(r"[1-9][0-9]*", self.s[self.i:])
            if not m:
                self.i = save
                break
            self.i += m.end()
            if d in out:
                self.error(f"duplicate dimension '{d}'")
            out[d] = int(m.group())
        if not out:
            self.error(f"expected one of the dimensions '{allowed}'")
        return out

    # inf_para: BACKEND (":" | ".") inf_dim+
    def inf_para(self):
        save = self.i
        backend = self.word(INFER_BACKENDS)
        if backend is None or not (self.eat(":") or self.eat(".")):
            self.i = save
            return None
        d = self.dims("dtp")
        return backend, ParallelStrategy(**{_DIM_FIELD[k]: v for k, v in d.items()})

    def hybrid(self):
        paren = self.eat("(")
        if not self.eat("attn:"):
            self.error("expected 'attn:'")
        attn = self.dims("cdtp")
        if not self.eat("|"):
            self.error("expected '|' between attn and ffn sections")
        if not self.eat("ffn:"):
            self.error("expected 'ffn:'")
        ffn = self.dims("detp")
        if paren and not self.eat(")"):
            self.error("expected ')'")
        a = {"d": 1, "t": 1, "p": 1, "c": 1}
        a.update(attn)
        if "p" in ffn and ffn["p"] != a["p"]:
            raise AllocationValidationError(
                "Pipeline parallel size for attention and FFN modules must be identical. "
                f"Got attention: {a['p']}, FFN: {ffn['p']}.")
        attn_ws = a["d"] * a["t"] * a["p"] * a["c"]
        ffn_ws = ffn.get("d", a["d"]) * ffn.get("p", a["p"]) * ffn.get("t", a["t"]) * ffn.get("e", 1)
        if attn_ws != ffn_ws:
            raise InvalidAllocationModeError(
                "World size for expert modules and attention modules must be identical. "
                f"Got attention: {attn_ws}, expert: {ffn_ws}.")
        kw = dict(data_parallel_size=a["d"], tensor_parallel_size=a["t"], pipeline_parallel_size=a["p"],
                  context_parallel_size=a["c"], expert_parallel_size=ffn.get("e", 1))
        if "t" in ffn:
            kw["expert_tensor_parallel_size"] = ffn["t"]
        return ParallelStrategy(**kw)

    # train_para: (BACKEND ":")? (common_dim+ | hybrid)
    def train_para(self):
        save = self.i
        backend = self.word(TRAIN_BACKENDS)
        if backend is not None and not self.eat(":"):
            self.i = save
            backend = None
        if self.peek() == "(" or self.s.startswith("attn:", self.i):
            strat = self.hybrid()
            backend = backend or "megatron"
        else:
            d = self.dims("pdtce")
            strat = ParallelStrategy(**{_DIM_FIELD[k]: v for k, v in d.items()})
        if backend is None:
            backend = "megatron" if (strat.pp_size > 1 or strat.ep_size > 1) else "fsdp"
        return backend, strat

    def parse(self) -> "AllocationMode":
        inf = self.inf_para()
        if inf is None:
            backend, strat = self.train_para()
            self.finish()
            return AllocationMode(AllocationType.COLOCATE, strat, strat, None, backend)
        gb, gs = inf
        if self.i == len(self.s):
            return AllocationMode(AllocationType.LLM_SERVER_ONLY, gs, None, gb, None)
        if self.eat("+"):
            ev = self.word(("cpu", "eval"))
            if ev is not None and self.i == len(self.s):
                return AllocationMode(AllocationType.DECOUPLED_EVAL, gs, None, gb, None)
            if ev is not None:
                self.i -= len(ev)
            tb, ts = self.train_para()
            self.finish()
            return AllocationMode(AllocationType.DECOUPLED_TRAIN, gs, ts, gb, tb)
        if self.eat("|"):
            tb, ts = self.train_para()
            self.finish()
            if gs.world_size != ts.world_size:
                raise AllocationValidationError(
                    "World sizes must match for colocated expressions. "
                    f"Inference world size: {gs.world_size}, Training world size: {ts.world_size}")
            return AllocationMode(AllocationType.COLOCATE, gs, ts, gb, tb)
        self.error("unexpected trailing input")

    def finish(self):
        if self.i != len(self.s):
            self.error("unexpected trailing input")


@dataclass
class AllocationMode:
    type_: AllocationType
    gen: Optional[ParallelStrategy] = None
    train: Optional[ParallelStrategy] = None
    gen_backend: Optional[str] = None
    train_backend: Optional[str] = None

    @property
    def gen_instance_size(self) -> int:
        return self.gen.tp_size * self.gen.pp_size

    @property
    def gen_world_size(self) -> int:
        return self.gen.world_size if self.gen is not None and self.gen_backend is not None else 0

    @property
    def train_world_size(self) -> int:
        return self.train.world_size if self.train is not None else 0

    @classmethod
    def from_str(cls, allocation_mode: str) -> "AllocationMode":
        if not allocation_mode or not allocation_mode.strip():
            raise ValueError("Parsing error: empty allocation mode")
        return _Parser(allocation_mode).parse()
