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
