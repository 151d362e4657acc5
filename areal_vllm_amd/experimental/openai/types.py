"""Chat-completion value types and the cached (completion, tokens, reward) record.

The `openai` package is not a dependency: the response objects are plain dataclasses that expose the
attribute surface agent code reads (``completion.id``, ``.choices[0].message.content / .tool_calls``,
``.choices[0].finish_reason``, ``.usage.*``) plus ``model_dump()``. The cached record converts to one
training row exactly like the reference's ``CompletionWithTokenLogpReward.to_tensor_dict``
(``areal/experimental/openai/types.py:11-38``).
"""
import dataclasses
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

import torch

from ...api.io_struct import ModelResponse


class _Dumpable:
    def model_dump(self) -> Dict[str, Any]:
        return dataclasses.asdict(self)

    def __getitem__(self, k):
        return getattr(self, k)


@dataclass
class Function(_Dumpable):
    name: str
    arguments: str


@dataclass
class ChatCompletionMessageFunctionToolCall(_Dumpable):
    id: str
    function: Function
    type: str = "function"


@dataclass
class ChatCompletionMessage(_Dumpable):
    content: Optional[str]
    role: str = "assistant"
    tool_calls: Optional[List[ChatCompletionMessageFunctionToolCall]] = None


@dataclass
class Choice(_Dumpable):
    finish_reason: str
    index: int
    message: ChatCompletionMessage
    logprobs: Optional[Any] = None


@dataclass
class CompletionUsage(_Dumpable):
    completion_tokens: int
    prompt_tokens: int
    total_tokens: int


@dataclass
class ChatCompletion(_Dumpable):
    id: str
    choices: List[Choice]
    created: int
    model: str = "None"
    object: str = "chat.completion"
    service_tier: Optional[str] = None
    system_fingerprint: Optional[str] = None
    usage: Optional[CompletionUsage] = None


@dataclass
class CompletionWithTokenLogpReward:
    completion: ChatCompletion
    response: ModelResponse
    messages: List[dict] = field(default_factory=list)
    reward: Optional[float] = None

    def to_tensor_dict(self) -> Dict[str, torch.Tensor]:
        r = self.response
        seq = r.input_tokens + r.output_tokens
        assert self.reward is not None, "reward was never set for this completion"
        return dict(
            input_ids=torch.tensor(seq).unsqueeze(0),
            loss_mask=torch.tensor([0] * r.input_len + [1] * r.output_len).unsqueeze(0),
            logprobs=torch.tensor([0.0] * r.input_len + r.output_logprobs).unsqueeze(0),
            versions=torch.tensor([-1] * r.input_len + r.output_versions).unsqueeze(0),
            attention_mask=torch.ones(len(seq), dtype=torch.bool).unsqueeze(0),
            rewards=torch.tensor([float(self.reward)]),
        )
