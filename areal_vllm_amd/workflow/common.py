"""Building blocks shared by the rollout workflows.

A trajectory is a token sequence made of CONTEXT segments (prompt, tool output, feedback: not trained on,
loss mask 0, log-prob 0, version -1) and GENERATED segments (the policy's own tokens with their behaviour
log-probs and the weight version that sampled each token). `Transcript` accumulates the segments of one
trajectory and renders the row schema every trainer consumes (``areal/workflow/rlvr.py:67-100`` in the
reference): input_ids, loss_mask, logprobs, versions, attention_mask, rewards - all ``[1, S]`` / ``[1]``.
"""
import os
import uuid
from typing import Any, Callable, Dict, Iterable, List, Optional, Sequence

import torch

from ..api.reward_api import AsyncRewardWrapper
from ..utils import stats_tracker

_PROMPT_KEYS = ("messages", "input_ids", "images")
_ID_KEYS = ("query_id", "id", "qid")


class Transcript:
    def __init__(self):
        self.tokens: List[int] = []
        self.mask: List[int] = []
        self.logprobs: List[float] = []
        self.versions: List[int] = []

    def __len__(self) -> int:
        return len(self.tokens)

    def context(self, tokens: Sequence[int]) -> "Transcript":
        n = len(tokens)
        self.tokens.extend(int(t) for t in tokens)
        self.mask.extend([0] * n)
        self.logprobs.extend([0.0] * n)
        self.versions.extend([-1] * n)
        return self

    def generated(self, tokens: Sequence[int], logprobs: Sequence[float], versions: Sequence[int]) -> "Transcript":
        assert len(tokens) == len(logprobs) == len(versions)
        self.tokens.extend(int(t) for t in tokens)
        self.mask.extend([1] * len(tokens))
        self.logprobs.extend(float(x) for x in logprobs)
        self.versions.extend(int(v) for v in versions)
        return self

    def extend_to(self, prefix: Sequence[int]) -> "Transcript":
        """Append, as context, whatever `prefix` holds beyond this transcript (the next turn's prompt must
        continue the episode token for token)."""
        have = len(self.tokens)
        if list(prefix[:have]) != self.tokens:
            raise ValueError("turn prompt does not extend the episode")
        return self.context(prefix[have:])

    @classmethod
    def of_response(cls, resp) -> "Transcript":
        return cls().context(resp.input_tokens).generated(resp.output_tokens, resp.output_logprobs, resp.output_versions)

    def row(self, reward: float, **extra) -> Dict[str, Any]:
        S = len(self.tokens)
        out = dict(input_ids=torch.tensor(self.tokens, dtype=torch.long).view(1, S),
                   loss_mask=torch.tensor(self.mask, dtype=torch.long).view(1, S),
                   logprobs=torch.tensor(self.logprobs, dtype=torch.float32).view(1, S),
                   versions=torch.tensor(self.versions, dtype=torch.long).view(1, S),
                   attention_mask=torch.ones(1, S, dtype=torch.bool),
                   rewards=torch.tensor([float(reward)], dtype=torch.float32))
        out.update(extra)
        return out


class Scorer:
    """Reward function behind one call signature: inline (cheap, deterministic rewards) or through the shared
    process pool with a timeout (``api/reward_api.py``). Dataset fields other than the prompt travel as
    keyword arguments."""

    def __init__(self, reward_fn: Callable, in_process: bool = False, timeout: float = 15.0):
        self.fn = reward_fn
        self.pooled = None if in_process else AsyncRewardWrapper(reward_fn, timeout)

    async def __call__(self, prompt: str, completion: str, prompt_ids, completion_ids, fields: Dict[str, Any]) -> float:
        kw = {k: v for k, v in fields.items() if k not in _PROMPT_KEYS}
        if self.pooled is None:
            return float(self.fn(prompt, completion, prompt_ids, completion_ids, **kw))
        return float(await self.pooled(prompt, completion, prompt_ids, completion_ids, **kw))


class RolloutDump:
    """Text dumps of sampled groups: ``<dir>/<weight version>/<query id>.txt`` (the layout the reference's
    workflows write, ``areal/workflow/rlvr.py``)."""

    def __init__(self, directory: Optional[str]):
        self.dir = directory
        if directory is not None:
            os.makedirs(directory, exist_ok=True)

    def write(self, version: int, data: Dict[str, Any], entries: Iterable[Dict[str, Any]]):
        if self.dir is None:
            return
        entries = list(entries)
        folder = os.path.join(self.dir, str(version))
        os.makedirs(folder, exist_ok=True)
        qid = next((str(data[k]) for k in _ID_KEYS if k in data), None) or uuid.uuid4().hex
        with open(os.path.join(folder, f"{qid}.txt"), "a") as f:
            for i, e in enumerate(entries, 1):
                head = f"idx: {i} / {len(entries)}, seqlen: {e['seqlen']}"
                if "prompt_len" in e:
                    head += f", prompt_len: {e['prompt_len']}"
                f.write(f"{head}, reward: {e['reward']}\nprompt is \n{e['prompt']}\nsequence is: \n{e['completion']}\n")


def prompt_token_ids(data: Dict[str, Any], tokenizer, enable_thinking: bool = False) -> List[int]:
    """Token ids of the episode's prompt: given directly, or rendered from chat messages."""
    ids = data.get("input_ids")
    if ids is not None:
        return ids.tolist() if torch.is_tensor(ids) else [int(t) for t in ids]
    if tokenizer is None:
        raise ValueError("a tokenizer is needed for `messages` prompts")
    return list(tokenizer.apply_chat_template(data["messages"], tokenize=True, add_generation_prompt=True,
                                              enable_thinking=enable_thinking))


def detok(tokenizer, ids) -> str:
    return tokenizer.decode(ids) if tokenizer is not None else ""


def record(scope: str, **scalars):
    stats_tracker.get(scope).scalar(**scalars)
