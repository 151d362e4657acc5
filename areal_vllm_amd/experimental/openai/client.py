"""OpenAI-compatible async chat client backed by a rollout engine, with token-level bookkeeping.

``client.chat.completions.create(messages=..., tools=..., ...)`` renders the chat template, calls
``engine.agenerate`` and returns a ChatCompletion; every stored completion keeps its token ids,
behaviour log-probs and weight versions so that an agent written against the OpenAI API produces
trainable trajectories. ``set_reward`` attaches rewards, ``export_completions(turn_discount)``
back-propagates them through the conversation tree (a completion whose message list is a role-prefix of
another is its parent; a parent receives ``turn_discount * mean(children rewards)`` on top of its own).
Semantics of ``areal/experimental/openai/client.py:42-342``.
"""
import time
import uuid
from collections import defaultdict
from copy import deepcopy
from typing import Any, Dict, Iterable, List, Optional, Tuple, Union

from ...api.cli_args import GenerationHyperparameters
from ...api.io_struct import ModelRequest
from .tool_call_parser import process_tool_calls
from .types import (
    ChatCompletion,
    ChatCompletionMessage,
    Choice,
    CompletionUsage,
    CompletionWithTokenLogpReward,
)


class _NotGiven:
    def __bool__(self):
        return False

    def __repr__(self):
        return "NOT_GIVEN"


NOT_GIVEN = _NotGiven()


class AsyncCompletionsWithReward:
    def __init__(self, client, engine, tokenizer, cache: Dict[str, CompletionWithTokenLogpReward],
                 tool_call_parser: Optional[str] = None):
        self._client = client
        self.engine = engine
        self.tokenizer = tokenizer
        self.tool_call_parser = tool_call_parser
        self._cache = cache

    async def create(self, *, messages: Iterable[dict], frequency_penalty=NOT_GIVEN, max_completion_tokens=NOT_GIVEN,
                     max_tokens=NOT_GIVEN, metadata=NOT_GIVEN, stop=NOT_GIVEN, store=NOT_GIVEN, temperature=NOT_GIVEN,
                     tool_choice=NOT_GIVEN, tools=NOT_GIVEN, top_p=NOT_GIVEN, extra_body: Optional[dict] = None,
                     **ignored) -> ChatCompletion:
        messages_list = list(messages)
        if not messages_list:
            raise ValueError("messages cannot be empty")
        extra_body = extra_body or {}
        tools = tools if tools is not NOT_GIVEN else None
        kw = dict(extra_body.get("chat_template_kwargs", {}))
        if tools:
            kw["tools"] = tools
        prompt_ids = list(self.tokenizer.apply_chat_template(messages_list, add_generation_prompt=True, tokenize=True,
                                                             **kw))
        temp = 1.0 if temperature is NOT_GIVEN else (temperature or 0.0)
        max_new = 512
        if max_tokens is not NOT_GIVEN and max_tokens is not None:
            max_new = max_tokens - len(prompt_ids)
            if max_new <= 0:
                raise RuntimeError("max_tokens must be greater than the number of prompt tokens")
        if max_completion_tokens is not NOT_GIVEN and max_completion_tokens is not None:
            max_new = min(max_new, max_completion_tokens)
        stop_v = None if stop is NOT_GIVEN else stop
        stop_ids = {t for t in (self.tokenizer.eos_token_id, getattr(self.tokenizer, "pad_token_id", None))
                    if t is not None}
        gconfig = GenerationHyperparameters(
            n_samples=1, temperature=temp, max_new_tokens=max_new,
            top_p=1.0 if top_p is NOT_GIVEN else (top_p or 1.0),
            stop=stop_v if isinstance(stop_v, list) else ([stop_v] if stop_v else None), greedy=temp == 0,
            frequency_penalty=0.0 if frequency_penalty is NOT_GIVEN or frequency_penalty is None else frequency_penalty,
            stop_token_ids=sorted(stop_ids))
        req = ModelRequest(input_ids=prompt_ids, gconfig=gconfig, rid=str(uuid.uuid4()),
                           metadata={} if metadata is NOT_GIVEN or metadata is None else dict(metadata),
                           tokenizer=self.tokenizer)
        resp = await self.engine.agenerate(req)
        text = self.tokenizer.decode(resp.output_tokens)
        tool_calls = None
        if tool_choice != "none" and tools:
            tool_calls, text, resp.stop_reason = process_tool_calls(text, tools, self.tool_call_parser,
                                                                    resp.stop_reason)
        cid = f"chatcmpl-{uuid.uuid4().hex[:29]}"
        completion = ChatCompletion(
            id=cid, created=int(time.time()),
            choices=[Choice(finish_reason=resp.stop_reason, index=0,
                            message=ChatCompletionMessage(content=text, role="assistant", tool_calls=tool_calls))],
            usage=CompletionUsage(completion_tokens=len(resp.output_tokens), prompt_tokens=len(resp.input_tokens),
                                  total_tokens=len(resp.input_tokens) + len(resp.output_tokens)))
        if store is NOT_GIVEN or store:
            self._cache[cid] = CompletionWithTokenLogpReward(completion=deepcopy(completion), response=resp,
                                                            messages=deepcopy(messages_list))
        return completion


class _Chat:
    def __init__(self, completions):
        self.completions = completions


class ArealOpenAI:
    def __init__(self, engine, tokenizer, tool_call_parser: Optional[str] = None, **kwargs):
        self.engine = engine
        self.tokenizer = tokenizer
        self.tool_call_parser = tool_call_parser
        self._completion_cache: Dict[str, CompletionWithTokenLogpReward] = {}
        self.chat = _Chat(AsyncCompletionsWithReward(self, engine, tokenizer, self._completion_cache,
                                                     tool_call_parser=tool_call_parser))

    def get_completions(self, completion_id: str) -> Optional[CompletionWithTokenLogpReward]:
        return self._completion_cache.get(completion_id)

    def set_reward(self, completion_id: str, reward: float) -> None:
        if completion_id not in self._completion_cache:
            raise KeyError(f"Completion with ID {completion_id} not found in cache")
        self._completion_cache[completion_id].reward = reward

    def export_completions(self, turn_discount: float = 1.0) -> Dict[str, CompletionWithTokenLogpReward]:
        cache = self._completion_cache
        roles = {cid: tuple(m.get("role", "user") for m in c.messages) for cid, c in cache.items()}
        # parent = the completion with the longest role sequence that is a proper prefix of mine
        parent: Dict[str, str] = {}
        for cid, r in roles.items():
            best = None
            for pid, pr in roles.items():
                if pid != cid and len(pr) < len(r) and r[:len(pr)] == pr:
                    if best is None or len(pr) >= len(roles[best]):
                        best = pid
            if best is not None:
                parent[cid] = best
        children = defaultdict(list)
        for cid, pid in parent.items():
            children[pid].append(cid)
        # leaves first: deeper conversations have longer role sequences
        for cid in sorted(cache, key=lambda c: -len(roles[c])):
            node = cache[cid]
            if node.reward is None:
                node.reward = 0.0
            kids = [cache[k].reward for k in children[cid] if cache[k].reward is not None]
            if kids:
                node.reward += turn_discount * sum(kids) / len(kids)
        return dict(cache)
