"""Tool-using agent episodes (search agents and similar) on top of the ``Environment`` API.

One episode interleaves model turns and tool observations in a single growing token sequence:

    prompt | model turn 1 (may call a tool) | observation 1 | model turn 2 | ... | final answer

Model tokens carry loss mask 1, their behaviour log-probs and the weight version that sampled them;
prompt and observation tokens carry loss mask 0 / version -1, exactly the schema of the single-turn
workflow, so the PPO actor trains on the model's own tokens only. Every turn re-uses the same request id
(prefix-cache affinity on the rollout server) and generation stays interruptible: a weight update in the
middle of an episode only changes the version tags of later tokens.

The reference configures such a run in ``examples/search-agent/local_1.5b_example.yaml`` (``max_turns``,
``log_agent_stats_keys``, n_samples 16, staleness 4); its agent code lives in an external repository, so
the episode logic here follows the interfaces it relies on: ``areal/api/env_api.py`` (tools),
``areal/experimental/openai/tool_call_parser.py`` (call syntax) and the multi-turn token bookkeeping of
``areal/workflow/multi_turn.py:61-136``.
"""
import asyncio
import json
import uuid
from typing import Any, Callable, Dict, List, Optional

import torch

from ..api.cli_args import GenerationHyperparameters
from ..api.env_api import Environment
from ..api.io_struct import ModelRequest
from ..api.workflow_api import RolloutWorkflow
from ..experimental.openai.tool_call_parser import process_tool_calls
from ..utils import stats_tracker
from ..utils.data import concat_padded_tensors


class ToolAgentWorkflow(RolloutWorkflow):
    def __init__(self, reward_fn: Callable, gconfig: GenerationHyperparameters, tokenizer,
                 env_factory: Callable[[], Environment], max_turns: int = 8, tool_call_parser: str = "qwen25",
                 max_observation_tokens: int = 512, max_total_tokens: Optional[int] = None,
                 rollout_stat_scope: str = "rollout", agent_stats_keys: Optional[List[str]] = None,
                 observation_template: str = "\n<tool_response>\n{observation}\n</tool_response>\n"):
        self.reward_fn = reward_fn
        self.gconfig = gconfig
        self.tokenizer = tokenizer
        self.env_factory = env_factory
        self.max_turns = max_turns
        self.tool_call_parser = tool_call_parser
        self.max_observation_tokens = max_observation_tokens
        self.max_total_tokens = max_total_tokens
        self.rollout_stat_scope = rollout_stat_scope
        self.agent_stats_keys = list(agent_stats_keys or [])
        self.observation_template = observation_template

    def _prompt_ids(self, data: Dict[str, Any], tools) -> List[int]:
        if "input_ids" in data:
            return list(data["input_ids"])
        return list(self.tokenizer.apply_chat_template(data["messages"], tools=tools or None, tokenize=True,
                                                       add_generation_prompt=True))

    async def _run_one_episode(self, engine, data: Dict[str, Any], rid: str):
        env = self.env_factory()
        await env.ainitialize()
        tools = env.list_tools()
        stats = dict(num_llm_gens=0, num_tool_calls=0, num_failed_tool_calls=0, num_input_tokens=0,
                     num_output_tokens=0)
        try:
            seq = self._prompt_ids(data, tools)
            logprobs, loss_mask, versions = [0.0] * len(seq), [0] * len(seq), [-1] * len(seq)
            stats["num_input_tokens"] = len(seq)
            answer_text, budget_left = "", self.gconfig.max_new_tokens
            for turn in range(self.max_turns):
                if self.max_total_tokens is not None and len(seq) >= self.max_total_tokens:
                    break
                g = self.gconfig.new(n_samples=1, max_new_tokens=min(
                    budget_left, self.max_total_tokens - len(seq) if self.max_total_tokens else budget_left))
                resp = await engine.agenerate(ModelRequest(rid=rid, input_ids=list(seq), gconfig=g,
                                                           tokenizer=self.tokenizer))
                stats["num_llm_gens"] += 1
                stats["num_output_tokens"] += resp.output_len
                seq += resp.output_tokens
                logprobs += resp.output_logprobs
                loss_mask += [1] * resp.output_len
                versions += resp.output_versions
                text = self.tokenizer.decode(resp.output_tokens)
                calls, rest, reason = process_tool_calls(text, tools, self.tool_call_parser, resp.stop_reason)
                if reason != "tool_calls" or not calls:
                    answer_text = text
                    break
                obs_parts = []
                for c in calls:
                    stats["num_tool_calls"] += 1
                    try:
                        args = json.loads(c.function.arguments) if c.function.arguments else {}
                        obs_parts.append(str(await env.aexecute(c.function.name, args)))
                    except Exception as e:  # a broken call is an observation too: the policy must learn from it
                        stats["num_failed_tool_calls"] += 1
                        obs_parts.append(f"tool error: {e!r}")
                obs_ids = list(self.tokenizer.encode(self.observation_template.format(observation="\n".join(obs_parts))))
                obs_ids = obs_ids[: self.max_observation_tokens]
                seq += obs_ids
                logprobs += [0.0] * len(obs_ids)
                loss_mask += [0] * len(obs_ids)
                versions += [-1] * len(obs_ids)
                answer_text = text
            extra = {k: v for k, v in data.items() if k not in ("messages", "input_ids")}
            n_prompt = stats["num_input_tokens"]
            reward = float(self.reward_fn(self.tokenizer.decode(seq[:n_prompt]), answer_text, seq[:n_prompt],
                                          seq[n_prompt:], **extra))
        finally:
            await env.aclose()
        stats["score"] = reward
        for k, v in getattr(env, "stats", {}).items():
            stats[k] = v
        tracker = stats_tracker.get(self.rollout_stat_scope)
        tracker.scalar(reward=reward, num_turns=stats["num_llm_gens"])
        # every configured key is present on every row (0 when the episode never touched it): the actor
        # logs them over the rows flagged `begin_of_trajectory`
        keep = {k: float(stats.get(k, 0.0)) for k in self.agent_stats_keys}
        if keep:
            tracker.scalar(**keep)
        out = dict(
            input_ids=torch.tensor(seq).unsqueeze(0), logprobs=torch.tensor(logprobs).unsqueeze(0),
            loss_mask=torch.tensor(loss_mask).unsqueeze(0), versions=torch.tensor(versions).unsqueeze(0),
            rewards=torch.tensor([reward]), attention_mask=torch.ones(len(seq), dtype=torch.bool).unsqueeze(0))
        out["begin_of_trajectory"] = torch.tensor([1])
        for k, v in keep.items():  # per-trajectory agent statistics travel with the batch (log_agent_stats)
            out[k] = torch.tensor([v])
        return out

    async def arun_episode(self, engine, data: Dict[str, Any]):
        rid = uuid.uuid4().hex
        results = await asyncio.gather(*[self._run_one_episode(engine, data, f"{rid}-{i}")
                                         for i in range(self.gconfig.n_samples)])
        return concat_padded_tensors(results)
