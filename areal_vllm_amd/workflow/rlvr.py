"""Single-turn RL-with-verifiable-reward workflow.

One episode = ``n_samples`` independent generations of the same prompt, each scored by the reward
function in the shared process pool; the result is a padded batch whose rows are the group members
(adjacent, as group normalisation and dynamic sampling require). Sample schema and dump format follow
``areal/workflow/rlvr.py`` in the reference: input_ids, loss_mask (0 on prompt), logprobs (0 on prompt),
versions (-1 on prompt), attention_mask, rewards.
"""
import asyncio
import os
import uuid
from typing import Any, Callable, Dict, List, Optional

import torch

from ..api.cli_args import GenerationHyperparameters
from ..api.io_struct import ModelRequest
from ..api.reward_api import AsyncRewardWrapper
from ..api.workflow_api import RolloutWorkflow
from ..utils import logging, stats_tracker
from ..utils.data import concat_padded_tensors

logger = logging.getLogger("rlvr-workflow")


class RLVRWorkflow(RolloutWorkflow):
    def __init__(self, reward_fn: Callable, gconfig: GenerationHyperparameters, tokenizer=None,
                 enable_thinking: bool = False, rollout_stat_scope: str = "rollout", dump_dir: Optional[str] = None,
                 reward_in_process: bool = False, reward_timeout: float = 15.0):
        self.reward_fn = reward_fn
        self.gconfig = gconfig
        self.tokenizer = tokenizer
        self.enable_thinking = enable_thinking
        self.dump_dir = dump_dir
        self.rollout_stat_scope = rollout_stat_scope
        self.reward_in_process = reward_in_process
        self.async_reward_fn = None if reward_in_process else AsyncRewardWrapper(reward_fn, reward_timeout)
        if self.dump_dir is not None:
            os.makedirs(self.dump_dir, exist_ok=True)

    def _prompt_ids(self, data: Dict[str, Any]) -> List[int]:
        if "input_ids" in data:
            ids = data["input_ids"]
            return ids.tolist() if torch.is_tensor(ids) else list(ids)
        assert self.tokenizer is not None, "a tokenizer is needed for `messages` prompts"
        return self.tokenizer.apply_chat_template(data["messages"], tokenize=True, add_generation_prompt=True,
                                                  enable_thinking=self.enable_thinking)

    async def _reward(self, prompt_str, completion_str, prompt_ids, completion_ids, data) -> float:
        extra = {k: v for k, v in data.items() if k not in ("messages", "input_ids")}
        if self.reward_in_process:
            return float(self.reward_fn(prompt_str, completion_str, prompt_ids, completion_ids, **extra))
        return float(await self.async_reward_fn(prompt_str, completion_str, prompt_ids, completion_ids, **extra))

    async def arun_episode(self, engine, data: Dict[str, Any]):
        input_ids = self._prompt_ids(data)
        n = self.gconfig.n_samples
        req = ModelRequest(rid=uuid.uuid4().hex, input_ids=input_ids, gconfig=self.gconfig.new(n_samples=1),
                           tokenizer=self.tokenizer)
        resps = await asyncio.gather(*[engine.agenerate(req.copy()) for _ in range(n)])
        version = engine.get_version()
        prompt_str = self.tokenizer.decode(input_ids) if self.tokenizer is not None else ""
        results = []
        dump = []
        for resp in resps:
            seq = resp.input_tokens + resp.output_tokens
            completion_str = self.tokenizer.decode(resp.output_tokens) if self.tokenizer is not None else ""
            reward = await self._reward(prompt_str, completion_str, resp.input_tokens, resp.output_tokens, data)
            stats_tracker.get(self.rollout_stat_scope).scalar(reward=reward)
            res = dict(
                input_ids=torch.tensor(seq).unsqueeze(0),
                loss_mask=torch.tensor([0] * resp.input_len + [1] * resp.output_len).unsqueeze(0),
                logprobs=torch.tensor([0.0] * resp.input_len + resp.output_logprobs).unsqueeze(0),
                versions=torch.tensor([-1] * resp.input_len + resp.output_versions).unsqueeze(0),
                attention_mask=torch.ones(len(seq), dtype=torch.bool).unsqueeze(0),
                rewards=torch.tensor([float(reward)]),
            )
            results.append(res)
            dump.append((len(seq), resp.input_len, reward, prompt_str, completion_str))
        if self.dump_dir is not None:
            d = os.path.join(self.dump_dir, str(version))
            os.makedirs(d, exist_ok=True)
            qid = next((str(data[k]) for k in ("query_id", "id", "qid") if k in data), uuid.uuid4().hex)
            with open(os.path.join(d, f"{qid}.txt"), "a") as f:
                for i, (sl, pl, r, p, c) in enumerate(dump):
                    f.write(f"idx: {i + 1} / {n}, seqlen: {sl}, prompt_len: {pl}, reward: {r}\n"
                            f"prompt is \n{p}\nsequence is: \n{c}\n")
        return concat_padded_tensors(results)
