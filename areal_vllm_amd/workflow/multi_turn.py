"""Multi-turn self-correction workflow: retry with a fixed feedback message until the reward is
positive or ``max_turns`` is reached; the final reward is discounted per extra turn.

The concatenated sequence keeps prompt and feedback tokens with loss mask 0 / version -1 and the model's
own tokens with loss mask 1 and their behaviour log-probs; feedback is appended as TOKEN IDS computed
once (difference of two chat-template renderings), which avoids decode/encode drift. Behaviour of
``areal/workflow/multi_turn.py:23-173``.
"""
import asyncio
import os
import uuid
from typing import Any, Dict, Optional

import torch

from ..api.cli_args import GenerationHyperparameters
from ..api.io_struct import ModelRequest
from ..api.reward_api import AsyncRewardWrapper
from ..api.workflow_api import RolloutWorkflow
from ..utils import logging, stats_tracker
from ..utils.data import concat_padded_tensors

logger = logging.getLogger("multi-turn-workflow")

FEEDBACK = ("Your answer is either wrong or not parsable to the reward function. You may misunderstand the original "
            "question. Please carefully read the original question, check the previous errors, and try to answer "
            "it again.")


class MultiTurnWorkflow(RolloutWorkflow):
    def __init__(self, reward_fn, gconfig: GenerationHyperparameters, tokenizer, max_turns: int, turn_discount: float,
                 rollout_stat_scope: str = "rollout", dump_dir: Optional[str] = None, reward_in_process: bool = False,
                 feedback: str = FEEDBACK):
        self.reward_fn = reward_fn
        self.gconfig = gconfig
        self.tokenizer = tokenizer
        self.max_turns = max_turns
        self.turn_discount = turn_discount
        self.rollout_stat_scope = rollout_stat_scope
        self.reward_in_process = reward_in_process
        self.async_reward_fn = None if reward_in_process else AsyncRewardWrapper(reward_fn)
        self.dump_dir = dump_dir
        if dump_dir is not None:
            os.makedirs(dump_dir, exist_ok=True)
        base = [{"role": "assistant", "content": "some random message."}]
        s1 = tokenizer.apply_chat_template(base, tokenize=True)
        s2 = tokenizer.apply_chat_template(base + [{"role": "user", "content": feedback}], tokenize=True,
                                           add_generation_prompt=True)
        self.multi_turn_prompt_ids = list(s2[len(s1):])

    async def _reward(self, prompt_str, completion_str, prompt_ids, completion_ids, data) -> float:
        extra = {k: v for k, v in data.items() if k not in ("messages", "input_ids")}
        if self.reward_in_process:
            return float(self.reward_fn(prompt_str, completion_str, prompt_ids, completion_ids, **extra))
        return float(await self.async_reward_fn(prompt_str, completion_str, prompt_ids, completion_ids, **extra))

    async def _run_one_episode(self, engine, data: Dict[str, Any], rid: str):
        seq, logprobs, loss_mask, versions = [], [], [], []
        if "input_ids" in data:
            input_ids = list(data["input_ids"])
        else:
            input_ids = list(self.tokenizer.apply_chat_template(data["messages"], tokenize=True,
                                                                add_generation_prompt=True))
        t, reward, discount = 0, 0.0, 1.0
        prompt_str = completion_str = ""
        while reward == 0 and t < self.max_turns:
            req = ModelRequest(rid=rid, input_ids=input_ids, gconfig=self.gconfig.new(n_samples=1),
                               tokenizer=self.tokenizer)
            resp = await engine.agenerate(req)
            prompt_str = self.tokenizer.decode(input_ids)
            completion_str = self.tokenizer.decode(resp.output_tokens)
            reward = await self._reward(prompt_str, completion_str, resp.input_tokens, resp.output_tokens, data)
            new_in = len(resp.input_tokens) - len(seq)
            assert len(seq) == 0 or resp.input_tokens[:len(seq)] == seq, "turn prompt does not extend the episode"
            seq += resp.input_tokens[len(resp.input_tokens) - new_in:] + resp.output_tokens
            logprobs += [0.0] * new_in + resp.output_logprobs
            loss_mask += [0] * new_in + [1] * resp.output_len
            versions += [-1] * new_in + resp.output_versions
            t += 1
            if reward == 0 and t < self.max_turns:
                input_ids = input_ids + resp.output_tokens
                if not resp.output_tokens or resp.output_tokens[-1] != self.tokenizer.eos_token_id:
                    input_ids = input_ids + [self.tokenizer.eos_token_id]
                input_ids = input_ids + self.multi_turn_prompt_ids
                discount *= self.turn_discount
        reward = float(reward * discount)
        stats_tracker.get(self.rollout_stat_scope).scalar(reward=reward, num_turns=t)
        res = dict(
            input_ids=torch.tensor(seq).unsqueeze(0),
            logprobs=torch.tensor(logprobs).unsqueeze(0),
            loss_mask=torch.tensor(loss_mask).unsqueeze(0),
            versions=torch.tensor(versions).unsqueeze(0),
            rewards=torch.tensor([reward]),
            attention_mask=torch.ones(len(seq), dtype=torch.bool).unsqueeze(0),
        )
        return res, prompt_str, completion_str, reward, len(seq)

    async def arun_episode(self, engine, data: Dict[str, Any]):
        rid = uuid.uuid4().hex
        results = await asyncio.gather(*[self._run_one_episode(engine, data, f"{rid}-{i}")
                                         for i in range(self.gconfig.n_samples)])
        if self.dump_dir is not None:
            d = os.path.join(self.dump_dir, str(engine.get_version()))
            os.makedirs(d, exist_ok=True)
            qid = next((str(data[k]) for k in ("query_id", "id", "qid") if k in data), uuid.uuid4().hex)
            with open(os.path.join(d, f"{qid}.txt"), "a") as f:
                for i, (_, p, c, r, sl) in enumerate(results):
                    f.write(f"idx: {i + 1} / {len(results)}, seqlen: {sl}, reward is {r}.\nprompt is \n{p}\n"
                            f"sequence is: \n{c}\n")
        return concat_padded_tensors([r[0] for r in results])
