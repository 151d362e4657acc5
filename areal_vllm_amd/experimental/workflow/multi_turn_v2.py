"""Multi-turn self-correction written against the OpenAI-style client.

Each retry appends the assistant answer and a reflection message to the conversation and issues a new
chat completion through ``ArealOpenAI``; only the final completion carries the (turn-discounted) reward
and the episode returns the client's exported completions (one training row each). Behaviour of
``areal/experimental/workflow/multi_turn_v2.py:21-143``.
"""
import asyncio
import os
import uuid
from copy import deepcopy
from typing import Optional

from ...api.cli_args import GenerationHyperparameters
from ...api.reward_api import AsyncRewardWrapper
from ...api.workflow_api import RolloutWorkflow
from ...utils import stats_tracker
from ...workflow.multi_turn import FEEDBACK
from ..openai import ArealOpenAI


class MultiTurnWorkflow(RolloutWorkflow):
    def __init__(self, reward_fn, gconfig: GenerationHyperparameters, tokenizer, max_turns: int, turn_discount: float,
                 rollout_stat_scope: str = "rollout", dump_dir: Optional[str] = None, reward_in_process: bool = False):
        self.reward_fn = reward_fn
        self.gconfig = gconfig
        self.tokenizer = tokenizer
        self.max_turns = max_turns
        self.turn_discount = turn_discount
        self.reward_in_process = reward_in_process
        self.async_reward_fn = None if reward_in_process else AsyncRewardWrapper(reward_fn)
        self.dump_dir = dump_dir
        self.rollout_stat_scope = rollout_stat_scope
        if dump_dir is not None:
            os.makedirs(dump_dir, exist_ok=True)
        self.reflection_msg = [{"role": "user", "content": FEEDBACK}]

    async def _run_one_episode(self, engine, data, rid):
        client = ArealOpenAI(engine=engine, tokenizer=self.tokenizer)
        messages = deepcopy(data["messages"])
        extra = {k: v for k, v in data.items() if k != "messages"}
        t, reward, discount = 0, 0.0, 1.0
        comp = _comp = None
        while reward == 0 and t < self.max_turns:
            _comp = await client.chat.completions.create(
                messages=messages, frequency_penalty=self.gconfig.frequency_penalty,
                max_completion_tokens=self.gconfig.max_new_tokens, stop=self.gconfig.stop, store=True,
                temperature=self.gconfig.temperature, top_p=self.gconfig.top_p)
            comp = client.get_completions(_comp.id)
            prompt_str = self.tokenizer.apply_chat_template(messages, tokenize=False, add_generation_prompt=True)
            args = (prompt_str, _comp.choices[0].message.content, comp.response.input_tokens,
                    comp.response.output_tokens)
            reward = float(self.reward_fn(*args, **extra) if self.reward_in_process
                           else await self.async_reward_fn(*args, **extra))
            t += 1
            if reward == 0 and t < self.max_turns:
                messages += [{"role": "assistant", "content": _comp.choices[0].message.content}]
                messages += self.reflection_msg
                discount *= self.turn_discount
        reward = float(reward * discount)
        stats_tracker.get(self.rollout_stat_scope).scalar(reward=reward, num_turns=t)
        client.set_reward(_comp.id, reward)
        return client.export_completions(turn_discount=0.0), comp

    async def arun_episode(self, engine, data):
        rid = uuid.uuid4().hex
        results = await asyncio.gather(*[self._run_one_episode(engine, data, rid)
                                         for _ in range(self.gconfig.n_samples)])
        if self.dump_dir is not None:
            d = os.path.join(self.dump_dir, str(engine.get_version()))
            os.makedirs(d, exist_ok=True)
            qid = next((str(data[k]) for k in ("query_id", "id", "qid") if k in data), uuid.uuid4().hex)
            with open(os.path.join(d, f"{qid}.txt"), "a") as f:
                for i, (_, comp) in enumerate(results):
                    sl = comp.response.input_len + comp.response.output_len
                    f.write(f"idx: {i + 1} / {len(results)}, seqlen: {sl}, reward is {comp.reward}.\n"
                            f"prompt is \n{comp.messages}\nsequence is: \n"
                            f"{comp.completion.choices[0].message.content}\n")
        ret = {}
        for completions, _ in results:
            ret.update(completions)
        return ret
