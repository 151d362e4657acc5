"""Multi-turn self-correction on top of the OpenAI-style client.

The conversation lives in chat messages: a failed attempt is followed by the assistant's answer and a
reflection request, and the next attempt is a fresh chat completion through ``ArealOpenAI`` (which records
tokens, log-probs and versions of every completion it serves). Only the LAST completion of a sample gets the
(turn-discounted) reward; the episode returns the client's exported completions, one training row each.
Counterpart of ``areal/experimental/workflow/multi_turn_v2.py:21-143``.
"""
import asyncio
from typing import Any, Dict, List, Optional

from ...api.cli_args import GenerationHyperparameters
from ...api.workflow_api import RolloutWorkflow
from ...workflow.common import RolloutDump, Scorer, record
from ...workflow.multi_turn import FEEDBACK
from ..openai import ArealOpenAI


class MultiTurnWorkflow(RolloutWorkflow):
    def __init__(self, reward_fn, gconfig: GenerationHyperparameters, tokenizer, max_turns: int, turn_discount: float,
                 rollout_stat_scope: str = "rollout", dump_dir: Optional[str] = None, reward_in_process: bool = False):
        self.gconfig = gconfig
        self.tokenizer = tokenizer
        self.max_turns = max(1, int(max_turns))
        self.turn_discount = turn_discount
        self.rollout_stat_scope = rollout_stat_scope
        self.score = Scorer(reward_fn, reward_in_process)
        self.dump = RolloutDump(dump_dir)

    def _sampling_kwargs(self) -> Dict[str, Any]:
        g = self.gconfig
        return dict(frequency_penalty=g.frequency_penalty, max_completion_tokens=g.max_new_tokens, stop=g.stop,
                    temperature=g.temperature, top_p=g.top_p, store=True)

    async def _sample(self, engine, data: Dict[str, Any]):
        client = ArealOpenAI(engine=engine, tokenizer=self.tokenizer)
        chat: List[Dict[str, str]] = [dict(m) for m in data["messages"]]
        reward, stored, reply = 0.0, None, None
        for turn in range(self.max_turns):
            reply = await client.chat.completions.create(messages=chat, **self._sampling_kwargs())
            stored = client.get_completions(reply.id)
            answer = reply.choices[0].message.content
            rendered = self.tokenizer.apply_chat_template(chat, tokenize=False, add_generation_prompt=True)
            reward = await self.score(rendered, answer, stored.response.input_tokens, stored.response.output_tokens,
                                      data)
            if reward != 0:
                break
            chat = chat + [{"role": "assistant", "content": answer}, {"role": "user", "content": FEEDBACK}]
        reward = float(reward) * self.turn_discount ** turn
        record(self.rollout_stat_scope, reward=reward, num_turns=turn + 1)
        client.set_reward(reply.id, reward)
        return client.export_completions(turn_discount=0.0), stored

    async def arun_episode(self, engine, data: Dict[str, Any]):
        samples = await asyncio.gather(*(self._sample(engine, data) for _ in range(self.gconfig.n_samples)))
        self.dump.write(engine.get_version(), data,
                        (dict(seqlen=c.response.input_len + c.response.output_len, reward=c.reward, prompt=c.messages,
                              completion=c.completion.choices[0].message.content) for _, c in samples))
        merged: Dict[str, Any] = {}
        for rows, _ in samples:
            merged.update(rows)
        return merged
