"""Single-step math/code agent: ask once for a group of answers, score them with the environment,
shape the rewards and drop prompts that are too easy / too hard for the current policy.

Reward shaping ``((success - 0.5) * 2 - bias) * scaling`` and the success-rate window follow
``realhf/impl/agent/math_single_step_agent.py:23-250``.
"""
import asyncio
import time
from typing import List, Optional

import numpy as np

from ...system.agent_api import Agent, EnvironmentService, register_agent
from ...system.data_api import SequenceSample
from ...system.model_api import BundledGenerationOutputs
from ...utils import logging
from ._common import decode, group_sample, log_rewards_to_file

logger = logging.getLogger("math-single-step-agent")


class MathSingleStepAgent(Agent):
    def __init__(self, gconfig, tokenizer_path: Optional[str] = None, answer_save_path: Optional[str] = None,
                 success_rate_lb: float = 0.0, success_rate_ub: float = 1.0, reward_scaling: float = 1.0,
                 reward_bias: float = 0.0, tokenizer=None):
        self.gconfig = gconfig
        self.tokenizer = tokenizer
        if tokenizer is None and tokenizer_path:
            from ...utils.hf_utils import load_hf_tokenizer

            self.tokenizer = load_hf_tokenizer(tokenizer_path)
        self.answer_save_path = answer_save_path
        self.success_rate_lb, self.success_rate_ub = success_rate_lb, success_rate_ub
        self.reward_scaling, self.reward_bias = reward_scaling, reward_bias

    async def collect_trajectory(self, prompt: SequenceSample, env: EnvironmentService, obs_queue: asyncio.Queue,
                                 act_queue: asyncio.Queue) -> List[SequenceSample]:
        await env.reset()
        assert prompt.bs == 1
        prompt_ids = prompt.data["packed_prompts"].cpu().numpy().tolist()
        qid = prompt.ids[0]
        birth_time = int(time.time() * 1000)
        await obs_queue.put((qid, prompt_ids, self.gconfig))
        act: BundledGenerationOutputs = await act_queue.get()
        prompt_str = decode(self.tokenizer, act.prompt_ids)
        answers = [decode(self.tokenizer, o) for o in act.output_ids]
        _, success, *_ = await env.step((qid, answers, act.output_ids))
        rewards = [((float(s) - 0.5) * 2 - self.reward_bias) * self.reward_scaling for s in success]
        log_rewards_to_file(self.answer_save_path, str(qid), prompt_str, answers, act.seqlens, len(prompt_ids),
                            rewards, [bool(s) for s in success], act.version_start, act.version_end)
        rate = float(np.mean([float(s) for s in success]))
        if rate < self.success_rate_lb:
            logger.info(f"Query ID {qid} reward too low: {rate} < {self.success_rate_lb}.")
            return []
        if rate > self.success_rate_ub:
            logger.info(f"Query ID {qid} reward too high: {rate} > {self.success_rate_ub}.")
            return []
        task_id = int(prompt.data["task_ids"][0]) if prompt.data.get("task_ids") is not None else 0
        return [group_sample(qid, prompt_ids, act, rewards, birth_time, task_id)]


register_agent("math-single-step", MathSingleStepAgent)
