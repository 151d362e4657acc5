"""Agent that performs `episode_length` generation round-trips and returns random-reward trajectories;
used to exercise the rollout plumbing without a tokenizer or verifier
(``realhf/impl/agent/null_agent.py``)."""
import asyncio
import random
import time
from typing import List

from ...system.agent_api import Agent, EnvironmentService, register_agent
from ...system.data_api import SequenceSample
from ._common import group_sample


class NullAgent(Agent):
    def __init__(self, gconfig, episode_length: int = 1, traj_size: int = 1, seed: int = 0):
        self.gconfig, self.episode_length, self.traj_size = gconfig, episode_length, traj_size
        self.rng = random.Random(seed)

    async def collect_trajectory(self, prompt: SequenceSample, env: EnvironmentService, obs_queue: asyncio.Queue,
                                 act_queue: asyncio.Queue) -> List[SequenceSample]:
        qid = prompt.ids[0]
        prompt_ids = prompt.data["packed_prompts"].cpu().numpy().tolist()
        act = None
        for _ in range(self.episode_length):
            await obs_queue.put((qid, prompt_ids, self.gconfig))
            act = await act_queue.get()
        out = []
        for i in range(self.traj_size):
            rewards = [self.rng.random() for _ in act.seqs]
            out.append(group_sample(f"{qid}-{i}" if self.traj_size > 1 else qid, prompt_ids, act, rewards,
                                    int(time.time() * 1000)))
        return out


register_agent("null", NullAgent)
