"""Multi-turn math agent: up to `num_turns` attempts; after a wrong answer a feedback turn is appended
and the model tries again. Every turn becomes one sequence of the returned item; turn rewards are
discounted backwards (``r_t += discount * r_{t+1}``) so that earlier turns are credited for eventual
success. Behaviour of ``realhf/impl/agent/math_multi_turn_agent.py:23-296``.
"""
import asyncio
import time
from typing import List, Optional

import torch

from ...system.agent_api import Agent, EnvironmentService, register_agent
from ...system.data_api import SequenceSample
from ...system.model_api import BundledGenerationOutputs
from ...utils import logging
from ._common import decode, log_rewards_to_file

logger = logging.getLogger("math-multi-turn-agent")

FEEDBACK_OK = "Congratulations! You are correct!"
FEEDBACK_WRONG = "Unfortunately your answer is wrong. Let's try again."


class MathMultiTurnAgent(Agent):
    def __init__(self, gconfig, tokenizer_path: Optional[str] = None, answer_save_path: Optional[str] = None,
                 reward_scaling: float = 1.0, reward_bias: float = 0.0, turn_level_discount: float = 1.0,
                 num_turns: int = 5, tokenizer=None):
        self.gconfig = gconfig.new(n_samples=1)
        self.tokenizer = tokenizer
        if tokenizer is None and tokenizer_path:
            from ...utils.hf_utils import load_hf_tokenizer

            self.tokenizer = load_hf_tokenizer(tokenizer_path)
        self.answer_save_path = answer_save_path
        self.reward_scaling, self.reward_bias = reward_scaling, reward_bias
        self.turn_level_discount = turn_level_discount
        self.num_turns = num_turns

    def _feedback_ids(self, ok: bool) -> List[int]:
        text = FEEDBACK_OK if ok else FEEDBACK_WRONG
        if self.tokenizer is None:
            return [3, 4] if ok else [5, 6]
        msgs = [{"role": "user", "content": text}]
        return list(self.tokenizer.apply_chat_template(msgs, add_generation_prompt=True, tokenize=True))

    async def collect_trajectory(self, prompt: SequenceSample, env: EnvironmentService, obs_queue: asyncio.Queue,
                                 act_queue: asyncio.Queue) -> List[SequenceSample]:
        await env.reset()
        assert prompt.bs == 1
        prompt_ids = prompt.data["packed_prompts"].cpu().numpy().tolist()
        qid = prompt.ids[0]
        birth_time = int(time.time() * 1000)
        token_ids = list(prompt_ids)
        seqs, logps, pmasks, no_eos, vs, ve, rewards, answers, success = [], [], [], [], [], [], [], [], []
        for turn in range(self.num_turns):
            await obs_queue.put((f"{qid}-{turn}", token_ids, self.gconfig))
            act: BundledGenerationOutputs = await act_queue.get()
            seq = act.seqs[0]
            ans = decode(self.tokenizer, act.output_ids[0])
            _, ok, *_ = await env.step((qid, [ans], [act.output_ids[0]]))
            ok = bool(ok[0])
            seqs.append(seq)
            logps.append(act.logprobs[0])
            pmasks.append([1] * act.prompt_len + [0] * (len(seq) - act.prompt_len))
            no_eos.append(act.no_eos[0])
            vs.append(act.version_start[0])
            ve.append(act.version_end[0])
            rewards.append(((float(ok) - 0.5) * 2 - self.reward_bias) * self.reward_scaling)
            answers.append(ans)
            success.append(ok)
            token_ids = seq + self._feedback_ids(ok)
        for i in reversed(range(len(rewards) - 1)):
            rewards[i] += self.turn_level_discount * rewards[i + 1]
        log_rewards_to_file(self.answer_save_path, str(qid), decode(self.tokenizer, prompt_ids), answers,
                            [len(s) for s in seqs], len(prompt_ids), rewards, success, vs, ve)
        n = len(seqs)
        data = dict(
            packed_input_ids=torch.tensor([t for s in seqs for t in s], dtype=torch.long),
            prompt_mask=torch.tensor([m for p in pmasks for m in p], dtype=torch.bool),
            packed_logprobs=torch.tensor([x for lp in logps for x in lp], dtype=torch.float32),
            seq_no_eos_mask=torch.tensor(no_eos, dtype=torch.bool),
            packed_prompts=torch.tensor(prompt_ids, dtype=torch.long),
            version_start=torch.tensor(vs, dtype=torch.int), version_end=torch.tensor(ve, dtype=torch.int),
            rewards=torch.tensor(rewards, dtype=torch.float32), birth_time=torch.tensor([birth_time], dtype=torch.long),
        )
        seqlens = dict(packed_input_ids=[[len(s) for s in seqs]], prompt_mask=[[len(s) for s in seqs]],
                       packed_logprobs=[[len(s) - 1 for s in seqs]], seq_no_eos_mask=[[1] * n],
                       packed_prompts=[[len(prompt_ids)]], version_start=[[1] * n], version_end=[[1] * n],
                       rewards=[[1] * n], birth_time=[[1]])
        return [SequenceSample(keys=set(data), ids=[qid], seqlens=seqlens, data=data)]


register_agent("math-multi-turn", MathMultiTurnAgent)
