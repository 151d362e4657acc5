"""Multi-turn self-correction: answer, and while the reward stays zero append a fixed feedback turn and try
again, up to ``max_turns``; every extra turn multiplies the final reward by ``turn_discount``.

One trajectory spans the whole conversation. The model's answers are generated segments; the prompt and the
feedback turns are context. Feedback is spliced in as TOKEN IDS (the tail that the chat template adds for
an extra user turn, computed once), so no decode -> re-encode drift can break the prefix property that
`Transcript.extend_to` checks. Same request id across the turns of one sample: a remote engine keeps it on
the server that already holds its prefix. Counterpart of ``areal/workflow/multi_turn.py:23-173``.
"""
import asyncio
import uuid
from typing import Any, Dict, List, Optional

from ..api.cli_args import GenerationHyperparameters
from ..api.io_struct import ModelRequest
from ..api.workflow_api import RolloutWorkflow
from ..utils import logging
from ..utils.data import concat_padded_tensors
from .common import RolloutDump, Scorer, Transcript, prompt_token_ids, record

logger = logging.getLogger("multi-turn-workflow")

FEEDBACK = ("Your answer is either wrong or not parsable to the reward function. You may misunderstand the original "
            "question. Please carefully read the original question, check the previous errors, and try to answer "
            "it again.")


def feedback_turn_ids(tokenizer, text: str) -> List[int]:
    """Tokens the chat template appends for `text` as a user turn (plus the next generation prompt) after an
    assistant turn."""
    stub = [{"role": "assistant", "content": "some random message."}]
    before = tokenizer.apply_chat_template(stub, tokenize=True)
    after = tokenizer.apply_chat_template(stub + [{"role": "user", "content": text}], tokenize=True,
                                          add_generation_prompt=True)
    return list(after[len(before):])


class MultiTurnWorkflow(RolloutWorkflow):
    def __init__(self, reward_fn, gconfig: GenerationHyperparameters, tokenizer, max_turns: int, turn_discount: float,
                 rollout_stat_scope: str = "rollout", dump_dir: Optional[str] = None, reward_in_process: bool = False,
                 feedback: str = FEEDBACK):
        self.gconfig = gconfig
        self.tokenizer = tokenizer
        self.max_turns = max(1, int(max_turns))
        self.turn_discount = turn_discount
        self.rollout_stat_scope = rollout_stat_scope
        self.score = Scorer(reward_fn, reward_in_process)
        self.dump = RolloutDump(dump_dir)
        self.multi_turn_prompt_ids = feedback_turn_ids(tokenizer, feedback)

    async def _sample(self, engine, data: Dict[str, Any], rid: str) -> Dict[str, Any]:
        tok = self.tokenizer
        prompt = prompt_token_ids(data, tok)
        script = Transcript()
        reward, turns, prompt_text, answer_text = 0.0, 0, "", ""
        for turns in range(1, self.max_turns + 1):
            resp = await engine.agenerate(ModelRequest(rid=rid, input_ids=prompt, gconfig=self.gconfig.new(n_samples=1),
                                                       tokenizer=tok))
            script.extend_to(resp.input_tokens).generated(resp.output_tokens, resp.output_logprobs,
                                                          resp.output_versions)
            prompt_text, answer_text = tok.decode(prompt), tok.decode(resp.output_tokens)
            reward = await self.score(prompt_text, answer_text, resp.input_tokens, resp.output_tokens, data)
            if reward != 0:
                break
            closing = [] if resp.output_tokens and resp.output_tokens[-1] == tok.eos_token_id else [tok.eos_token_id]
            prompt = prompt + list(resp.output_tokens) + closing + self.multi_turn_prompt_ids
        reward = float(reward) * self.turn_discount ** (turns - 1)
        record(self.rollout_stat_scope, reward=reward, num_turns=turns)
        return dict(row=script.row(reward), seqlen=len(script), reward=reward, prompt=prompt_text,
                    completion=answer_text)

    async def arun_episode(self, engine, data: Dict[str, Any]):
        base = uuid.uuid4().hex
        samples = await asyncio.gather(*(self._sample(engine, data, f"{base}-{i}")
                                         for i in range(self.gconfig.n_samples)))
        self.dump.write(engine.get_version(), data, samples)
        return concat_padded_tensors([s["row"] for s in samples])
