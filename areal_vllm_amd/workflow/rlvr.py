"""Single-turn RL with a verifiable reward.

An episode samples the prompt ``n_samples`` times (independent requests, so a remote engine may spread them
over servers and the prefix cache shares their prefill), scores all samples, and returns the group as one
padded batch with the members in adjacent rows (group normalisation and dynamic sampling rely on that).
Counterpart of ``areal/workflow/rlvr.py`` in the reference; subclasses change how the request is built
(`prepare`) without touching the episode.
"""
import asyncio
import uuid
from typing import Any, Callable, Dict, List, Optional, Tuple

from ..api.cli_args import GenerationHyperparameters
from ..api.io_struct import ModelRequest
from ..api.workflow_api import RolloutWorkflow
from ..utils import logging
from ..utils.data import concat_padded_tensors
from .common import RolloutDump, Scorer, Transcript, detok, prompt_token_ids, record

logger = logging.getLogger("rlvr-workflow")


class RLVRWorkflow(RolloutWorkflow):
    def __init__(self, reward_fn: Callable, gconfig: GenerationHyperparameters, tokenizer=None,
                 enable_thinking: bool = False, rollout_stat_scope: str = "rollout", dump_dir: Optional[str] = None,
                 reward_in_process: bool = False, reward_timeout: float = 15.0):
        self.gconfig = gconfig
        self.tokenizer = tokenizer
        self.enable_thinking = enable_thinking
        self.rollout_stat_scope = rollout_stat_scope
        self.score = Scorer(reward_fn, reward_in_process, reward_timeout)
        self.dump = RolloutDump(dump_dir)

    def prepare(self, data: Dict[str, Any]) -> Tuple[ModelRequest, Dict[str, Any]]:
        """-> (the request every sample of the group issues, extra columns attached to every row)."""
        req = ModelRequest(rid=uuid.uuid4().hex, input_ids=prompt_token_ids(data, self.tokenizer, self.enable_thinking),
                           gconfig=self.gconfig.new(n_samples=1), tokenizer=self.tokenizer)
        return req, {}

    async def arun_episode(self, engine, data: Dict[str, Any]):
        request, columns = self.prepare(data)
        group: List[Any] = await asyncio.gather(*(engine.agenerate(request.copy())
                                                  for _ in range(self.gconfig.n_samples)))
        prompt_text = detok(self.tokenizer, request.input_ids)
        texts = [detok(self.tokenizer, r.output_tokens) for r in group]
        rewards = await asyncio.gather(*(self.score(prompt_text, text, r.input_tokens, r.output_tokens, data)
                                         for r, text in zip(group, texts)))
        rows = []
        for r, reward in zip(group, rewards):
            record(self.rollout_stat_scope, reward=reward)
            rows.append(Transcript.of_response(r).row(reward, **columns))
        self.dump.write(engine.get_version(), data,
                        (dict(seqlen=r.input_len + r.output_len, prompt_len=r.input_len, reward=reward,
                              prompt=prompt_text, completion=text) for r, reward, text in zip(group, rewards, texts)))
        return concat_padded_tensors(rows)
