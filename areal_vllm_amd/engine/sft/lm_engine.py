"""Supervised fine-tuning on packed sequences: token-mean NLL + per-sequence perplexity statistics.

Counterpart of ``areal/engine/sft/lm_engine.py`` (83 lines) in the reference.
"""
from typing import Any, Dict

import torch

from ...api.cli_args import TrainEngineConfig
from ...api.engine_api import TrainEngine
from ...utils import stats_tracker
from ..train_engine import LogitsProxy, NativeTrainEngine
from ... import ops


class LMEngine:
    def __init__(self, engine: TrainEngine):
        self.engine = engine

    def train_lm(self, data: Dict[str, Any]):
        self.engine.train()
        return self.engine.train_batch(input_=data, loss_fn=compute_packed_sft_loss,
                                       loss_weight_fn=lambda x: x["loss_mask"].count_nonzero())

    def evaluate_lm(self, data: Dict[str, Any]):
        self.engine.eval()
        return self.engine.eval_batch(input_=data, loss_fn=compute_packed_sft_loss,
                                      loss_weight_fn=lambda x: x["loss_mask"].count_nonzero())


def compute_packed_sft_loss(logits, input_: Dict[str, Any]) -> torch.Tensor:
    input_ids = input_["input_ids"]
    cu = input_["cu_seqlens"]
    # loss_mask[t] marks token t as a target; shifting makes position t predict token t+1
    loss_mask = torch.roll(input_["loss_mask"].bool(), shifts=-1, dims=-1).clone()
    loss_mask[cu[1:].long() - 1] = False  # the last position of a sequence predicts nothing
    labels = torch.roll(input_ids, shifts=-1, dims=-1)
    if isinstance(logits, LogitsProxy):
        logprobs = logits.logprobs(labels, 1.0)
    else:
        logprobs = ops.gather_logprobs(logits, labels, 1.0)
    logprobs = torch.where(loss_mask, logprobs, 0)
    loss = -logprobs.sum() / loss_mask.count_nonzero().clamp_min(1)
    with torch.no_grad():
        lens = (cu[1:] - cu[:-1]).long()
        seq_id = torch.repeat_interleave(torch.arange(lens.numel(), device=logprobs.device), lens)
        seq_lp = torch.zeros(lens.numel(), device=logprobs.device).index_add_(0, seq_id, logprobs.detach().float())
        seq_n = torch.zeros(lens.numel(), device=logprobs.device).index_add_(0, seq_id, loss_mask.float())
        seqlogp = seq_lp / seq_n.clamp_min(1)
    stats_tracker.denominator(n_seqs=torch.ones(lens.numel(), dtype=torch.bool, device=logprobs.device),
                              n_tokens=torch.ones(logprobs.shape[0], dtype=torch.bool, device=logprobs.device),
                              n_valid_tokens=loss_mask, prompt_tokens=~loss_mask)
    stats_tracker.stat(ppl=(-seqlogp).exp().float(), denominator="n_seqs")
    stats_tracker.stat(loss=-logprobs.detach().float(), denominator="n_valid_tokens")
    return loss


class NativeLMEngine(NativeTrainEngine):
    def __init__(self, config: TrainEngineConfig):
        super().__init__(config)
        self.lm_engine = LMEngine(self)

    def train_lm(self, data):
        return self.lm_engine.train_lm(data)

    def evaluate_lm(self, data):
        return self.lm_engine.evaluate_lm(data)


FSDPLMEngine = NativeLMEngine
MegatronLMEngine = NativeLMEngine
