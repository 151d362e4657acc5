"""Supervised fine-tuning interface: packed next-token loss on the answer tokens, evaluation as
token-mean loss / perplexity over an evaluation dataloader. Behaviour of
``realhf/impl/model/interface/sft_interface.py`` (146 lines)."""
from typing import Dict, Optional

import torch

from ...engine.sft.lm_engine import compute_packed_sft_loss
from ...system.data_api import SequenceSample
from ...system.model_api import Model, ModelInterface, register_interface
from ...utils import stats_tracker
from ._convert import to_padded_batch


def _sft_batch(data: SequenceSample, device) -> Dict[str, torch.Tensor]:
    batch = to_padded_batch(data, device, token_keys=("prompt_mask",))
    pm = batch.pop("prompt_mask", None)
    am = batch["attention_mask"]
    batch["loss_mask"] = (am & ~pm.bool()) if pm is not None else am.clone()
    return batch


class SFTInterface(ModelInterface):
    def __init__(self, token_normalize_scope: str = "global", enable_save: bool = True, **_):
        self.token_normalize_scope = token_normalize_scope
        self.enable_save = enable_save

    def train_step(self, model: Model, data: SequenceSample, mb_spec) -> Dict[str, float]:
        eng = model.module
        eng.config.mb_spec = mb_spec
        batch = _sft_batch(data, eng.device)
        eng.train()
        stat = eng.train_batch(batch, loss_fn=compute_packed_sft_loss,
                               loss_weight_fn=lambda x: x["loss_mask"].count_nonzero())
        stats_tracker.scalar(**stat)
        out = stats_tracker.export(reduce_group=eng.data_parallel_group)
        out["n_tokens"] = float(batch["attention_mask"].sum())
        model.inc_version()
        return out

    @torch.no_grad()
    def evaluate(self, model: Model, eval_dataloader) -> Dict[str, float]:
        eng = model.module
        tot, n = 0.0, 0
        for data in eval_dataloader:
            batch = _sft_batch(data, eng.device)
            eng.eval()
            loss = eng.eval_batch(batch, loss_fn=compute_packed_sft_loss,
                                  loss_weight_fn=lambda x: x["loss_mask"].count_nonzero())
            stats_tracker.export(reduce_group=eng.data_parallel_group)  # drop per-batch training statistics
            tot += float(loss)
            n += 1
        if n == 0:
            return {}
        return dict(loss=tot / n, ppl=float(torch.tensor(tot / n).exp()))

    def save(self, model: Model, save_dir: str):
        if self.enable_save:
            from ...api.io_struct import SaveLoadMeta

            model.module.save(SaveLoadMeta(path=save_dir, weight_format="hf", with_optim=False,
                                           tokenizer=model.tokenizer))


register_interface("sft", SFTInterface)
