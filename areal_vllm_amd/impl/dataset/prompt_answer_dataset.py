"""Prompt + answer dataset for supervised fine-tuning: items carry ``packed_input_ids`` (prompt followed
by answer and EOS) and a ``prompt_mask`` (1 on prompt tokens). Rows: {"prompt", "answer"} as text or
{"prompt_ids", "answer_ids"} pre-tokenised. Behaviour of ``realhf/impl/dataset/prompt_answer_dataset.py``."""
from typing import Callable, Dict, List, Optional

import torch

from ...system.data_api import SequenceSample
from . import load_shuffle_split_dataset, register_dataset


class PromptAnswerDataset:
    def __init__(self, seed: int, dp_rank: int, world_size: int, tokenizer=None, max_length: Optional[int] = None,
                 dataset_path: Optional[str] = None, dataset_builder: Optional[Callable[[], List[Dict]]] = None,
                 eos_token_id: Optional[int] = None):
        rows = load_shuffle_split_dataset(seed, dp_rank, world_size, dataset_path, dataset_builder)
        self.ids, self.tokens, self.masks = [], [], []
        for r in rows:
            if "prompt_ids" in r:
                p, a = list(r["prompt_ids"]), list(r["answer_ids"])
                if eos_token_id is not None:
                    a = a + [eos_token_id]
            else:
                assert tokenizer is not None, "rows hold text: a tokenizer is required"
                p = tokenizer(r["prompt"], padding=False, return_attention_mask=False)["input_ids"]
                a = tokenizer(r["answer"] + tokenizer.eos_token, padding=False, add_special_tokens=False,
                              return_attention_mask=False)["input_ids"]
            seq = (p + a)[: max_length] if max_length else p + a
            if len(seq) <= len(p):
                continue
            self.ids.append(str(r.get("id", r.get("query_id"))))
            self.tokens.append(seq)
            self.masks.append([1] * len(p) + [0] * (len(seq) - len(p)))

    def __len__(self):
        return len(self.ids)

    def __getitem__(self, i) -> SequenceSample:
        t = torch.tensor(self.tokens[i], dtype=torch.long)
        m = torch.tensor(self.masks[i], dtype=torch.bool)
        return SequenceSample(keys={"packed_input_ids", "prompt_mask"}, ids=[self.ids[i]],
                              seqlens=dict(packed_input_ids=[[len(t)]], prompt_mask=[[len(t)]]),
                              data=dict(packed_input_ids=t, prompt_mask=m))


register_dataset("prompt_answer", PromptAnswerDataset)
