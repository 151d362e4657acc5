"""Paired preference dataset for reward modelling / DPO. Rows: {"id", "prompt", "pos_answers": [...],
"neg_answers": [...]} with one-to-one (chosen, rejected) pairs, or pre-tokenised {"prompt_ids",
"pos_answer_ids", "neg_answer_ids"}. An item packs up to ``max_pairs_per_prompt`` randomly chosen pairs
as [chosen_0, rejected_0, chosen_1, ...] (the order ``dpo_loss`` expects) with the prompt length of the
group alongside. Behaviour of ``realhf/impl/dataset/rw_paired_dataset.py``."""
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from ...system.data_api import SequenceSample
from . import load_shuffle_split_dataset, register_dataset


class RewardModelingPairedDataset:
    def __init__(self, seed: int, dp_rank: int, world_size: int, tokenizer=None, max_length: int = 1024,
                 max_pairs_per_prompt: int = 2, dataset_path: Optional[str] = None,
                 dataset_builder: Optional[Callable[[], List[Dict]]] = None, eos_token_id: Optional[int] = None):
        rows = load_shuffle_split_dataset(seed, dp_rank, world_size, dataset_path, dataset_builder)
        self.rng = np.random.RandomState(seed)
        self.max_pairs_per_prompt = max_pairs_per_prompt
        self.ids, self.prompt_lens, self.pos, self.neg = [], [], [], []
        for r in rows:
            if "prompt_ids" in r:
                p = list(r["prompt_ids"])
                eos = [eos_token_id] if eos_token_id is not None else []
                pos = [(p + list(a) + eos)[:max_length] for a in r["pos_answer_ids"]]
                neg = [(p + list(a) + eos)[:max_length] for a in r["neg_answer_ids"]]
            else:
                assert tokenizer is not None, "rows hold text: a tokenizer is required"
                enc = lambda s: tokenizer(s, max_length=max_length, truncation=True, padding=False)["input_ids"]  # noqa: E731
                p = enc(r["prompt"])
                pos = [enc(r["prompt"] + a + tokenizer.eos_token) for a in r["pos_answers"]]
                neg = [enc(r["prompt"] + a + tokenizer.eos_token) for a in r["neg_answers"]]
            if len(pos) != len(neg):
                raise RuntimeError("pos_answers and neg_answers must be one-to-one pairs.")
            if not pos:
                raise RuntimeError("pos_answers and neg_answers must be non-empty.")
            self.ids.append(str(r.get("id", r.get("query_id"))))
            self.prompt_lens.append(min(len(p), max_length))
            self.pos.append(pos)
            self.neg.append(neg)

    def __len__(self):
        return len(self.ids)

    def __getitem__(self, i) -> SequenceSample:
        n = len(self.pos[i])
        k = min(self.max_pairs_per_prompt, n)
        pick = self.rng.choice(n, k, replace=False)
        seqs: List[List[int]] = []
        for j in pick:
            seqs += [self.pos[i][j], self.neg[i][j]]
        lens = [len(s) for s in seqs]
        ids = torch.tensor([t for s in seqs for t in s], dtype=torch.long)
        return SequenceSample(keys={"packed_input_ids", "prompt_lens"}, ids=[self.ids[i]],
                              seqlens=dict(packed_input_ids=[lens], prompt_lens=[[1]]),
                              data=dict(packed_input_ids=ids,
                                        prompt_lens=torch.tensor([self.prompt_lens[i]], dtype=torch.int32)))


register_dataset("rw_pair", RewardModelingPairedDataset)
