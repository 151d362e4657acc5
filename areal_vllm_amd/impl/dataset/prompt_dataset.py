"""Prompt-only dataset for RL: items carry ``packed_prompts`` (token ids) and ``task_ids``.
Rows: {"prompt": str | "input_ids": [int], "query_id"/"id": str}. Behaviour of
``realhf/impl/dataset/prompt_dataset.py``."""
from typing import Callable, Dict, List, Optional

import torch

from ...system.data_api import SequenceSample
from ...utils import logging
from . import load_shuffle_split_dataset, register_dataset

logger = logging.getLogger("prompt-dataset")
RL_TASKS = ["math", "code", "rlhf", "stem"]


def tokenize_prompts(rows: List[Dict], tokenizer, max_length: Optional[int]) -> List[List[int]]:
    out = []
    for r in rows:
        if "input_ids" in r:
            ids = [int(t) for t in r["input_ids"]]
        else:
            assert tokenizer is not None, "rows hold text prompts: a tokenizer is required"
            ids = tokenizer(r["prompt"], truncation=max_length is not None, max_length=max_length,
                            padding=False, return_attention_mask=False)["input_ids"]
        out.append(ids)
    return out


class PromptDataset:
    def __init__(self, seed: int, dp_rank: int, world_size: int, tokenizer=None, max_length: Optional[int] = None,
                 dataset_path: Optional[str] = None, dataset_builder: Optional[Callable[[], List[Dict]]] = None,
                 filter_threshold: float = 1e4, max_filter_percentage: float = 0.0):
        rows = load_shuffle_split_dataset(seed, dp_rank, world_size, dataset_path, dataset_builder)
        toks = tokenize_prompts(rows, tokenizer, None)
        keep = [i for i, t in enumerate(toks) if max_length is None or len(t) <= max_length]
        self.ids = [str(rows[i].get("query_id", rows[i].get("id"))) for i in keep]
        self.prompts = [toks[i] for i in keep]
        self.tasks = [RL_TASKS.index(rows[i].get("task", "math")) for i in keep]
        self.active_indices = list(range(len(self.ids)))
        self.filter_threshold, self.max_filter_percentage = filter_threshold, max_filter_percentage
        logger.info(f"prompt dataset shard {dp_rank}/{world_size}: {len(self.ids)} prompts")

    def __len__(self):
        return len(self.active_indices)

    def __getitem__(self, idx) -> SequenceSample:
        i = self.active_indices[idx]
        p = torch.tensor(self.prompts[i], dtype=torch.long)
        return SequenceSample(keys={"packed_prompts", "task_ids"}, ids=[self.ids[i]],
                              seqlens=dict(packed_prompts=[[len(p)]], task_ids=[[1]]),
                              data=dict(packed_prompts=p, task_ids=torch.tensor([self.tasks[i]], dtype=torch.long)))

    def filter(self, eval_scores: Dict[str, float]):
        """Drop prompts the policy already solves (score above the threshold), at most a fixed share per
        call, so that later epochs focus on what is still hard."""
        scored = [(eval_scores[self.ids[i]], i) for i in self.active_indices if self.ids[i] in eval_scores]
        cand = sorted([x for x in scored if x[0] > self.filter_threshold], reverse=True)
        n_max = int(len(self.active_indices) * self.max_filter_percentage)
        drop = {i for _, i in cand[:n_max]}
        self.active_indices = [i for i in self.active_indices if i not in drop]
        logger.info(f"filtered {len(drop)} prompts, {len(self.active_indices)} remain")


register_dataset("prompt", PromptDataset)
