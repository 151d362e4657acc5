"""GSM8K in the two shapes the trainers read.

* supervised: one row per problem, ``input_ids`` = tokens of question + worked answer + EOS and a
  ``loss_mask`` that is 1 exactly on the answer (and EOS) tokens;
* RL: one row per problem with a single-turn chat ``messages`` prompt that asks for a boxed final answer;
  the ``answer`` column stays for the reward function.

Rows are sharded by data-parallel rank with `datasets`' node splitter. Role of ``areal/dataset/gsm8k.py``.
"""
from typing import Any, Dict, Optional

from . import load_raw_dataset

BOXED_SUFFIX = "\nPlease put your final answer within \\boxed{}."


def _finish(ds, keep_if, max_length: Optional[int], rank: int, world_size: int):
    from datasets.distributed import split_dataset_by_node

    if max_length is not None:
        ds = ds.filter(lambda row: keep_if(row) <= max_length)
    return split_dataset_by_node(ds, rank=rank, world_size=world_size)


def get_gsm8k_sft_dataset(path: str, split: str, tokenizer, rank: int, world_size: int,
                          max_length: Optional[int] = None):
    def tokenize(row: Dict[str, Any]) -> Dict[str, Any]:
        question = tokenizer.encode(row["question"])
        full = tokenizer.encode(row["question"] + row["answer"] + tokenizer.eos_token)
        n_answer = len(full) - len(question)
        return dict(input_ids=full, loss_mask=[0] * len(question) + [1] * n_answer)

    ds = load_raw_dataset(path, split, name="main").map(tokenize, remove_columns=["question", "answer"])
    return _finish(ds, lambda row: len(row["input_ids"]), max_length, rank, world_size)


def get_gsm8k_rl_dataset(path: str, split: str, tokenizer, rank: int, world_size: int,
                         max_length: Optional[int] = None):
    def as_chat(row: Dict[str, Any]) -> Dict[str, Any]:
        return dict(messages=[dict(role="user", content=row["question"] + BOXED_SUFFIX)])

    ds = load_raw_dataset(path, split, name="main").map(as_chat, remove_columns=["question"])
    return _finish(ds, lambda row: len(tokenizer.encode(row["messages"][0]["content"])), max_length, rank, world_size)
