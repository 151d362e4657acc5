"""GSM8K: SFT rows (question+answer token ids with a loss mask over the answer) and RL rows (a chat
message asking for a boxed answer, plus the reference answer). Reference: ``areal/dataset/gsm8k.py:7-69``."""
from typing import Optional

from . import load_raw_dataset

BOXED_SUFFIX = "\nPlease put your final answer within \\boxed{}."


def _shard(ds, rank: int, world_size: int):
    from datasets.distributed import split_dataset_by_node

    return split_dataset_by_node(ds, rank=rank, world_size=world_size)


def get_gsm8k_sft_dataset(path: str, split: str, tokenizer, rank: int, world_size: int,
                          max_length: Optional[int] = None):
    ds = load_raw_dataset(path, split, name="main")

    def process(sample):
        seq = tokenizer.encode(sample["question"] + sample["answer"] + tokenizer.eos_token)
        n_prompt = len(tokenizer.encode(sample["question"]))
        return {"input_ids": seq, "loss_mask": [0] * n_prompt + [1] * (len(seq) - n_prompt)}

    ds = ds.map(process).remove_columns(["question", "answer"])
    if max_length is not None:
        ds = ds.filter(lambda x: len(x["input_ids"]) <= max_length)
    return _shard(ds, rank, world_size)


def get_gsm8k_rl_dataset(path: str, split: str, tokenizer, rank: int, world_size: int,
                         max_length: Optional[int] = None):
    ds = load_raw_dataset(path, split, name="main")
    ds = ds.map(lambda s: {"messages": [{"role": "user", "content": s["question"] + BOXED_SUFFIX}]})
    ds = ds.remove_columns(["question"])
    if max_length is not None:
        ds = ds.filter(lambda s: len(tokenizer.encode(s["messages"][0]["content"])) <= max_length)
    return _shard(ds, rank, world_size)
