"""Countdown arithmetic task: reach a target with +, -, *, / using each given number exactly once.

The reference ships this as an example (``examples/countdown/countdown.py`` builds a JSONL file with
``numbers``/``target`` rows, ``examples/countdown/train.py`` trains on it). Here the generator is a
library function so the dataset registry can serve ``countdown`` paths offline: a missing file is
generated deterministically from the seed.
"""
import json
import os
import random
from typing import Any, Dict, List, Optional

PROMPT = ("A conversation between User and Assistant. The user asks a question, and the Assistant solves it. "
          "The assistant first thinks about the reasoning process in the mind and then provides the user with the "
          "answer.\nUser: Using the numbers {numbers}, create an equation that equals {target}. You can use basic "
          "arithmetic operations (+, -, *, /) and each number can only be used once. Show your work in <think> "
          "</think> tags. And return the final answer in <answer> </answer> tags, for example <answer> (1 + 2) / 3 "
          "</answer>.\nAssistant: Let me solve this step by step.\n<think>")


def _random_expression(rng: random.Random, nums: List[int]):
    """Combine all `nums` in random order with random operators; returns (value, text) or None when a
    division is inexact or the value leaves [1, 999]."""
    order = nums[:]
    rng.shuffle(order)
    val, text = order[0], str(order[0])
    for n in order[1:]:
        op = rng.choice("+-*/")
        if op == "+":
            val, text = val + n, f"({text} + {n})"
        elif op == "-":
            val, text = val - n, f"({text} - {n})"
        elif op == "*":
            val, text = val * n, f"({text} * {n})"
        else:
            if n == 0 or val % n != 0:
                return None
            val, text = val // n, f"({text} / {n})"
    if not (1 <= val <= 999):
        return None
    return val, text


def generate(n_rows: int, seed: int = 0, min_numbers: int = 3, max_numbers: int = 4, max_value: int = 99
             ) -> List[Dict[str, Any]]:
    """Solvable instances: the target is the value of a random expression over the numbers."""
    rng = random.Random(seed)
    rows: List[Dict[str, Any]] = []
    while len(rows) < n_rows:
        k = rng.randint(min_numbers, max_numbers)
        nums = [rng.randint(1, max_value) for _ in range(k)]
        res = _random_expression(rng, nums)
        if res is None:
            continue
        rows.append({"numbers": nums, "target": res[0], "solution": res[1], "query_id": f"cd{len(rows)}"})
    return rows


def write_jsonl(path: str, rows: List[Dict[str, Any]]):
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")


def get_countdown_rl_dataset(path: str, split: Optional[str], tokenizer, rank: int, world_size: int,
                             max_length: Optional[int] = None, n_rows: int = 4096, **kwargs):
    """Rows {messages | input_ids, numbers, target, query_id}; `path` is a JSONL file (created when absent)
    or any string containing 'countdown' (generated in memory)."""
    from .synthetic import ListDataset

    if os.path.isfile(path):
        with open(path) as f:
            rows = [json.loads(l) for l in f if l.strip()]
    else:
        rows = generate(n_rows if split != "test" else max(64, n_rows // 16), seed=1 if split == "test" else 0)
    out = []
    for r in rows:
        text = PROMPT.format(numbers=r["numbers"], target=r["target"])
        row = dict(r, prompt=text, messages=[{"role": "user", "content": text}])
        if tokenizer is not None:
            row["input_ids"] = tokenizer.encode(text)
            if max_length is not None and len(row["input_ids"]) > max_length:
                continue
        out.append(row)
    return ListDataset(out[rank::world_size])
