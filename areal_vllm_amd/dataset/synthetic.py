"""Synthetic datasets for offline smoke runs, tests and benchmarks (no tokenizer, no network).

``synthetic:<task>[:n=<rows>][:len=<prompt tokens>][:vocab=<V>]`` as a dataset path selects one:

* ``copy-token``: prompts are random token ids; the reward of a completion is the fraction of its
  tokens equal to the prompt's first token - a task a tiny random-init model learns in a few GRPO steps.
* ``low-token``: reward = fraction of completion tokens with id < vocab/8 (a dense signal from step 0;
  used by the end-to-end learning test).
* ``random``: random prompts and a pseudo-random reward keyed on the completion (throughput benchmarks).
* SFT rows (``type="sft"``) are arithmetic-progression token sequences with the loss on the second half.
"""
import random
from typing import Any, Dict, List, Optional


def parse_spec(path: str) -> Dict[str, Any]:
    parts = path.split(":")
    assert parts[0] == "synthetic"
    spec: Dict[str, Any] = dict(task=parts[1] if len(parts) > 1 else "copy-token", n=512, len=16, vocab=512)
    for p in parts[2:]:
        k, _, v = p.partition("=")
        spec[k] = int(v)
    return spec


class ListDataset(list):
    """A list that also answers the few `datasets.Dataset` calls the training scripts make."""

    def shard(self, num_shards: int, index: int):
        return ListDataset(self[index::num_shards])


def get_synthetic_dataset(path: str, rank: int, world_size: int, type: str = "rl", split: Optional[str] = None,
                          seed: int = 0, **kwargs) -> ListDataset:
    spec = parse_spec(path)
    rng = random.Random(seed + (1 if split == "test" else 0))
    rows: List[Dict[str, Any]] = []
    for i in range(spec["n"]):
        if type == "sft":
            start, step = rng.randrange(2, spec["vocab"] // 2), rng.randrange(1, 4)
            n = rng.randrange(spec["len"], 2 * spec["len"])
            ids = [2 + (start + step * j) % (spec["vocab"] - 2) for j in range(n)]
            rows.append({"input_ids": ids, "loss_mask": [0] * (n // 2) + [1] * (n - n // 2)})
        else:
            n = rng.randrange(max(2, spec["len"] // 2), spec["len"] + 1)
            ids = [rng.randrange(2, spec["vocab"]) for _ in range(n)]
            rows.append({"input_ids": ids, "query_id": f"syn{i}", "target": ids[0]})
    return ListDataset(rows[rank::world_size])


def copy_token_reward(prompt, completions, prompt_ids, completion_ids, target=None, **kwargs) -> float:
    if not completion_ids:
        return 0.0
    t = prompt_ids[0] if target is None else target
    return sum(1 for x in completion_ids if x == t) / len(completion_ids)


def low_token_reward(prompt, completions, prompt_ids, completion_ids, vocab: int = 512, **kwargs) -> float:
    if not completion_ids:
        return 0.0
    return sum(1 for x in completion_ids if x < vocab // 8) / len(completion_ids)


def random_reward(prompt, completions, prompt_ids, completion_ids, **kwargs) -> float:
    return float((sum(completion_ids[:8]) * 2654435761 % 1000) < 500)


def get_synthetic_reward_fn(path: str):
    return {"copy-token": copy_token_reward, "low-token": low_token_reward, "random": random_reward}[parse_spec(path)["task"]]
