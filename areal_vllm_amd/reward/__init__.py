"""Reward functions. Signature: ``fn(prompt, completions, prompt_ids, completion_ids, **data) -> float``.

`get_custom_reward_fn(path)` picks the function from the dataset path, as ``areal/reward/__init__.py:1-17`` does
in the reference; here the families are a table of (path key, module, attribute), imported on demand."""
from importlib import import_module

from .gsm8k import gsm8k_reward_fn  # noqa: F401

# checked in this order: the first key contained in the path wins
_BY_PATH_KEY = (
    ("clevr_count_70k", ".clevr_count_70k", "clevr_count_70k_reward_fn"),
    ("geometry3k", ".geometry3k", "geometry3k_reward_fn"),
    ("countdown", ".countdown", "countdown_reward_fn"),
    ("gsm8k", ".gsm8k", "gsm8k_reward_fn"),
)
VALID_REWARD_FN = sorted(key for key, _, _ in _BY_PATH_KEY)


def get_custom_reward_fn(path: str, **kwargs):
    if path.startswith("synthetic:"):
        from ..dataset.synthetic import get_synthetic_reward_fn

        return get_synthetic_reward_fn(path)
    for key, module, attr in _BY_PATH_KEY:
        if key in path:
            return getattr(import_module(module, __name__), attr)
    raise ValueError(f"Reward function {path} is not supported. Supported reward functions are: {VALID_REWARD_FN}.")
