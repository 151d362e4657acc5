"""Reward functions. Signature: ``fn(prompt, completions, prompt_ids, completion_ids, **data) -> float``."""
from .gsm8k import gsm8k_reward_fn  # noqa: F401

VALID_REWARD_FN = ["clevr_count_70k", "countdown", "geometry3k", "gsm8k"]


def get_custom_reward_fn(path: str, **kwargs):
    """Select a reward function by dataset path (``areal/reward/__init__.py:1-17`` in the reference)."""
    if path.startswith("synthetic:"):
        from ..dataset.synthetic import get_synthetic_reward_fn

        return get_synthetic_reward_fn(path)
    if "clevr_count_70k" in path:
        from .clevr_count_70k import clevr_count_70k_reward_fn

        return clevr_count_70k_reward_fn
    if "geometry3k" in path:
        from .geometry3k import geometry3k_reward_fn

        return geometry3k_reward_fn
    if "countdown" in path:
        from .countdown import countdown_reward_fn

        return countdown_reward_fn
    if "gsm8k" in path:
        return gsm8k_reward_fn
    raise ValueError(f"Reward function {path} is not supported. Supported reward functions are: {VALID_REWARD_FN}.")
