"""Countdown reward used by `train.py`: thin entry point kept for recipes that import it by this path; the
implementation (answer extraction from <answer> tags, check that every given number is used exactly once,
safe arithmetic evaluation, format / correctness scores) lives in `areal_vllm_amd.reward.countdown`."""
from areal_vllm_amd.reward.countdown import *  # noqa: F401,F403
from areal_vllm_amd.reward.countdown import countdown_reward_fn as compute_score  # noqa: F401
