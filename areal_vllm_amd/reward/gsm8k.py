"""GSM8K reward: 1 when the boxed / final answer of the completion equals the dataset answer
(``examples/math/gsm8k_grpo.py:24-29`` in the reference)."""
from .math_parser import process_results


def gsm8k_reward_fn(prompt, completions, prompt_ids, completion_ids, answer, **kwargs) -> float:
    return float(process_results(completions, answer)[0])
