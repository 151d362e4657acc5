"""Geometry3K reward: bracketed final answer compared as a normalised string / number
(``areal/reward/geometry3k.py:1-29``)."""
import re

from .math_parser import math_equal, strip_string


def extract_answer(pred_str: str, data_name=None, use_last_number: bool = True) -> str:
    spans = re.findall(r"\[([^\]]+)\]", pred_str)
    return spans[-1].strip() if spans else ""


def geometry3k_reward_fn(prompt, completions, prompt_ids, completion_ids, answer, **kwargs) -> float:
    sol = extract_answer(completions).replace(" ", "")
    ans = str(answer).replace(" ", "")
    if not sol or not ans:
        return 0.0
    if sol.lower() == ans.lower():
        return 1.0
    try:
        return float(math_equal(strip_string(sol), strip_string(ans)))
    except Exception:
        return 0.0
