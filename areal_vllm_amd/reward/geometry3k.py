"""Geometry3K reward: the LAST square-bracketed span of the completion is the answer; it scores 1 when it equals
the reference as a string (case and blanks ignored) or as a mathematical value (``areal/reward/geometry3k.py:1-29``)."""
import re

from .math_parser import math_equal, strip_string

_BRACKETED = re.compile(r"\[([^\]]+)\]")


def extract_answer(pred_str: str, data_name=None, use_last_number: bool = True) -> str:
    found = None
    for found in _BRACKETED.finditer(pred_str):
        pass
    return found.group(1).strip() if found is not None else ""


def _squash(text) -> str:
    return "".join(str(text).split())


def geometry3k_reward_fn(prompt, completions, prompt_ids, completion_ids, answer, **kwargs) -> float:
    given, wanted = _squash(extract_answer(completions)), _squash(answer)
    if "" in (given, wanted):
        return 0.0
    if given.casefold() == wanted.casefold():
        return 1.0
    try:
        same = math_equal(strip_string(given), strip_string(wanted))
    except Exception:  # the symbolic comparison may choke on malformed answers: that is a wrong answer
        return 0.0
    return 1.0 if same else 0.0
