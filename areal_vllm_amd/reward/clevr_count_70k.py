"""CLEVR counting reward: the last number inside the last [bracketed] span must equal the label
(``areal/reward/clevr_count_70k.py:1-27``)."""
import re


def extract_answer(pred_str: str, data_name=None, use_last_number: bool = True) -> str:
    spans = re.findall(r"\[([^\]]+)\]", pred_str)
    if not spans:
        return ""
    nums = re.findall(r"-?\d+(?:\.\d+)?", spans[-1])
    return nums[-1] if nums else ""


def clevr_count_70k_reward_fn(prompt, completions, prompt_ids, completion_ids, answer, **kwargs) -> float:
    sol = extract_answer(completions)
    ans = str(answer).strip()
    if not sol or not ans:
        return 0.0
    try:
        return float(float(sol) == float(ans))
    except ValueError:
        return float(sol == ans)
