"""Reward of the countdown task (role of ``examples/countdown/reward_score.py`` in the reference):
1.0 for an equation inside the last ``<answer>`` block that uses exactly the given numbers and evaluates
to the target, `format_score` when an answer block exists but is wrong, 0 without one."""
import ast
import operator
import re
from typing import List, Optional

_OPS = {ast.Add: operator.add, ast.Sub: operator.sub, ast.Mult: operator.mul, ast.Div: operator.truediv}


def extract_equation(text: str) -> Optional[str]:
    m = re.findall(r"<answer>(.*?)</answer>", text, re.DOTALL)
    return m[-1].strip() if m else None


def _eval(node):
    """Arithmetic-only evaluator over the parsed expression (no names, calls or attribute access)."""
    if isinstance(node, ast.Expression):
        return _eval(node.body)
    if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)):
        return node.value
    if isinstance(node, ast.BinOp) and type(node.op) in _OPS:
        return _OPS[type(node.op)](_eval(node.left), _eval(node.right))
    if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
        v = _eval(node.operand)
        return -v if isinstance(node.op, ast.USub) else v
    raise ValueError("not an arithmetic expression")


def evaluate_equation(eq: str) -> Optional[float]:
    if not re.fullmatch(r"[\d+\-*/().\s]+", eq or ""):
        return None
    try:
        return float(_eval(ast.parse(eq, mode="eval")))
    except Exception:
        return None


def uses_numbers_once(eq: str, numbers: List[int]) -> bool:
    return sorted(int(n) for n in re.findall(r"\d+", eq)) == sorted(int(n) for n in numbers)


def compute_score(completion: str, numbers: List[int], target: float, format_score: float = 0.1,
                  score: float = 1.0) -> float:
    eq = extract_equation(completion)
    if eq is None:
        return 0.0
    if not uses_numbers_once(eq, numbers):
        return format_score
    val = evaluate_equation(eq)
    if val is None or abs(val - float(target)) > 1e-5:
        return format_score
    return score


def countdown_reward_fn(prompt, completions, prompt_ids, completion_ids, numbers=None, target=None, **kwargs) -> float:
    return compute_score(completions or "", numbers or [], target if target is not None else float("nan"))
