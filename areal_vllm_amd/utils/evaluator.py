"""Periodic evaluation: runs the caller's evaluation closure when the configured epoch / step / time
frequency says so (role of ``areal/utils/evaluator.py`` in the reference)."""
from typing import Any, Callable, Optional

from .periodic import PeriodicTrigger


class Evaluator(PeriodicTrigger):
    def evaluate(self, evaluate_fn: Callable[[], Any], epoch: int, step: int, global_step: int) -> Optional[Any]:
        return evaluate_fn() if self.due(step) else None
