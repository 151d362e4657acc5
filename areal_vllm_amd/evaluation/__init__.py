"""Offline evaluation of checkpoints (math answer accuracy, code unit tests) with the native rollout
engine, and the watcher that evaluates every newly saved checkpoint of a trial. Role of the reference's
``evaluation/`` scripts and ``realhf/scheduler/evaluator.py``."""
from .evaluator import AutomaticEvaluator  # noqa: F401
from .math_eval import evaluate_checkpoint, pass_at_k  # noqa: F401
