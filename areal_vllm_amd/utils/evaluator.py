"""Frequency-controlled evaluation trigger (``areal/utils/evaluator.py`` in the reference)."""
from typing import Callable

from ..api.cli_args import EvaluatorConfig
from ..api.io_struct import FinetuneSpec
from . import timeutil


class Evaluator:
    def __init__(self, config: EvaluatorConfig, ft_spec: FinetuneSpec):
        self.config = config
        self.ft_spec = ft_spec
        self.freq_ctl = timeutil.EpochStepTimeFreqCtl(freq_epoch=config.freq_epochs, freq_step=config.freq_steps,
                                                      freq_sec=config.freq_secs)

    def state_dict(self):
        return self.freq_ctl.state_dict()

    def load_state_dict(self, state_dict):
        self.freq_ctl.load_state_dict(state_dict)

    def evaluate(self, evaluate_fn: Callable, epoch: int, step: int, global_step: int):
        if not self.freq_ctl.check(epochs=int(step == self.ft_spec.steps_per_epoch - 1), steps=1):
            return None
        return evaluate_fn()
