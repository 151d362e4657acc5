"""Things that fire every so many epochs / steps / seconds of training (checkpoints, evaluations)."""
from typing import Any, Dict

from ..api.io_struct import FinetuneSpec
from . import timeutil


class PeriodicTrigger:
    """Holds the frequency control of one periodic action. `due(step)` is called once per training step with
    the step's index INSIDE its epoch and answers whether the action runs now; the control's state is what
    recover checkpoints persist (``state_dict`` / ``load_state_dict``)."""

    def __init__(self, config, ft_spec: FinetuneSpec):
        self.config = config
        self.ft_spec = ft_spec
        self.freq_ctl = timeutil.EpochStepTimeFreqCtl(freq_epoch=config.freq_epochs, freq_step=config.freq_steps,
                                                      freq_sec=config.freq_secs)

    def due(self, step_in_epoch: int) -> bool:
        epoch_ends = step_in_epoch == self.ft_spec.steps_per_epoch - 1
        return bool(self.freq_ctl.check(epochs=1 if epoch_ends else 0, steps=1))

    def state_dict(self) -> Dict[str, Any]:
        return self.freq_ctl.state_dict()

    def load_state_dict(self, state: Dict[str, Any]) -> None:
        self.freq_ctl.load_state_dict(state)
