"""Frequency control for saving / evaluation / recover dumps (epoch-, step- and time-based triggers).

Counterpart of ``areal/utils/timeutil.py`` in the reference (``FrequencyControl``,
``EpochStepTimeFreqCtl``); the time check is agreed across ranks with one MAX all-reduce.
"""
import dataclasses
import time
from typing import Optional

import torch
import torch.distributed as dist


class FrequencyControl:
    """check() returns True every `frequency_steps` calls or `frequency_seconds` seconds (whichever first)."""

    @dataclasses.dataclass
    class State:
        start_time: float
        last_time: float
        last_step: int
        total_steps: int
        interval_steps: int

    def __init__(self, frequency_seconds: Optional[float] = None, frequency_steps: Optional[int] = None,
                 initial_value: bool = False):
        self.frequency_seconds = frequency_seconds
        self.frequency_steps = frequency_steps
        now = time.monotonic()
        self.__start_time = now
        self.__last_time = now
        self.__steps = 0
        self.__last_steps = 0
        self.__interval_steps = 0
        self.__initial_value = initial_value

    @property
    def total_seconds(self):
        return time.monotonic() - self.__start_time

    @property
    def total_steps(self):
        return self.__steps

    @property
    def interval_seconds(self):
        return time.monotonic() - self.__last_time

    @property
    def interval_steps(self):
        return self.__steps - self.__last_steps

    def state_dict(self):
        return dict(start_offset=time.monotonic() - self.__start_time, last_offset=time.monotonic() - self.__last_time,
                    steps=self.__steps, last_steps=self.__last_steps)

    def load_state_dict(self, sd):
        now = time.monotonic()
        self.__start_time = now - sd["start_offset"]
        self.__last_time = now - sd["last_offset"]
        self.__steps = sd["steps"]
        self.__last_steps = sd["last_steps"]

    def check(self, steps: int = 1) -> bool:
        self.__steps += steps
        if self.__initial_value:
            self.__initial_value = False
            self.__last_time = time.monotonic()
            self.__last_steps = self.__steps
            return True
        by_steps = self.frequency_steps is not None and (self.__steps - self.__last_steps) >= self.frequency_steps
        by_time = self.frequency_seconds is not None and self.interval_seconds >= self.frequency_seconds
        if dist.is_initialized() and self.frequency_seconds is not None:
            dev = "cuda" if "nccl" in str(dist.get_backend()) else "cpu"
            flag = torch.tensor([int(by_time)], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            by_time = bool(flag.item())
        if by_steps or by_time:
            self.__last_time = time.monotonic()
            self.__last_steps = self.__steps
            return True
        return False

    def reset_time(self):
        self.__last_time = time.monotonic()


class EpochStepTimeFreqCtl:
    """Trigger on epoch boundaries, every N steps or every T seconds."""

    def __init__(self, freq_epoch: Optional[int], freq_step: Optional[int], freq_sec: Optional[float]):
        self.freq_epoch, self.freq_step, self.freq_sec = freq_epoch, freq_step, freq_sec
        self.epoch_ctl = FrequencyControl(frequency_steps=freq_epoch)
        self.step_ctl = FrequencyControl(frequency_steps=freq_step)
        self.time_ctl = FrequencyControl(frequency_seconds=freq_sec)

    def state_dict(self):
        return dict(epoch_ctl=self.epoch_ctl.state_dict(), step_ctl=self.step_ctl.state_dict(),
                    time_ctl=self.time_ctl.state_dict())

    def load_state_dict(self, sd):
        self.epoch_ctl.load_state_dict(sd["epoch_ctl"])
        self.step_ctl.load_state_dict(sd["step_ctl"])
        self.time_ctl.load_state_dict(sd["time_ctl"])

    def check(self, epochs: int, steps: int) -> bool:
        e = self.epoch_ctl.check(steps=epochs)
        s = self.step_ctl.check(steps=steps)
        t = self.time_ctl.check()
        if e:
            self.time_ctl.reset_time()
            self.step_ctl.reset_time()
        if s:
            self.epoch_ctl.reset_time()
            self.time_ctl.reset_time()
        if t:
            self.epoch_ctl.reset_time()
            self.step_ctl.reset_time()
        return e or s or t
