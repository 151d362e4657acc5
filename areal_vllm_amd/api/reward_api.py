"""Asynchronous reward computation in a shared process pool.

Behaviour of the reference's ``AsyncRewardWrapper`` (``areal/api/reward_api.py``): reward functions are
plain (possibly slow, possibly crashing) Python callables; they run in a process pool shared by all
wrappers, a call that exceeds ``timeout_seconds`` yields reward 0, and a broken pool is rebuilt.
"""
import asyncio
import atexit
import threading
from concurrent.futures import ProcessPoolExecutor
from concurrent.futures.process import BrokenProcessPool
from typing import Callable, Optional

from ..utils import logging

logger = logging.getLogger("reward")


class AsyncRewardWrapper:
    _executor: Optional[ProcessPoolExecutor] = None
    _lock = threading.Lock()
    _instances = 0
    _max_workers = 4

    def __init__(self, reward_fn: Callable, timeout_seconds: float = 15.0, max_workers: Optional[int] = None,
                 max_retries: int = 3):
        self.reward_fn = reward_fn
        self.timeout_seconds = timeout_seconds
        self.max_retries = max_retries
        with AsyncRewardWrapper._lock:
            if max_workers:
                AsyncRewardWrapper._max_workers = max_workers
            AsyncRewardWrapper._instances += 1
            self._ensure_pool()

    @classmethod
    def _ensure_pool(cls):
        if cls._executor is None:
            import multiprocessing as mp

            cls._executor = ProcessPoolExecutor(max_workers=cls._max_workers, mp_context=mp.get_context("spawn"))
            atexit.register(cls.shutdown)

    @classmethod
    def _rebuild_pool(cls):
        with cls._lock:
            if cls._executor is not None:
                cls._executor.shutdown(wait=False, cancel_futures=True)
                cls._executor = None
            cls._ensure_pool()

    @classmethod
    def shutdown(cls):
        with cls._lock:
            if cls._executor is not None:
                cls._executor.shutdown(wait=False, cancel_futures=True)
                cls._executor = None

    async def __call__(self, *args, **kwargs) -> float:
        loop = asyncio.get_running_loop()
        for attempt in range(self.max_retries + 1):
            try:
                with AsyncRewardWrapper._lock:
                    self._ensure_pool()
                    ex = AsyncRewardWrapper._executor
                fut = loop.run_in_executor(ex, _call, self.reward_fn, args, kwargs)
                return await asyncio.wait_for(fut, timeout=self.timeout_seconds)
            except asyncio.TimeoutError:
                logger.warning(f"reward function timed out after {self.timeout_seconds}s, returning 0")
                return 0
            except BrokenProcessPool:
                logger.warning("reward process pool broke; rebuilding")
                self._rebuild_pool()
                if attempt == self.max_retries:
                    return 0
            except Exception as e:  # reward code raised
                logger.warning(f"reward function raised {type(e).__name__}: {e}; returning 0")
                return 0
        return 0


def _call(fn, args, kwargs):
    return fn(*args, **kwargs)
