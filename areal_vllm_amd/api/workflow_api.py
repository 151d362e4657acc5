"""Rollout workflows and the asynchronous executor that runs them under staleness control.

Contract (what trainers and engines rely on; same as ``areal/api/workflow_api.py:31-323`` of the reference):

* ``RolloutWorkflow.arun_episode(engine, data)`` produces one trajectory batch (or ``None`` to drop it).
* At most ``max_concurrent_rollouts // dp`` episodes run at once, and the number of trajectories that exist
  (running + finished and not rejected) never exceeds ``(eta + version + 1) * (consumer_batch_size // dp)``:
  with ``eta = max_head_offpolicyness`` no sample is more than ``eta`` weight versions old when it is trained on.
* ``wait(count)`` hands out the ``count`` OLDEST accepted trajectories (in shuffled order) as one padded batch;
  a trajectory that ``should_accept`` refuses gives its slot back. ``prepare_batch`` keeps the pipeline fed from
  a dataloader, ``rollout_batch`` is the synchronous submit-all / wait-all form, ``pause`` stops NEW episodes.

Design: nothing polls. One thread owns an asyncio loop; every change that can open capacity (a submission, an
episode ending, a rejection, ``resume``, a version bump) schedules ``_admit`` on that loop, which starts as many
queued episodes as the two limits allow. Each episode task carries a done-callback that books the result and
notifies the condition variable ``wait`` sleeps on. The reference loops with fixed sleeps instead
(``workflow_api.py:122-210``).
"""
import asyncio
import collections
import itertools
import random
import threading
import time
import traceback
from typing import TYPE_CHECKING, Any, Callable, Deque, Dict, List, Optional, Tuple

import torch.distributed as dist

from ..utils import logging
from ..utils.data import concat_padded_tensors
from .cli_args import InferenceEngineConfig
from .io_struct import RolloutStat

if TYPE_CHECKING:
    from .engine_api import InferenceEngine


class RolloutWorkflow:
    async def arun_episode(self, engine: "InferenceEngine", data: Dict[str, Any]) -> Optional[Dict[str, Any]]:
        """Run one episode; return a padded batch (dict of [n, S] tensors) or None to drop it."""
        raise NotImplementedError()


def _to_padded_batch(result):
    """Agent-style workflows return {completion_id: completion-with-reward}: one padded row per completion."""
    if isinstance(result, dict) and result and all(hasattr(v, "to_tensor_dict") for v in result.values()):
        return concat_padded_tensors([v.to_tensor_dict() for v in result.values()])
    return result


class StalenessGate:
    """The two admission limits. `room` says how many more episodes may start given the number running now,
    the trajectories that already count against the allowance and the current weight version."""

    def __init__(self, config: InferenceEngineConfig):
        self.config = config
        self.dp = 1

    def room(self, live: int, counted: int, version: int) -> int:
        c = self.config
        slots = max(1, c.max_concurrent_rollouts // self.dp) - live
        allowance = (c.max_head_offpolicyness + version + 1) * max(1, c.consumer_batch_size // self.dp)
        return min(slots, allowance - counted)


class WorkflowExecutor:
    def __init__(self, config: InferenceEngineConfig, inference_engine: "InferenceEngine"):
        config.max_concurrent_rollouts = config.max_concurrent_rollouts or config.consumer_batch_size
        self.config = config
        self.inference_engine = inference_engine
        self.paused = threading.Event()  # read by the engines' generate loops as well
        self.exiting = threading.Event()
        self.rollout_stat = RolloutStat()
        self.gate = StalenessGate(config)
        self._limit = config.queue_size or config.max_concurrent_rollouts * 16
        self._mutex = threading.Lock()  # guards the counters and both containers below
        self._backlog: Deque[Tuple[Dict[str, Any], RolloutWorkflow]] = collections.deque()
        self._finished: List[Tuple[int, Optional[Dict[str, Any]]]] = []  # (start time, trajectory)
        self._ready = threading.Condition(self._mutex)  # signalled whenever `_finished` grows
        self._kept: List[Tuple[int, Dict[str, Any]]] = []  # accepted, not yet handed out (consumer side only)
        self._loop: Optional[asyncio.AbstractEventLoop] = None
        self._thread: Optional[threading.Thread] = None
        self._started = threading.Event()
        self._live: Dict[int, asyncio.Task] = {}
        self._serial = itertools.count()
        self._feed = None

    # ------------------------------------------------------------------------------ life cycle
    def initialize(self, logger=None, train_data_parallel_size: Optional[int] = None):
        self.logger = logger or logging.getLogger("WorkflowExecutor")
        if train_data_parallel_size is not None:
            self.gate.dp = train_data_parallel_size
        else:
            self.gate.dp = dist.get_world_size() if dist.is_initialized() else 1
        self._thread = threading.Thread(target=self._serve, daemon=True, name="rollout-executor")
        self._thread.start()
        self._started.wait()

    def _serve(self):
        self._loop = asyncio.new_event_loop()
        asyncio.set_event_loop(self._loop)
        self._started.set()
        try:
            self._loop.run_forever()
            pending = [t for t in self._live.values() if not t.done()]
            for t in pending:
                t.cancel()
            if pending:
                self._loop.run_until_complete(asyncio.gather(*pending, return_exceptions=True))
        except Exception:
            traceback.print_exc()
        finally:
            self._loop.close()

    def destroy(self):
        self.exiting.set()
        with self._ready:
            self._ready.notify_all()
        if self._loop is not None and self._thread is not None and self._thread.is_alive():
            self._loop.call_soon_threadsafe(self._loop.stop)
            self._thread.join(timeout=10)

    def _poke(self):
        """Ask the loop thread to re-evaluate admission (safe from any thread)."""
        if self._loop is not None and not self.exiting.is_set():
            try:
                self._loop.call_soon_threadsafe(self._admit)
            except RuntimeError:  # loop already closed
                pass

    # ------------------------------------------------------------------------ staleness control
    def _room(self) -> int:
        """Episodes that may start now. Caller holds `_mutex`."""
        stat = self.rollout_stat
        return self.gate.room(len(self._live), stat.accepted + stat.running, self.inference_engine.get_version())

    @property
    def dp_world_size(self) -> int:
        return self.gate.dp

    def get_capacity(self) -> int:
        with self._mutex:
            return self._room()

    # ------------------------------------------------------------------- loop-thread callbacks
    def _admit(self):
        if self.exiting.is_set() or self.paused.is_set():
            return
        started = []
        with self._mutex:
            room = self._room()
            while room > 0 and self._backlog:
                data, workflow = self._backlog.popleft()
                sid = next(self._serial)
                task = self._loop.create_task(workflow.arun_episode(self.inference_engine, data), name=f"episode-{sid}")
                self._live[sid] = task
                self.rollout_stat.submitted += 1
                self.rollout_stat.running += 1
                room -= 1
                started.append((sid, task))
        for sid, task in started:
            t_start = time.monotonic_ns()
            task.add_done_callback(lambda t, sid=sid, t_start=t_start: self._episode_ended(sid, t_start, t))
            if self.config.enable_rollout_tracing:
                self.logger.info(f"episode {sid} started ({self._stat_line()})")

    def _episode_ended(self, sid: int, t_start: int, task: asyncio.Task):
        traj = None
        if not task.cancelled():
            err = task.exception()
            if err is None:
                try:
                    traj = _to_padded_batch(task.result())
                except Exception:
                    traceback.print_exc()
            else:
                traceback.print_exception(type(err), err, err.__traceback__)
        with self._ready:
            self._live.pop(sid, None)
            self.rollout_stat.running -= 1
            self.rollout_stat.accepted += 1  # provisional: `wait` takes it back if the trajectory is refused
            if len(self._finished) >= self._limit:
                self.exiting.set()
                self._ready.notify_all()
                raise RuntimeError("Output queue full. Please increase queue_size.")
            self._finished.append((t_start, traj))
            self._ready.notify_all()
        if self.config.enable_rollout_tracing:
            self.logger.info(f"episode {sid} finished ({self._stat_line()})")
        self._admit()

    def _stat_line(self) -> str:
        s = self.rollout_stat
        return f"submitted {s.submitted}, running {s.running}, accepted {s.accepted}"

    # ------------------------------------------------------------------------------- public API
    def submit(self, data: Dict[str, Any], workflow: Optional[RolloutWorkflow] = None,
               workflow_builder: Optional[Callable] = None) -> None:
        if workflow is None:
            workflow = workflow_builder()
        with self._mutex:
            if len(self._backlog) >= self._limit:
                raise RuntimeError("Input queue full. Please increase queue_size.")
            self._backlog.append((data, workflow))
        self._poke()

    def wait(self, count: int, timeout: Optional[float] = None, should_accept: Optional[Callable] = None):
        deadline = time.monotonic() + (timeout if timeout else 7 * 24 * 3600.0)
        while True:
            with self._ready:
                fresh, self._finished = self._finished, []
            refused = 0
            for t_start, traj in fresh:
                if traj is not None and (should_accept is None or should_accept(traj)):
                    self._kept.append((t_start, traj))
                else:
                    refused += 1
            if refused:
                with self._mutex:
                    self.rollout_stat.accepted -= refused
                self._poke()
            if len(self._kept) >= count:
                break
            with self._ready:
                if self.exiting.is_set():
                    raise RuntimeError("Rollout engine is exiting, cannot wait for results.")
                if not self._finished:
                    left = deadline - time.monotonic()
                    if left <= 0 or not self._ready.wait(timeout=left):
                        if not self._finished:
                            raise TimeoutError(
                                f"Timed out waiting for {count} rollouts, only received {len(self._kept)}.")
        self._kept.sort(key=lambda item: item[0])
        picked, self._kept = self._kept[:count], self._kept[count:]
        random.shuffle(picked)
        return concat_padded_tensors([traj for _, traj in picked])

    def rollout_batch(self, data: List[Dict[str, Any]], workflow: Optional[RolloutWorkflow] = None,
                      workflow_builder: Optional[Callable] = None, should_accept: Optional[Callable] = None):
        for item in data:
            self.submit(item, workflow, workflow_builder)
        return self.wait(count=len(data), should_accept=should_accept)

    def prepare_batch(self, dataloader, workflow: Optional[RolloutWorkflow] = None,
                      workflow_builder: Optional[Callable] = None, should_accept: Optional[Callable] = None):
        """One training batch; before waiting, another dataloader batch is queued whenever the limits would let
        part of it start and the backlog has room, so generation never idles behind the trainer."""
        if self._feed is None:
            self._feed = itertools.cycle(dataloader)
        bs = dataloader.batch_size
        assert bs is not None
        while True:
            with self._mutex:
                refill = self._room() + bs > 0 and len(self._backlog) + bs < self._limit
            if refill:
                for item in next(self._feed):
                    self.submit(item, workflow=workflow, workflow_builder=workflow_builder)
            try:
                return self.wait(bs, timeout=1, should_accept=should_accept)
            except TimeoutError:
                continue

    def pause(self):
        self.paused.set()

    def resume(self):
        self.paused.clear()
        self._poke()

    def notify_version_changed(self):
        """A weight-version bump raises the staleness allowance: re-evaluate admission."""
        self._poke()
