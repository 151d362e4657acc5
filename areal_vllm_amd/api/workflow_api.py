"""Rollout workflow interface and the asynchronous rollout executor with staleness control.

Semantics follow ``areal/api/workflow_api.py:31-323`` in the reference:

* ``capacity = min(max(1, max_concurrent_rollouts // dp) - running,
                   (eta + version + 1) * max(1, consumer_batch_size // dp) - (accepted + running))``
* ``wait`` pops finished trajectories, rejected ones release their slot, accepted ones are returned oldest
  first (then shuffled) and concatenated; ``prepare_batch`` keeps at least one extra batch in flight.

The executor is event-driven (the reference sleeps one second per loop iteration): the rollout thread
blocks on ``asyncio.wait`` with a short timeout only while tasks are in flight, is woken by submissions,
and results are handed to ``wait`` through a condition variable.
"""
import asyncio
import itertools
import queue
import random
import threading
import time
import traceback
from dataclasses import dataclass
from typing import TYPE_CHECKING, Any, Callable, Dict, List, Optional

import torch.distributed as dist

from ..utils import logging
from ..utils.data import concat_padded_tensors
from .cli_args import InferenceEngineConfig
from .io_struct import RolloutStat

if TYPE_CHECKING:
    from .engine_api import InferenceEngine

ROLLOUT_POLL_WAIT_TIME = 0.05


def _as_batch(traj):
    """Episodes may return {completion_id: CompletionWithTokenLogpReward} (OpenAI-style agent
    workflows): flatten them into one padded batch, one row per stored completion."""
    if isinstance(traj, dict) and traj and all(hasattr(v, "to_tensor_dict") for v in traj.values()):
        from ..utils.data import concat_padded_tensors

        return concat_padded_tensors([v.to_tensor_dict() for v in traj.values()])
    return traj


class RolloutWorkflow:
    async def arun_episode(self, engine: "InferenceEngine", data: Dict[str, Any]) -> Optional[Dict[str, Any]]:
        """Run one episode; return a padded batch (dict of [n, S] tensors) or None to drop it."""
        raise NotImplementedError()


@dataclass
class _TimedResult:
    t: int
    data: Optional[Dict[str, Any]]


class WorkflowExecutor:
    def __init__(self, config: InferenceEngineConfig, inference_engine: "InferenceEngine"):
        config.max_concurrent_rollouts = config.max_concurrent_rollouts or config.consumer_batch_size
        self.config = config
        self.exiting = threading.Event()
        self.paused = threading.Event()
        self.lock = threading.Lock()
        self.inference_engine = inference_engine
        qsize = config.queue_size or config.max_concurrent_rollouts * 16
        self.input_queue: "queue.Queue" = queue.Queue(maxsize=qsize)
        self.output_queue: "queue.Queue" = queue.Queue(maxsize=qsize)
        self.result_cache: List[_TimedResult] = []
        self.rollout_stat = RolloutStat()
        self._out_cv = threading.Condition()
        self._wake = threading.Event()
        self.rollout_tasks: Dict[str, asyncio.Task] = {}
        self.rollout_thread: Optional[threading.Thread] = None
        self.dp_world_size = 1

    def initialize(self, logger=None, train_data_parallel_size: Optional[int] = None):
        self.logger = logger or logging.getLogger("WorkflowExecutor")
        if train_data_parallel_size is not None:
            self.dp_world_size = train_data_parallel_size
        elif dist.is_initialized():
            self.dp_world_size = dist.get_world_size()
        else:
            self.dp_world_size = 1
        self.rollout_thread = threading.Thread(target=self._rollout_thread, daemon=True)
        self.rollout_thread.start()

    def destroy(self):
        self.exiting.set()
        self._wake.set()
        if self.rollout_thread is not None:
            self.rollout_thread.join(timeout=10)

    # ------------------------------------------------------------------ staleness control
    def get_capacity(self) -> int:
        with self.lock:
            max_concurrent = max(1, self.config.max_concurrent_rollouts // self.dp_world_size)
            capacity = max_concurrent - len(self.rollout_tasks)
            version = self.inference_engine.get_version()
            ofp = self.config.max_head_offpolicyness
            sample_cnt = self.rollout_stat.accepted + self.rollout_stat.running
            consumer_bs = max(1, self.config.consumer_batch_size // self.dp_world_size)
            capacity = min(capacity, (ofp + version + 1) * consumer_bs - sample_cnt)
        return capacity

    # ------------------------------------------------------------------ rollout thread
    def _rollout_thread(self):
        try:
            asyncio.run(self._rollout_thread_async())
        except Exception:
            traceback.print_exc()

    async def _rollout_thread_async(self):
        rollout_tasks = self.rollout_tasks
        create_time: Dict[str, int] = {}
        rid = 0
        try:
            while not self.exiting.is_set():
                capacity = self.get_capacity()
                with self.lock:
                    while capacity > 0 and not self.paused.is_set() and self.input_queue.qsize() > 0:
                        data, workflow = self.input_queue.get_nowait()
                        task = asyncio.create_task(workflow.arun_episode(self.inference_engine, data), name=str(rid))
                        rollout_tasks[str(rid)] = task
                        create_time[str(rid)] = time.monotonic_ns()
                        self.rollout_stat.submitted += 1
                        self.rollout_stat.running += 1
                        if self.config.enable_rollout_tracing:
                            self.logger.info(f"Submit rollout rid {rid}. Submit: {self.rollout_stat.submitted}, "
                                             f"running: {self.rollout_stat.running}, "
                                             f"accepted: {self.rollout_stat.accepted}.")
                        capacity -= 1
                        rid += 1
                    tasks = list(rollout_tasks.values())
                done = []
                if tasks:
                    done, _ = await asyncio.wait(tasks, timeout=ROLLOUT_POLL_WAIT_TIME,
                                                 return_when=asyncio.FIRST_COMPLETED)
                else:
                    # idle: sleep until a submission / resume / version bump wakes us (bounded)
                    self._wake.clear()
                    await asyncio.get_running_loop().run_in_executor(None, self._wake.wait, ROLLOUT_POLL_WAIT_TIME)
                for task in done:
                    try:
                        traj = _as_batch(await task)
                    except Exception:
                        traceback.print_exc()
                        traj = None
                    task_rid = task.get_name()
                    with self.lock:
                        rollout_tasks.pop(task_rid)
                        ct = create_time.pop(task_rid)
                        self.rollout_stat.accepted += 1
                        self.rollout_stat.running -= 1
                        if self.config.enable_rollout_tracing:
                            self.logger.info(f"Finish rollout {task_rid}. Submit: {self.rollout_stat.submitted}, "
                                             f"running: {self.rollout_stat.running}, "
                                             f"accepted: {self.rollout_stat.accepted}.")
                    try:
                        self.output_queue.put_nowait(_TimedResult(ct, traj))
                    except queue.Full:
                        raise RuntimeError("Output queue full. Please increase queue_size.")
                    with self._out_cv:
                        self._out_cv.notify_all()
        except Exception:
            traceback.print_exc()
        finally:
            for task in list(rollout_tasks.values()):
                if not task.done():
                    task.cancel()
                    try:
                        await task
                    except (asyncio.CancelledError, Exception):
                        pass

    # ------------------------------------------------------------------ public API
    def submit(self, data: Dict[str, Any], workflow: Optional[RolloutWorkflow] = None,
               workflow_builder: Optional[Callable] = None) -> None:
        try:
            if workflow is None:
                workflow = workflow_builder()
            self.input_queue.put_nowait((data, workflow))
            self._wake.set()
        except queue.Full:
            raise RuntimeError("Input queue full. Please increase queue_size.")

    def wait(self, count: int, timeout: Optional[float] = None, should_accept: Optional[Callable] = None):
        tik = time.perf_counter()
        timeout = timeout or float(7 * 24 * 3600)
        while not self.exiting.is_set() and time.perf_counter() - tik < timeout:
            while True:
                try:
                    r = self.output_queue.get_nowait()
                except queue.Empty:
                    break
                if r.data is not None and (should_accept is None or should_accept(r.data)):
                    self.result_cache.append(r)
                else:
                    with self.lock:
                        self.rollout_stat.accepted -= 1
                    self._wake.set()
            if len(self.result_cache) >= count:
                break
            with self._out_cv:
                remaining = timeout - (time.perf_counter() - tik)
                self._out_cv.wait(timeout=max(0.0, min(ROLLOUT_POLL_WAIT_TIME, remaining)))
        if self.exiting.is_set():
            raise RuntimeError("Rollout engine is exiting, cannot wait for results.")
        if len(self.result_cache) < count:
            raise TimeoutError(f"Timed out waiting for {count} rollouts, only received {len(self.result_cache)}.")
        self.result_cache.sort(key=lambda x: x.t)
        results, self.result_cache = self.result_cache[:count], self.result_cache[count:]
        random.shuffle(results)
        return concat_padded_tensors([r.data for r in results])

    def rollout_batch(self, data: List[Dict[str, Any]], workflow: Optional[RolloutWorkflow] = None,
                      workflow_builder: Optional[Callable] = None, should_accept: Optional[Callable] = None):
        for item in data:
            self.submit(item, workflow, workflow_builder)
        return self.wait(count=len(data), should_accept=should_accept)

    def prepare_batch(self, dataloader, workflow: Optional[RolloutWorkflow] = None,
                      workflow_builder: Optional[Callable] = None, should_accept: Optional[Callable] = None):
        if not hasattr(self, "data_generator"):
            self.data_generator = itertools.cycle(dataloader)
        bs = dataloader.batch_size
        assert bs is not None
        while True:
            if self.get_capacity() + bs > 0 and self.input_queue.qsize() + bs < self.input_queue.maxsize:
                for item in next(self.data_generator):
                    self.submit(item, workflow=workflow, workflow_builder=workflow_builder)
            try:
                return self.wait(bs, timeout=1, should_accept=should_accept)
            except TimeoutError:
                pass

    def pause(self):
        self.paused.set()

    def resume(self):
        self.paused.clear()
        self._wake.set()

    def notify_version_changed(self):
        """Wake the rollout thread after a weight-version bump (capacity may have grown)."""
        self._wake.set()
