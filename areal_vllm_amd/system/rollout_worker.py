"""Rollout worker: turns prompts into trajectories with an agent, an environment and the generation fleet.

Per poll: take the next prompt of this worker's dataset shard, ask the generation-server manager for
permission (``/allocate_rollout`` enforces the concurrency cap and the staleness bound), start the agent's
episode as an asyncio task, advance the chunked-generation client, route finished generations back to the
waiting agents, push finished trajectories to the trainer side and report ``/finish_rollout``. An epoch
barrier keeps a worker from starting the next epoch while episodes of the current one are in flight.
Behaviour of ``realhf/system/rollout_worker.py:43-372``.
"""
import asyncio
import json
import os
import time
from typing import Dict, Hashable, List, Optional

import aiohttp
import numpy as np
import torch

from ..impl.dataset import make_dataset
from ..utils import datapack, logging, name_resolve, names, seeding
from .agent_api import make_agent, make_env
from .data_api import SequenceSample
from .partial_rollout import PartialRolloutManager
from .recover import load_recover_info
from .streams import NameResolvingPusher
from .system_api import RolloutWorkerConfig
from .worker_base import AsyncWorker, ExpStatus, PollResult

logger = logging.getLogger("rollout-worker")
ROLLOUT_POLL_WAIT_TIME = 0.1


class RolloutWorker(AsyncWorker):
    def _configure(self, config: RolloutWorkerConfig):
        self.config = config
        self.env = make_env(config.env)
        self.agent = make_agent(config.agent)
        self.rollout_request_queue: Optional[asyncio.Queue] = None
        self.rollout_response_queue: Optional[asyncio.Queue] = None
        self.act_queues: Dict[Hashable, asyncio.Queue] = {}
        self.rollout_tasks: Dict[Hashable, asyncio.Task] = {}
        self.inference_maker: Optional[PartialRolloutManager] = None
        self.push_stream: Optional[NameResolvingPusher] = None
        self.gserver_manager_addr: Optional[str] = None
        self.data_generator = None
        self.is_new_epoch = False
        self._cur_data: Optional[SequenceSample] = None
        self._http: Optional[aiohttp.ClientSession] = None
        self.n_submitted = self.n_accepted = 0
        seeding.set_random_seed(config.base_seed, f"rollout_worker{self._info.worker_index}")
        rdir = os.path.join(config.fileroot, "recover", self.experiment_name, self.trial_name)
        self.recover_run, self.recover_info = load_recover_info(rdir)
        self._ignore = set(self.recover_info.hash_vals_to_ignore) if self.recover_run else set()
        return self._info

    # -------------------------------------------------------------------------------- data
    def make_datasets(self):
        cfg = self.config
        sets = [make_dataset(d, cfg.base_seed, self.worker_index, self.worker_count, cfg.tokenizer_path)
                for d in cfg.datasets]
        self.dataset = sets[0] if len(sets) == 1 else torch.utils.data.ConcatDataset(sets)
        self._gen = torch.Generator().manual_seed(seeding.get_seed())
        self._new_epoch_iter()

    def _new_epoch_iter(self):
        order = torch.randperm(len(self.dataset), generator=self._gen).tolist()
        self.data_generator = iter(order)

    def load_next_data(self) -> Optional[SequenceSample]:
        if self.is_new_epoch:  # epoch barrier: wait for the running episodes of the finished epoch
            if self.rollout_tasks:
                return None
            self.is_new_epoch = False
        try:
            idx = next(self.data_generator)
        except StopIteration:
            self.is_new_epoch = True
            scores = os.path.join(self.config.fileroot, "logs", self.experiment_name, self.trial_name,
                                  "dataset_eval_scores.json")
            if hasattr(self.dataset, "filter") and os.path.exists(scores):
                with open(scores) as f:
                    self.dataset.filter(json.load(f))
            self._new_epoch_iter()
            return None
        sample = self.dataset[idx]
        if sample.ids[0] in self._ignore:  # consumed before the restart
            self._ignore.discard(sample.ids[0])
            return None
        if sample.ids[0] in self.rollout_tasks:
            return None
        return sample

    # -------------------------------------------------------------------------------- http
    async def _post(self, route: str, payload: Dict) -> Dict:
        if self._http is None or self._http.closed:
            to = aiohttp.ClientTimeout(total=self.config.rollout_request_timeout)
            self._http = aiohttp.ClientSession(timeout=to)
        async with self._http.post(f"http://{self.gserver_manager_addr}{route}", json=payload) as r:
            r.raise_for_status()
            return await r.json()

    async def allocate_new_rollout(self, qid) -> bool:
        res = await self._post("/allocate_rollout", dict(qid=qid))
        if not res["success"]:
            logger.debug(f"Cannot allocate new rollout because: {res['reason']}")
        return bool(res["success"])

    # -------------------------------------------------------------------------------- poll
    async def _poll_async(self) -> PollResult:
        try:
            return await self._poll_step()
        except (aiohttp.ClientError, ConnectionError, asyncio.TimeoutError):
            # the manager and the servers leave as soon as the experiment is over: a request cut off by
            # that is the normal end of this worker, anything else is a failure
            try:
                st = name_resolve.get(names.experiment_status(self.experiment_name, self.trial_name))
            except name_resolve.NameEntryNotFoundError:
                st = None
            if st == ExpStatus.RUNNING.value:
                raise
            self.exit()
            return PollResult(0, 0)

    async def _poll_step(self) -> PollResult:
        cfg = self.config
        if self.data_generator is None:
            self.make_datasets()
            self.rollout_request_queue, self.rollout_response_queue = asyncio.Queue(1024), asyncio.Queue(1024)
        try:
            st = name_resolve.wait(names.experiment_status(self.experiment_name, self.trial_name), timeout=300)
        except TimeoutError:
            raise TimeoutError("experiment status never appeared: the master worker is not running")
        if st != ExpStatus.RUNNING.value:
            self.exit()
            return PollResult(0, 0)
        if self.push_stream is None:
            self.push_stream = NameResolvingPusher(self.experiment_name, self.trial_name, self.worker_index,
                                                   self.worker_count)
        if self.gserver_manager_addr is None:
            self.gserver_manager_addr = name_resolve.wait(
                names.gen_server_manager(self.experiment_name, self.trial_name), timeout=300)
            self.inference_maker = PartialRolloutManager(
                worker_index=self.worker_index, request_queue=self.rollout_request_queue,
                reply_queue=self.rollout_response_queue, new_tokens_per_chunk=cfg.new_tokens_per_chunk,
                stop_token_ids=cfg.stop_token_ids, timeout=cfg.rollout_request_timeout,
                experiment_name=self.experiment_name, trial_name=self.trial_name,
                gserver_manager_addr=self.gserver_manager_addr)
        if self._cur_data is None:
            self._cur_data = self.load_next_data()
        if self._cur_data is not None:
            data = self._cur_data
            qid = data.ids[0]
            if await self.allocate_new_rollout(qid):
                self.act_queues[qid] = asyncio.Queue(1024)
                self.rollout_tasks[qid] = asyncio.create_task(self.rollout_task(qid, data))
                self._cur_data = None
                self.n_submitted += 1
        done, _, _ = await asyncio.gather(self.poll_rollout_task(), self.poll_queue_dispatch_task(),
                                          self.inference_maker.run_step())
        batch_count = sample_count = 0
        for task in done:
            qid, trajs = await task
            self.rollout_tasks.pop(qid)
            self.act_queues.pop(qid)
            accepted = len(trajs) > 0
            n_tokens = 0
            if accepted:
                self.push_stream.push([t.as_json_compatible() for t in trajs])
                self.n_accepted += 1
            for t in trajs:
                n_tokens += max(sum(datapack.flat2d(s)) for s in t.seqlens.values())
                batch_count += t.bs
            sample_count += n_tokens
            res = await self._post("/finish_rollout", dict(qid=qid, accepted=accepted, n_tokens=n_tokens))
            assert res["success"]
        return PollResult(sample_count=sample_count, batch_count=batch_count)

    async def rollout_task(self, qid, data):
        return qid, await self.agent.collect_trajectory(prompt=data, env=self.env, obs_queue=self.rollout_request_queue,
                                                        act_queue=self.act_queues[qid])

    async def poll_rollout_task(self):
        tasks = list(self.rollout_tasks.values())
        if not tasks:
            await asyncio.sleep(ROLLOUT_POLL_WAIT_TIME / 4)
            return []
        done, _ = await asyncio.wait(tasks, timeout=ROLLOUT_POLL_WAIT_TIME, return_when=asyncio.FIRST_COMPLETED)
        return list(done)

    def _act_queue_of(self, qid) -> asyncio.Queue:
        """Multi-turn agents tag their generation requests ``{qid}-{turn}``: route by the episode's qid."""
        q = self.act_queues.get(qid)
        if q is None and isinstance(qid, str) and "-" in qid:
            q = self.act_queues.get(qid.rsplit("-", 1)[0])
        if q is None:
            raise KeyError(f"generation result for unknown episode {qid}")
        return q

    async def poll_queue_dispatch_task(self):
        for _ in range(20):
            try:
                resp = self.rollout_response_queue.get_nowait()
                self._act_queue_of(resp.qid).put_nowait(resp)
            except asyncio.QueueEmpty:
                await asyncio.sleep(0.005)

    def _exit_hook(self, exit_status):
        if self.push_stream is not None:
            self.push_stream.close()
        loop = getattr(self, "_loop", None)
        if loop is None or loop.is_closed():
            return

        async def cleanup():
            for t in self.rollout_tasks.values():
                t.cancel()
            await asyncio.gather(*self.rollout_tasks.values(), return_exceptions=True)
            if self.inference_maker is not None:
                for g in self.inference_maker.gen_requests.values():
                    for t in g.values():
                        t.cancel()
                await self.inference_maker.close()
            if self._http is not None and not self._http.closed:
                await self._http.close()

        loop.run_until_complete(cleanup())
