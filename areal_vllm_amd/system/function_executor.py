"""One training step of the master worker: load data, run every model function call once, clean up.

All MFC coroutines of the data-flow graph start together; each blocks on the sequence buffer until its
inputs exist, so independent calls overlap and dependent ones follow their producers. A loader coroutine
keeps asking the data-parallel heads of the source model for new samples (dataset batches, or
trajectories pulled from rollout workers in asynchronous runs) until the step has finished. Behaviour of
``realhf/system/function_executor.py:24-224``.
"""
import asyncio
from typing import Dict, List, Optional

from ..api.alloc_mode import ParallelStrategy
from ..utils import logging, name_resolve, names
from .buffer import AsyncIOSequenceBuffer, BufferFull
from .data_api import DataBatchMeta, SequenceSample
from .dfg import MFCDef, ModelName
from .model_function_call import ModelFunctionCall, RPCCoroutineControl, model_workers_of
from .redistributor import GlobalStorageTracker, RedistribPlanner
from .streams import NameResolvingRequestClient
from .system_api import ModelShardID

logger = logging.getLogger("function-executor")


class FunctionExecutor:
    def __init__(self, rpcs: List[MFCDef], msid2mwid: Dict[ModelShardID, int], stream: NameResolvingRequestClient,
                 buffer: AsyncIOSequenceBuffer, model_topos: Dict[ModelName, ParallelStrategy],
                 ctrl: RPCCoroutineControl, n_model_workers: int, experiment_name: str = "", trial_name: str = "",
                 fetch_interval: float = 0.05):
        self.rpcs = rpcs
        self.src_rpc = [r for r in rpcs if r.is_src][0]
        self.stream, self.buffer, self.ctrl = stream, buffer, ctrl
        self.n_model_workers = n_model_workers
        self.tracker = GlobalStorageTracker(n_model_workers)
        self.planner = RedistribPlanner(self.tracker)
        self.func_calls: Dict[str, ModelFunctionCall] = {
            r.name: ModelFunctionCall(r, stream, msid2mwid, model_topos, buffer, self.planner, ctrl, n_model_workers)
            for r in rpcs}
        _, _, self.data_heads = model_workers_of(self.src_rpc.model_name, model_topos[self.src_rpc.model_name],
                                                 msid2mwid)
        self.experiment_name, self.trial_name = experiment_name, trial_name
        self.fetch_interval = fetch_interval
        self.epoch_done = False
        self.n_fetched = 0
        self._ignore_ids = set()

    def ignore_ids(self, ids):
        """Sample ids consumed before a restart: fetched again by the re-created dataloaders, dropped here."""
        self._ignore_ids = {str(i) for i in ids}

    # -------------------------------------------------------------------------------- data
    async def load_data(self) -> int:
        """One fetch round over the data heads; returns the number of new samples."""
        loop = asyncio.get_running_loop()
        metas: List[DataBatchMeta] = await loop.run_in_executor(
            None, lambda: self.stream.call(self.data_heads, "fetch", [None] * len(self.data_heads)))
        n = 0
        for head, m in zip(self.data_heads, metas):
            if m is None or m.meta_sample is None:
                continue
            if m.is_final_batch:
                self.epoch_done = True
            items = m.meta_sample.unpack()
            drop = [x for x in items if x.ids[0] in self._ignore_ids]
            keep = [x for x in items if x.ids[0] not in self._ignore_ids]
            if drop:
                self._ignore_ids -= {x.ids[0] for x in drop}
                await loop.run_in_executor(None, lambda d=drop, h=head: self.stream.call(
                    [h], "clear_data_cache", [[x.ids[0] for x in d]]))
            if not keep:
                continue
            try:
                await self.buffer.put_batch(keep)
            except BufferFull:
                logger.warning("sequence buffer full: dropping a fetched batch")
                continue
            for x in keep:
                self.tracker.add_data(head, x.ids, list(x.keys))
                self.ctrl.used_ids_this_epoch.add(x.ids[0])
            n += len(keep)
        if n and self.experiment_name:
            self.n_fetched += n
            # the generation-server manager derives staleness from the number of samples the trainer took
            name_resolve.add(names.training_samples(self.experiment_name, self.trial_name), str(self.n_fetched),
                             replace=True)
        return n

    async def _load_loop(self):
        """Fetch until the source call of this step has taken its batch. Nothing is fetched ahead of the
        step: sample ids repeat from epoch to epoch, so a sample of the next epoch must not enter the
        buffer (or a worker's storage) while its previous incarnation is still being processed."""
        src = self.func_calls[self.src_rpc.name]
        while not src.batch_taken.is_set():
            if self.buffer.n_ready(self.src_rpc.name) < self.src_rpc.n_seqs:
                n = await self.load_data()
                if n == 0:
                    await asyncio.sleep(self.fetch_interval)
            else:
                await asyncio.sleep(self.fetch_interval)

    async def clear_consumed(self):
        ids = sorted(self.ctrl.ids_to_clear)
        if not ids:
            return
        loop = asyncio.get_running_loop()
        ws = list(range(self.n_model_workers))
        await loop.run_in_executor(None, lambda: self.stream.call(ws, "clear_data_cache", [ids] * len(ws)))
        self.tracker.clear_data(ids)
        self.ctrl.ids_to_clear.clear()

    # -------------------------------------------------------------------------------- step
    async def execute_step_async(self) -> Dict:
        self.ctrl.step_stats = {}
        for fc in self.func_calls.values():
            fc.batch_taken.clear()
        loader = asyncio.create_task(self._load_loop())
        calls = [asyncio.create_task(fc.run()) for fc in self.func_calls.values()]
        all_calls = asyncio.ensure_future(asyncio.gather(*calls))
        try:
            # the loader ends once the source call is served (or by raising); the step ends with the calls
            await asyncio.wait([all_calls, loader], return_when=asyncio.FIRST_COMPLETED)
            if loader.done():
                loader.result()
            await all_calls
        finally:
            if not all_calls.done():
                all_calls.cancel()
            loader.cancel()
            for c in calls:
                if not c.done():
                    c.cancel()
            try:
                await loader
            except (asyncio.CancelledError, Exception):
                pass
        await self.clear_consumed()
        return dict(self.ctrl.step_stats)

    def execute_step(self) -> Dict:
        try:
            loop = asyncio.get_event_loop()
        except RuntimeError:
            loop = asyncio.new_event_loop()
            asyncio.set_event_loop(loop)
        return loop.run_until_complete(self.execute_step_async())
