"""Master-side driver of ONE model function call (MFC).

Per training step an MFC waits until the buffer holds `n_seqs` samples with all of its input keys, splits
them over the data-parallel replicas of its model (balanced by token count), asks the planner how the
inputs reach those replicas, posts the request (with the transfer plan as a pre-hook and weight
publishing / offload as post-hooks) to every worker of the model, gathers the replies of the
data-parallel heads and amends the buffer with the metadata of the produced keys. Behaviour of
``realhf/system/model_function_call.py:54-509``.
"""
import asyncio
import dataclasses
import time
from typing import Any, Dict, Hashable, List, Optional, Tuple

from ..api.alloc_mode import ParallelStrategy
from ..utils import datapack, logging
from .buffer import AsyncIOSequenceBuffer
from .data_api import SequenceSample
from .dfg import MFCDef, ModelInterfaceType, ModelName, OffloadHook, ParamReallocHook
from .redistributor import GlobalStorageTracker, RedistribPlanner
from .streams import NameResolvingRequestClient, Payload
from .system_api import ModelShardID

logger = logging.getLogger("mfc")


@dataclasses.dataclass
class RPCCoroutineControl:
    """State shared by the MFC coroutines of one master worker."""
    # serialises the posting of requests that carry transfer plans: every model worker must meet the
    # plans in the same order, or two overlapping point-to-point plans could wait on each other
    post_lock: asyncio.Lock = dataclasses.field(default_factory=asyncio.Lock)
    ids_to_clear: set = dataclasses.field(default_factory=set)
    used_ids_this_epoch: set = dataclasses.field(default_factory=set)
    step_stats: Dict[str, Any] = dataclasses.field(default_factory=dict)
    data_owner: Dict[Tuple[Hashable, str], int] = dataclasses.field(default_factory=dict)


def model_workers_of(model_name: ModelName, topo: ParallelStrategy, msid2mwid: Dict[ModelShardID, int]):
    """-> (all worker ids of the model, worker ids grouped by dp rank, dp-head worker id per dp rank).
    The head of a replica is its last pipeline stage, tensor rank 0 (it holds the outputs)."""
    by_dp: List[List[int]] = []
    heads: List[int] = []
    for d in range(topo.dp_size):
        ws = [msid2mwid[ModelShardID(model_name, d, t, p)] for p in range(topo.pp_size) for t in range(topo.tp_size)]
        by_dp.append(ws)
        heads.append(msid2mwid[ModelShardID(model_name, d, 0, topo.pp_size - 1)])
    return sorted({w for ws in by_dp for w in ws}), by_dp, heads


class ModelFunctionCall:
    def __init__(self, rpc: MFCDef, stream: NameResolvingRequestClient, msid2mwid: Dict[ModelShardID, int],
                 model_topos: Dict[ModelName, ParallelStrategy], buffer: AsyncIOSequenceBuffer,
                 planner: RedistribPlanner, ctrl: RPCCoroutineControl, n_model_workers: int):
        self.rpc = rpc
        self.stream = stream
        self.buffer = buffer
        self.planner = planner
        self.tracker: GlobalStorageTracker = planner.tracker
        self.ctrl = ctrl
        self.n_model_workers = n_model_workers
        self.topo = model_topos[rpc.model_name]
        self.workers, self.workers_by_dp, self.dp_heads = model_workers_of(rpc.model_name, self.topo, msid2mwid)
        # workers of the models this call's hooks hand weights to (they take part in the transfer)
        self.realloc_peers: Dict[ModelName, List[int]] = {
            h.target: model_workers_of(h.target, model_topos[h.target], msid2mwid)[0]
            for h in rpc._post_hooks if isinstance(h, ParamReallocHook) and h.target is not None}
        self.batch_taken = asyncio.Event()  # set once this step's batch has left the buffer

    # --------------------------------------------------------------------------- dispatch
    def data_parallel_dispatch(self, sample: SequenceSample) -> Tuple[List[List[int]], List[List[str]]]:
        """Partition sample indices over the dp replicas, balanced by tokens; train steps get (almost)
        equal sample counts so that every replica runs the same number of optimizer steps."""
        dp = self.topo.dp_size
        lens = sample._item_tokens(sample._split_key()) if sample.keys else [1] * sample.bs
        if sample.bs < dp:
            raise RuntimeError(f"MFC {self.rpc.name}: {sample.bs} samples cannot feed {dp} data-parallel replicas")
        if self.rpc.interface_type == ModelInterfaceType.TRAIN_STEP:
            parts = datapack.balanced_partition(lens, dp, equal_size=True)
        else:
            parts = datapack.balanced_partition(lens, dp, equal_size=False)
        return parts, [[sample.ids[i] for i in p] for p in parts]

    def attach_payloads_with_hooks(self, payloads: Dict[int, Payload], meta: SequenceSample,
                                   dests: Dict[int, List[str]]) -> Dict[int, Payload]:
        """Add the data-transfer pre-hook (and the rpc's own hooks) to the model's payloads; owners of
        data that are not workers of this model get a stand-alone transfer request."""
        plan = self.planner.derive_plan(dests, list(self.rpc.input_keys)) if self.rpc.input_keys else []
        involved = {w for s in plan for w in [s.src] + s.dsts}
        hook = dict(meta=meta, plan=plan)
        for w, p in payloads.items():
            p.pre_hooks.append("data_transfer")
            p.pre_hook_data.append(hook)
        for w in sorted(involved - set(payloads)):
            payloads[w] = Payload(handler=w, handle_name="empty", pre_hooks=["data_transfer"], pre_hook_data=[hook])
        for h in self.rpc._post_hooks:
            if isinstance(h, ParamReallocHook) and h.target is not None:
                # a worker-to-worker transfer: every worker of the source AND of the target executes it
                for w in sorted(set(self.realloc_peers[h.target]) - set(payloads)):
                    payloads[w] = Payload(handler=w, handle_name="empty")
                for w in sorted(set(self.workers) | set(self.realloc_peers[h.target])):
                    payloads[w].post_hooks.append("param_realloc")
                    payloads[w].post_hook_data.append(dict(source=h.source, target=h.target, eta=h.eta))
                continue
            for w in self.workers:
                if isinstance(h, ParamReallocHook):
                    payloads[w].post_hooks.append("param_realloc")
                    payloads[w].post_hook_data.append(dict(source=h.source, target=h.target, eta=h.eta))
                elif isinstance(h, OffloadHook):
                    payloads[w].post_hooks.append("offload")
                    payloads[w].post_hook_data.append(dict(model_name=self.rpc.model_name))
        self._plan = plan
        return payloads

    # ------------------------------------------------------------------------------- step
    async def run_step(self, ids: List[str], sample: SequenceSample) -> Any:
        rpc = self.rpc
        tik = time.perf_counter()
        parts, id_parts = self.data_parallel_dispatch(sample)
        dests: Dict[int, List[str]] = {}
        payloads: Dict[int, Payload] = {}
        for d, (ws, part_ids) in enumerate(zip(self.workers_by_dp, id_parts)):
            for w in ws:
                dests[w] = part_ids
                payloads[w] = Payload(handler=w, handle_name=rpc.interface_type.value,
                                      data=dict(rpc_name=rpc.name, model_name=rpc.model_name, ids=part_ids,
                                                dp_rank=d, is_dp_head=(w == self.dp_heads[d])))
        async with self.ctrl.post_lock:
            payloads = self.attach_payloads_with_hooks(payloads, sample, dests)
            req_ids = {w: self.stream.post(p) for w, p in sorted(payloads.items())}
        loop = asyncio.get_running_loop()
        replies = await asyncio.gather(*[
            loop.run_in_executor(None, lambda r=r: self.stream.poll(r, block=True, timeout=None))
            for r in req_ids.values()])
        by_worker = dict(zip(req_ids.keys(), replies))
        for w, rep in by_worker.items():
            if isinstance(rep.data, dict) and rep.data.get("__error__"):
                raise RuntimeError(f"model worker {w} failed in {rpc.name}: {rep.data['__error__']}")
        self.planner.apply(self._plan)
        heads = [by_worker[h].data for h in self.dp_heads]
        result: Any
        if rpc.interface_type == ModelInterfaceType.TRAIN_STEP:
            result = _merge_stats([h for h in heads if h is not None])
        elif rpc.interface_type in (ModelInterfaceType.INFERENCE, ModelInterfaceType.GENERATE):
            outs = [h for h in heads if isinstance(h, SequenceSample)]
            result = SequenceSample.gather(outs) if outs else None
            if result is not None:
                for h, o in zip(self.dp_heads, heads):
                    if isinstance(o, SequenceSample):
                        self.tracker.add_data(h, o.ids, list(o.keys))
                order = {i: k for k, i in enumerate(result.ids)}
                metas = result.unpack()
                await self.buffer.amend_batch(ids, [metas[order[i]] for i in ids])
        else:
            result = heads
        self.ctrl.step_stats[rpc.name] = dict(time=time.perf_counter() - tik, n_seqs=sample.bs, result=result)
        if rpc.log_return_value:
            logger.info(f"MFC {rpc.name} returns {result}")
        return result

    async def run(self):
        """One execution per training step: wait for a ready batch, then run it."""
        ids, sample = await self.buffer.get_batch_for_rpc(self.rpc)
        self.batch_taken.set()
        if self.rpc.is_dst:
            self.ctrl.ids_to_clear.update(ids)
        return await self.run_step(ids, sample)


def _merge_stats(stats: List[Any]) -> Any:
    """Average the scalar statistics the dp heads report (lists of per-minibatch dicts are merged
    position-wise)."""
    if not stats:
        return {}
    if isinstance(stats[0], list):
        n = min(len(s) for s in stats)
        return [_merge_stats([s[i] for s in stats]) for i in range(n)]
    keys = set().union(*[set(s) for s in stats if isinstance(s, dict)])
    out = {}
    for k in keys:
        vals = [s[k] for s in stats if isinstance(s, dict) and k in s and isinstance(s[k], (int, float))]
        if vals:
            out[k] = sum(vals) / len(vals)
    return out
