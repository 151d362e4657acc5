"""In-process rollout client: the ``InferenceEngine`` API on top of a ``NativeInferenceEngine`` that lives
in the same process as the trainer (colocated allocation, one GPU shared by rollout and training).

Counterpart of the reference's experimental local engine (``areal/experimental/sglang_engine.py``) and of
the interruptible-generation client loop of ``areal/engine/sglang_remote.py:121-249``: a request whose
generation was aborted by a weight update keeps the tokens it already produced, is re-issued with
prompt + partial output, and the final response carries per-token weight versions.
"""
import asyncio
import time
import uuid
from concurrent.futures import Future, ThreadPoolExecutor
from typing import Any, Callable, Dict, List, Optional

from ..api.cli_args import InferenceEngineConfig
from ..api.engine_api import InferenceEngine
from ..api.io_struct import ModelRequest, ModelResponse, WeightUpdateMeta
from ..api.workflow_api import RolloutWorkflow, WorkflowExecutor
from ..inference.engine import NativeInferenceEngine
from ..utils import logging, stop_strings

logger = logging.getLogger("local-engine")


async def interruptible_generate(req: ModelRequest, gen_once: Callable, is_paused: Callable[[], bool],
                                 abort_sleep: float = 0.01) -> ModelResponse:
    """Shared client loop: call `gen_once(rid, input_ids, max_new_tokens)` until the request stops for a
    reason other than an abort; accumulate tokens / logprobs / versions across re-issues."""
    g = req.gconfig
    if g.n_samples != 1:
        raise ValueError("agenerate handles one sample per request; issue n_samples requests instead")
    start = time.perf_counter()
    acc_tokens: List[int] = []
    acc_logprobs: List[float] = []
    acc_versions: List[int] = []
    stop_reason = "length"
    ttft = None
    itl: List[float] = []
    prompt = list(req.input_ids)
    while len(acc_tokens) < g.max_new_tokens:
        while is_paused():
            await asyncio.sleep(abort_sleep)
        res = await gen_once(f"{req.rid}#{len(acc_tokens)}-{uuid.uuid4().hex[:6]}", prompt + acc_tokens,
                             g.max_new_tokens - len(acc_tokens), max(0, g.min_new_tokens - len(acc_tokens)))
        if ttft is None and res["output_tokens"]:
            ttft = time.perf_counter() - start
        acc_tokens += res["output_tokens"]
        acc_logprobs += res["output_logprobs"]
        acc_versions += res["output_versions"]
        itl += res.get("itl", [])
        if g.stop and req.tokenizer is not None:
            # the servers match stop strings as token-id sequences; the text decides (a stop string spelled with
            # other tokens): cut after the token that completes it
            first_new = len(acc_tokens) - len(res["output_tokens"])
            cut = stop_strings.first_cut(req.tokenizer, acc_tokens, g.stop, first_new)
            if cut is not None:
                del acc_tokens[cut:], acc_logprobs[cut:], acc_versions[cut:]
                stop_reason = "stop"
                break
        fr = res["finish_reason"]
        if fr == "abort":
            stop_reason = "interrupt"
            await asyncio.sleep(abort_sleep)
            continue
        stop_reason = fr
        break
    if stop_reason == "interrupt":  # budget exhausted while being interrupted
        stop_reason = "length"
    return ModelResponse(input_tokens=prompt, output_tokens=acc_tokens, output_logprobs=acc_logprobs,
                         output_versions=acc_versions, stop_reason=stop_reason, tokenizer=req.tokenizer,
                         latency=time.perf_counter() - start, ttft=ttft if ttft is not None else float("inf"),
                         itl=itl)


class LocalInferenceEngine(InferenceEngine):
    def __init__(self, config: InferenceEngineConfig, engine: NativeInferenceEngine):
        self.config = config
        self.engine = engine
        self.workflow_executor = WorkflowExecutor(config, self)
        self._pool = ThreadPoolExecutor(max_workers=1)

    def initialize(self, addr: Optional[str] = None, ft_spec=None, train_data_parallel_size: Optional[int] = None,
                   **kwargs):
        self.workflow_executor.initialize(logger=logger, train_data_parallel_size=train_data_parallel_size)
        return self

    def destroy(self):
        self.workflow_executor.destroy()
        self._pool.shutdown(wait=False)

    def get_version(self) -> int:
        return self.engine.get_version()

    def set_version(self, version: int) -> None:
        self.engine.set_version(version)
        self.workflow_executor.notify_version_changed()

    async def agenerate(self, req: ModelRequest) -> ModelResponse:
        g = req.gconfig
        mm = (req.metadata or {}).get("multi_modal_input")
        if g.stop and req.tokenizer is None:
            raise ValueError("stop strings need a tokenizer on the request")
        stop_seqs = [list(req.tokenizer.encode(s_, add_special_tokens=False)) for s_ in (g.stop or [])]

        async def gen_once(rid, ids, max_new, min_new):
            r = await self.engine.agenerate_once(
                rid, ids, max_new_tokens=max_new, min_new_tokens=min_new, temperature=g.temperature, top_p=g.top_p,
                top_k=g.top_k, stop_token_ids=g.stop_token_ids, greedy=g.greedy, multimodal=mm,
                stop_sequences=stop_seqs, frequency_penalty=g.frequency_penalty)
            return dict(output_tokens=r.output_tokens, output_logprobs=r.output_logprobs,
                        output_versions=r.output_versions, finish_reason=r.finish_reason, itl=r.itl)

        return await interruptible_generate(req, gen_once, self.workflow_executor.paused.is_set)

    def update_weights(self, meta: WeightUpdateMeta) -> Future:
        """For `inproc` updates the trainer pushes the weights itself (``upload_weights``); this only returns
        a completed future so that training scripts written for remote engines keep their shape. `disk`
        updates are loaded here."""
        if meta.type == "disk":
            return self._pool.submit(self.engine.load_weights_from_disk, meta.path, None)
        f: Future = Future()
        f.set_result(None)
        return f

    def submit(self, data, workflow=None, workflow_builder=None) -> None:
        return self.workflow_executor.submit(data, workflow, workflow_builder)

    def wait(self, count, timeout=None, should_accept=None):
        return self.workflow_executor.wait(count, timeout=timeout, should_accept=should_accept)

    def rollout_batch(self, data, workflow=None, workflow_builder=None, should_accept=None):
        return self.workflow_executor.rollout_batch(data, workflow, workflow_builder, should_accept)

    def prepare_batch(self, dataloader, workflow=None, workflow_builder=None, should_accept=None):
        return self.workflow_executor.prepare_batch(dataloader, workflow, workflow_builder, should_accept)

    def pause(self):
        self.workflow_executor.pause()

    def resume(self):
        self.workflow_executor.resume()
