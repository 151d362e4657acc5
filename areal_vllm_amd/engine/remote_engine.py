"""HTTP rollout client: the ``InferenceEngine`` API against one or many native rollout servers.

Behaviour of the reference's ``RemoteSGLangEngine`` (``areal/engine/sglang_remote.py:83-480``): server
discovery through ``AREAL_LLM_SERVER_ADDRS`` or the name-resolve ``gen_servers`` subtree, round-robin
scheduling with request-id affinity, interruptible ``agenerate`` (abort -> append partial output ->
re-issue), and the pause -> update (disk or collective) -> continue weight-update protocol driven from
trainer rank 0.
"""
import asyncio
import os
import shutil
import time
from concurrent.futures import Future, ProcessPoolExecutor, ThreadPoolExecutor
from typing import Any, Callable, Dict, List, Optional

import aiohttp
import requests

from ..api.cli_args import InferenceEngineConfig
from ..api.engine_api import InferenceEngine
from ..api.io_struct import ModelRequest, ModelResponse, WeightUpdateMeta
from ..api.workflow_api import WorkflowExecutor
from ..utils import logging, name_resolve, names
from ..utils.http import arequest_with_retry, get_default_connector
from .local_engine import interruptible_generate

logger = logging.getLogger("remote-engine")
RID_CACHE_SIZE = 128


class RemoteInferenceEngine(InferenceEngine):
    def __init__(self, config: InferenceEngineConfig):
        self.config = config
        self.rid_to_address: Dict[str, str] = {}
        self.rid_queue: List[str] = []
        self.addresses: List[str] = []
        self.server_idx = 0
        self.distributed_weight_update_initialized = False
        self._version = 0
        self.workflow_executor = WorkflowExecutor(config, self)
        self.executor = None

    # ------------------------------------------------------------------------------ set-up
    def _discover(self) -> List[str]:
        env = os.getenv("AREAL_LLM_SERVER_ADDRS")
        if env:
            return [a for a in env.split(",") if a]
        if self.config.experiment_name and self.config.trial_name:
            key = names.gen_servers(self.config.experiment_name, self.config.trial_name)
            deadline = time.time() + self.config.setup_timeout
            while time.time() < deadline:
                addrs = name_resolve.get_subtree(key)
                if addrs:
                    return sorted(addrs)
                time.sleep(0.5)
        return []

    def _wait_for_server(self, address: str):
        base = f"http://{address}"
        tik = time.time()
        while time.time() - tik < self.config.setup_timeout:
            try:
                if requests.get(f"{base}/health", timeout=30).status_code == 200:
                    return
            except requests.exceptions.RequestException:
                pass
            time.sleep(0.5)
        raise RuntimeError(f"server {address} did not become healthy within {self.config.setup_timeout}s")

    def initialize(self, addr: Optional[str] = None, ft_spec=None, train_data_parallel_size: Optional[int] = None,
                   **kwargs):
        self.addresses = [addr] if addr else self._discover()
        if not self.addresses:
            raise RuntimeError("No configured rollout servers (set AREAL_LLM_SERVER_ADDRS or register them in "
                               "name_resolve).")
        logger.info("Waiting for server ready...")
        for a in self.addresses:
            self._wait_for_server(a)
        logger.info("Servers are all ready!")
        self.executor = ThreadPoolExecutor(max_workers=1)
        self.workflow_executor.initialize(logger=logger, train_data_parallel_size=train_data_parallel_size)
        return self

    def destroy(self):
        self.workflow_executor.destroy()
        if self.executor is not None:
            self.executor.shutdown(wait=False)

    def set_version(self, version: int):
        self._version = version
        self.workflow_executor.notify_version_changed()

    def get_version(self) -> int:
        return self._version

    # -------------------------------------------------------------------------- scheduling
    def choose_server(self) -> str:
        if self.config.schedule_policy == "round_robin":
            s = self.addresses[self.server_idx]
            self.server_idx = (self.server_idx + 1) % len(self.addresses)
            return s
        raise NotImplementedError("Only round-robin scheduling is implemented.")

    def _server_for(self, rid: str) -> str:
        if rid in self.rid_to_address:
            return self.rid_to_address[rid]
        addr = self.choose_server()
        if len(self.rid_queue) >= RID_CACHE_SIZE:
            self.rid_to_address.pop(self.rid_queue.pop(0), None)
        self.rid_to_address[rid] = addr
        self.rid_queue.append(rid)
        return addr

    async def agenerate(self, req: ModelRequest) -> ModelResponse:
        g = req.gconfig
        if g.stop and req.tokenizer is None:
            raise ValueError("stop strings need a tokenizer on the request")
        addr = self._server_for(req.rid)

        # stop strings travel as token-id sequences: the server matches them against the generated tokens
        stop_seqs = [list(req.tokenizer.encode(s_, add_special_tokens=False)) for s_ in (g.stop or [])]

        async def gen_once(rid, ids, max_new, min_new):
            payload = {
                "rid": rid, "input_ids": ids, "return_logprob": True, "stream": False,
                "sampling_params": {"top_p": g.top_p, "top_k": g.top_k, "max_new_tokens": max_new,
                                    "min_new_tokens": min_new, "temperature": 0.0 if g.greedy else g.temperature,
                                    "stop_token_ids": g.stop_token_ids, "stop_sequences": stop_seqs,
                                    "frequency_penalty": g.frequency_penalty},
            }
            if req.image_data:  # base64 images; the server runs its own image processor on them
                payload["image_data"] = list(req.image_data)
            result = await arequest_with_retry(addr=addr, endpoint="/generate", payload=payload, method="POST",
                                               max_retries=self.config.request_retries,
                                               timeout=self.config.request_timeout)
            meta = result["meta_info"]
            lps = meta["output_token_logprobs"]
            toks = [x[1] for x in lps]
            vers = meta.get("output_versions") or [self._version] * len(toks)
            return dict(output_tokens=toks, output_logprobs=[x[0] for x in lps], output_versions=vers,
                        finish_reason=meta["finish_reason"]["type"])

        return await interruptible_generate(req, gen_once, self.workflow_executor.paused.is_set, abort_sleep=0.5)

    # ----------------------------------------------------------------------- weight update
    def update_weights(self, meta: WeightUpdateMeta) -> Future:
        next_version = self._version + 1
        if meta.type == "ipc":
            # the servers keep decoding while they pull the new weights into a staging buffer and only stop
            # for the flip between two engine steps: no pause / continue round trip
            return self.executor.submit(_update_from_ipc, meta, self.config.experiment_name, self.config.trial_name,
                                        self._version, self.addresses, self.config.request_retries,
                                        self.config.request_timeout, next_version)
        for a in self.addresses:
            requests.post(f"http://{a}/pause_generation", timeout=self.config.request_timeout).raise_for_status()
        if meta.type == "disk":
            fut = self.executor.submit(_update_from_disk, self.config.experiment_name, self.config.trial_name,
                                       self._version, self.addresses, meta.path, self.config.request_retries,
                                       self.config.request_timeout, next_version)
        elif meta.type in ("nccl", "xgmi"):
            fut = self.executor.submit(_update_from_distributed, meta, self.addresses, self.config.request_retries,
                                       self.config.request_timeout, not self.distributed_weight_update_initialized,
                                       next_version)
            self.distributed_weight_update_initialized = True
        else:
            raise NotImplementedError(f"Unsupported weight update type: {meta.type}")

        def done(_f):
            for a in self.addresses:
                try:
                    requests.post(f"http://{a}/continue_generation", timeout=self.config.request_timeout)
                except requests.exceptions.RequestException as e:
                    logger.warning(f"continue_generation failed on {a}: {e}")
            if meta.type == "disk" and meta.path and os.path.isdir(meta.path):
                shutil.rmtree(meta.path, ignore_errors=True)

        fut.add_done_callback(done)
        return fut

    def submit(self, data, workflow=None, workflow_builder=None) -> None:
        return self.workflow_executor.submit(data, workflow, workflow_builder)

    def wait(self, count, timeout=None, should_accept=None):
        return self.workflow_executor.wait(count, timeout=timeout, should_accept=should_accept)

    def rollout_batch(self, data, workflow=None, workflow_builder=None, should_accept=None):
        return self.workflow_executor.rollout_batch(data, workflow, workflow_builder, should_accept)

    def prepare_batch(self, dataloader, workflow=None, workflow_builder=None, should_accept=None):
        return self.workflow_executor.prepare_batch(dataloader, workflow, workflow_builder, should_accept)

    def pause(self):
        return self.workflow_executor.pause()

    def resume(self):
        return self.workflow_executor.resume()


def _post_all(addresses, endpoint, payload, retries, timeout):
    async def run():
        async with aiohttp.ClientSession(timeout=aiohttp.ClientTimeout(total=timeout),
                                         connector=get_default_connector()) as session:
            return await asyncio.gather(*[
                arequest_with_retry(addr=a, session=session, endpoint=endpoint, payload=payload, method="POST",
                                    max_retries=retries, timeout=timeout) for a in addresses])

    return asyncio.run(run())


def _update_from_disk(experiment_name, trial_name, model_version, addresses, path, retries, timeout, next_version):
    key = names.update_weights_from_disk(experiment_name, trial_name, model_version)
    save_ts = int(name_resolve.wait(key, timeout=120))
    load_ts = time.time_ns()
    logger.info(f"Begin update weights from {path}, responded in {(load_ts - save_ts) / 1e6:.2f} ms")
    res = _post_all(addresses, "/update_weights_from_disk",
                    dict(model_path=str(path), abort_all_requests=True, version=next_version), retries, timeout)
    for r in res:
        assert r.get("success"), r
    logger.info(f"Loading weights done in {(time.time_ns() - load_ts) / 1e6:.2f} ms")
    return True


def _update_from_ipc(meta: WeightUpdateMeta, experiment_name, trial_name, model_version, addresses, retries,
                     timeout, next_version):
    """Wait until the trainer says its buffers are final, then let every server pull: server i reads from the
    trainer's data-parallel replica i % dp (spreading the sources spreads the links); with a sharded trainer or a
    tensor-parallel server every server rank works out the ranges it needs from the trainer's layout."""
    key = names.update_weights_from_ipc(experiment_name, trial_name, model_version)
    name_resolve.wait(key, timeout=600)
    t0 = time.perf_counter()
    src = dict(meta.ipc_src or {})
    plain = not src or (src.get("tp", 1) == 1 and src.get("pp", 1) == 1 and src.get("zero3", 1) == 1
                        and src.get("ep", 1) == 1)
    n = len(meta.ipc_buffers)

    async def go():
        async with aiohttp.ClientSession(timeout=aiohttp.ClientTimeout(total=timeout),
                                         connector=get_default_connector()) as session:
            return await asyncio.gather(*[
                arequest_with_retry(addr=a, session=session, endpoint="/update_weights_from_ipc", method="POST",
                                    payload=dict(buffer=meta.ipc_buffers[i % n] if plain else None,
                                                 buffers=meta.ipc_buffers, src=src, replica=i,
                                                 layout=meta.ipc_layout, used_bytes=meta.ipc_used_bytes,
                                                 version=next_version, abort_all_requests=True),
                                    max_retries=retries, timeout=timeout)
                for i, a in enumerate(addresses)])

    for r in asyncio.run(go()):
        assert r.get("success"), r
    logger.info(f"peer-memory weight update of {len(addresses)} server(s) took {(time.perf_counter() - t0) * 1e3:.1f} ms")
    return True


def _update_from_distributed(meta: WeightUpdateMeta, addresses, retries, timeout, init_group, next_version):
    if init_group:
        from ..platforms import current_platform

        # RCCL between GPUs; on the CPU platform the same broadcast protocol runs over gloo (the trainer side
        # makes the same choice in NativeTrainEngine._init_weight_update_group)
        backend = "gloo" if (meta.type == "gloo" or current_platform.device_type == "cpu") else "nccl"
        tp = meta.alloc_mode.gen.tp_size if meta.alloc_mode is not None else 1
        world = (meta.alloc_mode.gen.world_size if meta.alloc_mode is not None else len(addresses)) + 1

        async def init_all():
            async with aiohttp.ClientSession(timeout=aiohttp.ClientTimeout(total=timeout),
                                             connector=get_default_connector()) as session:
                return await asyncio.gather(*[
                    arequest_with_retry(addr=a, session=session, endpoint="/init_weights_update_group", method="POST",
                                        payload=dict(master_address=meta.nccl_master_address,
                                                     master_port=str(meta.sender_port(s)), rank_offset=1 + i * tp,
                                                     world_size=world, backend=backend,
                                                     group_name=meta.sender_group_name(s)),
                                        max_retries=retries, timeout=timeout)
                    for s in range(max(1, meta.nccl_n_senders)) for i, a in enumerate(addresses)])

        for r in asyncio.run(init_all()):
            assert r.get("success"), r
    n = max(1, meta.nccl_n_senders)
    for lo in range(0, len(meta.nccl_param_specs), n):
        # one round: chunk lo + s arrives from sender s, all of them concurrently
        groups = [dict(group_name=meta.sender_group_name(s), names=[p.name for p in chunk],
                       dtypes=[p.dtype for p in chunk], shapes=[list(p.shape) for p in chunk])
                  for s, chunk in enumerate(meta.nccl_param_specs[lo:lo + n])]
        payload = dict(groups[0], abort_all_requests=True, version=next_version)
        if n > 1:
            payload["groups"] = groups
        for r in _post_all(addresses, "/update_weights_from_distributed", payload, retries, timeout):
            assert r.get("success"), r
    return True


RemoteSGLangEngine = RemoteInferenceEngine  # name used by the reference's scripts
