"""Chunked ("partial") rollout manager of a rollout worker.

Every generation is issued in chunks of at most `new_tokens_per_chunk` tokens. After each chunk the
request goes back through the manager's ``/schedule_request``: if the weights have not changed it
returns to the same server (warm KV prefix), otherwise it may move, and the accumulated tokens are
re-prefilled with the new weights. Log-probs of all chunks are concatenated and the first / last weight
versions are recorded; the n answers of a prompt are delivered together once all of them finished.
Behaviour of ``realhf/system/partial_rollout.py:30-290``.
"""
import asyncio
import time
from collections import defaultdict
from typing import Dict, Hashable, List, Optional

import aiohttp

from ..api.cli_args import GenerationHyperparameters
from ..utils import logging, name_resolve, names
from .model_api import APIGenerateInput, APIGenerateOutput, BundledGenerationOutputs, GenReqMeta

logger = logging.getLogger("partial-rollout")
GENERATION_POLL_WAIT_TIME = 0.05


async def generate_once(session: aiohttp.ClientSession, url: str, inp: APIGenerateInput) -> APIGenerateOutput:
    """One chunk against a native rollout server."""
    g = inp.gconfig
    payload = dict(rid=f"{inp.qid}-{inp.metadata.get('group_idx', 0)}-{time.monotonic_ns()}", input_ids=inp.input_ids,
                   return_logprob=True, stream=False,
                   sampling_params=dict(max_new_tokens=g.max_new_tokens, min_new_tokens=g.min_new_tokens,
                                        temperature=0.0 if g.greedy else g.temperature, top_p=g.top_p, top_k=g.top_k,
                                        stop_token_ids=inp.stop_token_ids))
    base = url if url.startswith("http") else f"http://{url}"
    async with session.post(f"{base}/generate", json=payload) as r:
        r.raise_for_status()
        res = await r.json()
    meta = res["meta_info"]
    toks = [x[1] for x in meta["output_token_logprobs"]]
    lps = [x[0] for x in meta["output_token_logprobs"]]
    out = APIGenerateOutput.from_input(inp)
    out.output_ids = [toks]
    out.output_logprobs = [lps]
    out.no_eos = [meta["finish_reason"]["type"] != "stop"]
    out.metadata = dict(inp.metadata, finish_reason=meta["finish_reason"]["type"])
    return out


class PartialRolloutManager:
    def __init__(self, worker_index: int, request_queue: asyncio.Queue, reply_queue: asyncio.Queue,
                 new_tokens_per_chunk: int, stop_token_ids: List[int], timeout: float, experiment_name: str,
                 trial_name: str, gserver_manager_addr: Optional[str] = None):
        self.worker_index = worker_index
        self.request_queue, self.reply_queue = request_queue, reply_queue
        self.new_tokens_per_chunk = new_tokens_per_chunk
        self.stop_token_ids = list(stop_token_ids)
        self.timeout = timeout
        self.experiment_name, self.trial_name = experiment_name, trial_name
        self.gserver_manager_addr = gserver_manager_addr
        self.gen_requests: Dict[Hashable, Dict[int, asyncio.Task]] = defaultdict(dict)
        self.gen_cache: Dict[Hashable, Dict[int, APIGenerateOutput]] = defaultdict(dict)
        self._session: Optional[aiohttp.ClientSession] = None

    def _sess(self) -> aiohttp.ClientSession:
        if self._session is None or self._session.closed:
            self._session = aiohttp.ClientSession(timeout=aiohttp.ClientTimeout(total=self.timeout))
        return self._session

    async def close(self):
        if self._session is not None and not self._session.closed:
            await self._session.close()

    def get_num_gen_requests(self) -> int:
        return len(self.gen_requests)

    async def _schedule_request(self, meta: GenReqMeta) -> Dict:
        if self.gserver_manager_addr is None:
            self.gserver_manager_addr = name_resolve.wait(
                names.gen_server_manager(self.experiment_name, self.trial_name), timeout=300)
        import dataclasses

        async with self._sess().post(f"http://{self.gserver_manager_addr}/schedule_request",
                                     json=dataclasses.asdict(meta)) as r:
            r.raise_for_status()
            return await r.json()

    async def _run_gen(self, url, qid, group_idx, prompt_ids, input_ids, prev_logprobs, version_start,
                       cur_server_version, raw_gconfig: GenerationHyperparameters) -> APIGenerateOutput:
        done = len(input_ids) - len(prompt_ids)
        budget = min(raw_gconfig.max_new_tokens - done, self.new_tokens_per_chunk)
        g = raw_gconfig.new(n_samples=1, max_new_tokens=budget,
                            min_new_tokens=max(0, min(budget, raw_gconfig.min_new_tokens - done)))
        inp = APIGenerateInput(qid=qid, prompt_ids=prompt_ids, input_ids=input_ids, gconfig=g,
                               stop_token_ids=self.stop_token_ids, version_start=version_start,
                               prev_logprobs=prev_logprobs,
                               metadata=dict(group_idx=group_idx, raw_gconfig=raw_gconfig, server_url=url,
                                             version=cur_server_version))
        out = await generate_once(self._sess(), url, inp)
        out.version_end = [cur_server_version] * out.group_size
        return out

    async def _issue_generation(self, url, qid, group_idx, prompt_ids, input_ids, prev_logprobs, version_start,
                                raw_gconfig, cur_server_version):
        self.gen_requests[qid][group_idx] = asyncio.create_task(self._run_gen(
            url, qid, group_idx, prompt_ids, input_ids, prev_logprobs, version_start, cur_server_version, raw_gconfig))
        await asyncio.sleep(0)

    async def refresh_generation(self):
        tasks = [t for g in self.gen_requests.values() for t in g.values()]
        if not tasks:
            await asyncio.sleep(GENERATION_POLL_WAIT_TIME)
            return
        done, _ = await asyncio.wait(tasks, timeout=GENERATION_POLL_WAIT_TIME, return_when=asyncio.FIRST_COMPLETED)
        for task in done:
            s: APIGenerateOutput = await task
            gi, raw = s.metadata["group_idx"], s.metadata["raw_gconfig"]
            self.gen_requests[s.qid].pop(gi)
            if not self.gen_requests[s.qid]:
                self.gen_requests.pop(s.qid)
            if s.no_eos[0] and s.gen_lens[0] < raw.max_new_tokens:
                # the chunk ran out of budget: continue where it stopped
                info = await self._schedule_request(GenReqMeta(
                    qid=s.qid, prompt_len=s.prompt_len, group_size=raw.n_samples, new_token_budget=raw.max_new_tokens,
                    previous_server_url=s.metadata["server_url"], previous_version=s.metadata["version"]))
                await self._issue_generation(info["url"], s.qid, gi, s.prompt_ids, s.input_ids + s.output_ids[0],
                                             list(s.prev_logprobs) + list(s.output_logprobs[0]), s.version_start,
                                             raw, info["version"])
            else:
                self.gen_cache[s.qid][gi] = s
                if len(self.gen_cache[s.qid]) >= raw.n_samples:
                    group = self.gen_cache.pop(s.qid)
                    self.reply_queue.put_nowait(BundledGenerationOutputs.from_api_outputs(
                        [group[i] for i in sorted(group)]))

    async def poll_fresh_requests_task(self):
        for _ in range(8):
            try:
                qid, prompt_ids, gconfig = self.request_queue.get_nowait()
            except asyncio.QueueEmpty:
                break
            info = await self._schedule_request(GenReqMeta(qid=qid, prompt_len=len(prompt_ids),
                                                           group_size=gconfig.n_samples,
                                                           new_token_budget=gconfig.max_new_tokens))
            for gi in range(gconfig.n_samples):
                await self._issue_generation(info["url"], qid, gi, list(prompt_ids), list(prompt_ids), [],
                                             info["version"], gconfig, info["version"])

    async def poll_old_requests_task(self):
        for _ in range(8):
            await self.refresh_generation()

    async def run_step(self):
        await asyncio.gather(self.poll_fresh_requests_task(), self.poll_old_requests_task())
