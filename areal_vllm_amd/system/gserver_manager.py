"""Generation-server manager: request routing, staleness gate and weight roll-out for a fleet of
rollout servers.

HTTP endpoints (consumed by rollout workers):
  POST /schedule_request  GenReqMeta -> {url, version}: sticky to the previous server while the weight
       version is unchanged (its KV prefix is warm), otherwise round_robin / least_requests /
       least_token_usage
  POST /get_model_version
  POST /allocate_rollout  {qid} -> {success, reason}: refuses when `max_concurrent_rollouts` are running or
       when starting another rollout would exceed the staleness bound
       ``(trained_samples + running) // train_batch_size > max_head_offpolicyness + version``
  POST /finish_rollout    GenRespMeta
The poll loop watches the trainer's published model version; on a new version every server is paused,
told to load the new weights from disk, and resumed; old weight directories are garbage-collected.
Behaviour of ``realhf/system/gserver_manager.py:36-520`` (FastAPI/uvicorn there, aiohttp here).
"""
import asyncio
import dataclasses
import os
import shutil
import threading
import time
from collections import defaultdict
from typing import Dict, List, Optional

import aiohttp
from aiohttp import web

from ..api.io_struct import RolloutStat
from ..utils import logging, name_resolve, names, network
from .worker_base import ExpStatus, PollResult, Worker

logger = logging.getLogger("gserver-manager")


@dataclasses.dataclass
class GserverManagerConfig:
    model_name: str = "actor"
    n_servers: int = 1
    schedule_policy: str = "round_robin"  # round_robin | least_requests | least_token_usage
    max_head_offpolicyness: int = 0
    train_batch_size: int = 1
    max_concurrent_rollouts: int = 128
    flush_request_timeout: float = 300.0
    realloc_root: str = "/tmp/areal/param_realloc"
    poll_interval: float = 1.0
    wait_experiment_status: bool = True


class GserverManager(Worker):
    def _configure(self, config: GserverManagerConfig):
        self.config = config
        self.schedule_policy = config.schedule_policy
        self.rollout_stat = RolloutStat()
        self.lock = threading.Lock()
        self.server_urls: List[str] = []
        self._qid_to_server: Dict[str, str] = {}
        self._server_request_counts: Dict[str, int] = defaultdict(int)
        self._server_token_usage: Dict[str, float] = defaultdict(float)
        self._version = 0
        self._rr = 0
        self._gen_tokens = 0
        self._last_thpt_time = time.time()
        self._staleness_warned = set()
        self._base_samples = None
        self.thread: Optional[threading.Thread] = None
        self.manager_addr: Optional[str] = None
        self._loop: Optional[asyncio.AbstractEventLoop] = None
        self._runner = None
        return self._info

    # --------------------------------------------------------------------------- policies
    def _round_robin_schedule(self, meta) -> int:
        r = self._rr
        self._rr = (self._rr + 1) % len(self.server_urls)
        return r

    def _least_requests_schedule(self, meta) -> int:
        return min(range(len(self.server_urls)), key=lambda i: self._server_request_counts[self.server_urls[i]])

    def _least_token_usage_schedule(self, meta) -> int:
        return min(range(len(self.server_urls)), key=lambda i: self._server_token_usage[self.server_urls[i]])

    def schedule(self, meta: Dict) -> Dict:
        with self.lock:
            prev = meta.get("previous_server_url")
            if prev and meta.get("previous_version") == self._version:
                return dict(url=prev, version=self._version)
            pick = {"round_robin": self._round_robin_schedule, "least_requests": self._least_requests_schedule,
                    "least_token_usage": self._least_token_usage_schedule}.get(self.schedule_policy)
            if pick is None:
                raise NotImplementedError(f"Unknown schedule policy {self.schedule_policy}")
            url = self.server_urls[pick(meta)]
            qid = str(meta["qid"])
            if qid not in self._qid_to_server:
                self._server_request_counts[url] += 1
            self._qid_to_server[qid] = url
            self._server_token_usage[url] += meta["prompt_len"] + meta["new_token_budget"] * meta["group_size"] * 0.4
            return dict(url=url, version=self._version)

    # --------------------------------------------------------------------------- staleness
    def get_training_sample_cnt(self) -> int:
        try:
            return int(name_resolve.get(names.training_samples(self.experiment_name, self.trial_name)))
        except name_resolve.NameEntryNotFoundError:
            return 0

    def is_staled(self) -> bool:
        # finished rollouts count as soon as they are accepted, not only once the trainer has fetched
        # them: samples waiting in the trainer's stream queue are just as committed as fetched ones
        if self._base_samples is None:
            self._base_samples = self.get_training_sample_cnt()
        n = max(self.get_training_sample_cnt(), self._base_samples + self.rollout_stat.accepted) \
            + self.rollout_stat.running
        expected_version = n // self.config.train_batch_size
        staled = expected_version > self.config.max_head_offpolicyness + self._version
        if staled and self._version not in self._staleness_warned:
            self._staleness_warned.add(self._version)
            logger.warning(f"expected version ({expected_version}) = global sample cnt ({n}) // batch size "
                           f"({self.config.train_batch_size}), current version {self._version}, offpolicyness "
                           f"{self.config.max_head_offpolicyness}: holding new rollouts")
        return staled

    def allocate_rollout(self, qid) -> Dict:
        with self.lock:
            has_capacity = self.rollout_stat.running < self.config.max_concurrent_rollouts
            staled = self.is_staled()
            if has_capacity and not staled:
                self.rollout_stat.submitted += 1
                self.rollout_stat.running += 1
                return dict(success=True, reason="")
            reason = []
            if not has_capacity:
                reason.append(f"capacity: {self.rollout_stat.running} >= {self.config.max_concurrent_rollouts}")
            if staled:
                reason.append("staled")
            return dict(success=False, reason=" and ".join(reason))

    def finish_rollout(self, meta: Dict) -> Dict:
        with self.lock:
            url = self._qid_to_server.pop(str(meta["qid"]), None)
            if url is not None:
                self._server_request_counts[url] = max(0, self._server_request_counts[url] - 1)
            self._gen_tokens += int(meta.get("n_tokens", 0))
            self.rollout_stat.running -= 1
            if meta.get("accepted"):
                self.rollout_stat.accepted += 1
            return dict(success=True)

    # ------------------------------------------------------------------------ HTTP service
    def _build_app(self) -> web.Application:
        app = web.Application()

        async def schedule_request(req):
            return web.json_response(self.schedule(await req.json()))

        async def get_model_version(req):
            with self.lock:
                return web.json_response(dict(version=self._version))

        async def allocate(req):
            return web.json_response(self.allocate_rollout((await req.json())["qid"]))

        async def finish(req):
            return web.json_response(self.finish_rollout(await req.json()))

        app.router.add_post("/schedule_request", schedule_request)
        app.router.add_post("/get_model_version", get_model_version)
        app.router.add_post("/allocate_rollout", allocate)
        app.router.add_post("/finish_rollout", finish)
        return app

    def _run_routing_service(self, started: threading.Event):
        self._loop = asyncio.new_event_loop()
        asyncio.set_event_loop(self._loop)
        port = network.find_free_ports(1)[0]
        host = network.gethostip()
        self._runner = web.AppRunner(self._build_app())
        self._loop.run_until_complete(self._runner.setup())
        self._loop.run_until_complete(web.TCPSite(self._runner, "0.0.0.0", port).start())
        self.manager_addr = f"{host}:{port}"
        started.set()
        self._loop.run_forever()

    def _discover_servers(self, n: int, timeout: float = 300.0) -> List[str]:
        key = names.gen_servers(self.experiment_name, self.trial_name)
        deadline = time.time() + timeout
        while True:
            urls = name_resolve.get_subtree(key)
            if len(urls) >= n:
                assert len(set(urls)) == len(urls), urls
                return sorted(u if u.startswith("http") else f"http://{u}" for u in urls)
            if time.time() > deadline:
                raise TimeoutError(f"only {len(urls)}/{n} generation servers registered")
            time.sleep(0.5)

    # ---------------------------------------------------------------------- weight updates
    def check_new_params(self) -> Optional[str]:
        try:
            v = int(name_resolve.get(names.model_version(self.experiment_name, self.trial_name,
                                                         self.config.model_name)))
        except name_resolve.NameEntryNotFoundError:
            return None
        if v > self._version:
            self._pending_version = v
            return os.path.join(self.config.realloc_root, self.config.model_name, str(v))
        return None

    async def flush_requests_and_update_weights(self, url: str, path: str, version: int):
        to = aiohttp.ClientTimeout(total=self.config.flush_request_timeout)
        async with aiohttp.ClientSession(url, timeout=to) as s:
            (await s.post("/pause_generation")).raise_for_status()
            async with s.post("/update_weights_from_disk", json=dict(model_path=path, version=version)) as r:
                r.raise_for_status()
                body = await r.json()
                assert body["success"], body
            (await s.post("/continue_generation")).raise_for_status()

    async def _get_metric(self, url: str, key: str) -> float:
        to = aiohttp.ClientTimeout(total=self.config.flush_request_timeout)
        async with aiohttp.ClientSession(url, timeout=to) as s:
            async with s.get("/metrics") as r:
                r.raise_for_status()
                for line in (await r.text()).split("\n"):
                    if line.startswith(key):
                        return float(line.split(" ")[-1])
        raise RuntimeError(f"metric {key} not exported by {url}")

    def _poll(self) -> PollResult:
        if self.thread is None:
            self.server_urls = self._discover_servers(self.config.n_servers)
            started = threading.Event()
            self.thread = threading.Thread(target=self._run_routing_service, args=(started,), daemon=True)
            self.thread.start()
            started.wait(30)
            name_resolve.add(names.gen_server_manager(self.experiment_name, self.trial_name), self.manager_addr,
                             replace=True)
            logger.info(f"GserverManager HTTP service started at {self.manager_addr}")
        if self.config.wait_experiment_status:
            try:
                st = name_resolve.wait(names.experiment_status(self.experiment_name, self.trial_name), timeout=300)
            except TimeoutError:
                raise TimeoutError("experiment status never appeared: the master worker is not running")
            if st != ExpStatus.RUNNING.value:
                self.exit()
                return PollResult(0, 0)
        path = self.check_new_params()
        if path is not None:
            v = self._pending_version

            async def go():
                await asyncio.gather(*[self.flush_requests_and_update_weights(u, path, v) for u in self.server_urls])

            asyncio.run(go())
            with self.lock:
                self._version = v
            logger.info(f"Generation servers updated weights from {path} (version {v})")
        if self.schedule_policy == "least_token_usage":
            async def usage():
                return await asyncio.gather(*[self._get_metric(u, "sglang:num_used_tokens") for u in self.server_urls])

            vals = asyncio.run(usage())
            with self.lock:
                for u, x in zip(self.server_urls, vals):
                    self._server_token_usage[u] = x
        if time.time() - self._last_thpt_time > 30:
            dt = time.time() - self._last_thpt_time
            logger.info(f"Generation throughput: {self._gen_tokens / dt:.2f} tokens/s")
            self._last_thpt_time, self._gen_tokens = time.time(), 0
        root = os.path.join(self.config.realloc_root, self.config.model_name)
        if os.path.isdir(root):  # keep the current and the previous version (recover needs one)
            for d in os.listdir(root):
                if d.isdigit() and int(d) < self._version - 1:
                    shutil.rmtree(os.path.join(root, d), ignore_errors=True)
        time.sleep(self.config.poll_interval)
        return PollResult(0, 0)

    def _exit_hook(self, exit_status):
        if self._loop is not None:
            self._loop.call_soon_threadsafe(self._loop.stop)
        if self.thread is not None:
            self.thread.join(timeout=3)
