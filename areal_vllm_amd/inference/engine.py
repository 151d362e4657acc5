"""Native rollout engine: paged-KV continuous batching with interruptible generation.

``NativeInferenceEngine`` owns the weights (its own copy of the flat parameter buffer), the paged KV
cache, the C++ scheduler (``csrc/runtime/scheduler.hpp``) and a step loop thread. It replaces the
SGLang / vLLM servers the reference launches (``areal/launcher/sglang_server.py``,
``realhf/impl/model/backend/{sglang,vllm}.py``) and offers two front ends:

* in-process (``LocalInferenceEngine`` in ``engine/local_engine.py``; the reference's experimental
  ``areal/experimental/sglang_engine.py``), used when rollout and training share a GPU;
* HTTP (``inference/server.py``) with the REST verbs the reference's client speaks.

Control verbs: ``pause`` / ``resume`` (stop scheduling between steps), ``abort_all`` (finish everything
in flight with reason "abort", release pages, flush the prefix cache), weight loading from the trainer's
flat buffer (device copy on a side stream), from an HF checkpoint directory, or from named tensors.
Each generated token carries the weight version that sampled it.
"""
import asyncio
import dataclasses
import math
import os
import queue
import threading
import time
import traceback
from typing import Any, Callable, Dict, List, Optional

import numpy as np
import torch

from ..api.cli_args import InferenceServerConfig
from ..models import hf_io
from ..models.config import ModelConfig
from ..models.transformer import Transformer
from ..ops.attention import PAGE_SIZE
from ..utils import logging
from .model_runner import ModelRunner

logger = logging.getLogger("inference-engine")

try:
    from areal_vllm_amd._C import _runtime
except ImportError as e:  # pragma: no cover
    _runtime = None
    _runtime_err = e


@dataclasses.dataclass
class GenerationResult:
    rid: str
    output_tokens: List[int]
    output_logprobs: List[float]
    output_versions: List[int]
    finish_reason: str  # stop | length | abort
    prompt_len: int = 0
    latency: float = 0.0
    ttft: float = 0.0
    itl: List[float] = dataclasses.field(default_factory=list)


def _kv_bytes_per_page(cfg: ModelConfig, n_layers: int, hkv: int, elt: int = 2) -> int:
    return n_layers * 2 * hkv * PAGE_SIZE * cfg.head_dim * elt


class NativeInferenceEngine:
    def __init__(self, config: InferenceServerConfig, model_config: Optional[ModelConfig] = None,
                 device: Optional[torch.device] = None, tp_size: int = 1, tp_rank: int = 0, tp_group=None,
                 num_pages: Optional[int] = None, kv_cache_gb: Optional[float] = None):
        if _runtime is None:
            raise RuntimeError(f"C++ runtime extension missing ({_runtime_err}); run `python build_ext.py`")
        self.config = config
        self.device = device or (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available()
                                 else torch.device("cpu"))
        self.dtype = getattr(torch, config.dtype)
        from ..engine.train_engine import _resolve_model_config

        self.model_config = model_config or _resolve_model_config(config.model_path, False)
        self.tp_size, self.tp_rank, self.tp_group = tp_size, tp_rank, tp_group
        self._num_pages_arg = num_pages
        self._kv_cache_gb = kv_cache_gb
        self.model: Optional[Transformer] = None
        self.runner: Optional[ModelRunner] = None
        self.sched = None
        self._req_q: "queue.Queue" = queue.Queue()
        self._ctrl_q: "queue.Queue" = queue.Queue()
        self._waiters: Dict[str, Any] = {}
        self._lock = threading.Lock()
        self._wake = threading.Event()
        self._stop = threading.Event()
        self._thread: Optional[threading.Thread] = None
        self.failed: Optional[str] = None  # why the engine loop died, if it did
        self._version = 0
        self._paused = False
        self.stats = dict(prefill_steps=0, decode_steps=0, prefill_time=0.0, decode_time=0.0, gen_tokens=0,
                          prompt_tokens=0, aborted=0, pipelined_plans=0)
        self._copy_stream = None
        self._stream = None
        self._inflight: Optional[Dict[str, Any]] = None  # decode plan running on the device, not booked yet
        self._last_decode_done = 0.0
        self.pipeline_decode = os.environ.get("AREAL_PIPELINE_DECODE", "1") != "0"
        self._free_penalty_slots: List[int] = []
        self._slots_to_clear: List[int] = []
        self._mm_ids: Dict[str, int] = {}  # request id -> scheduler id of requests that carry images
        self._streams: Dict[str, list] = {}  # streaming requests: request id -> [scheduler id, tokens sent, callback]
        self.update_group = None
        self.update_groups: Dict[str, Any] = {}
        # tensor-parallel serving: rank 0 of the TP group owns the scheduler and tells the other ranks
        # what to run (see inference/tp_worker.py)
        self.is_leader = tp_rank == 0
        self._tp_src = 0
        self._last_bcast = time.monotonic()
        if tp_size > 1:
            import torch.distributed as dist

            assert tp_group is not None or dist.is_initialized(), "tensor-parallel serving needs a process group"
            self._tp_src = dist.get_global_rank(tp_group, 0) if tp_group is not None else 0

    # ---------------------------------------------------------------------- TP command fan-out
    def _tp_bcast(self, cmd: str, payload: Any = None):
        """Leader only: send one command to the follower ranks (host objects, gloo / RCCL bootstrap)."""
        if self.tp_size > 1 and self.is_leader:
            import torch.distributed as dist

            dist.broadcast_object_list([(cmd, payload)], src=self._tp_src, group=self.tp_group)
            self._last_bcast = time.monotonic()

    # ------------------------------------------------------------------------------ set-up
    def initialize(self, load_weights: bool = True):
        cfg, mc = self.config, self.model_config
        cfg.reject_unsupported()
        self.model = Transformer(mc, dtype=self.dtype, device=self.device, tp_size=self.tp_size, tp_rank=self.tp_rank,
                                 tp_group=self.tp_group)
        self.model.requires_grad_(False)
        self.model.eval()
        self.model.flatten()
        if load_weights and cfg.model_path and os.path.isdir(cfg.model_path):
            hf_io.load_hf_weights(self.model, cfg.model_path)
        elif self.tp_size > 1:
            # random weights must agree across the TP group: build the unsharded model from one seed on
            # the host and keep this rank's slices
            full = Transformer(mc, dtype=torch.float32, device="cpu")
            full.init_weights(seed=1)
            assign_hf_named_tensors(self.model, dict(hf_io.native_to_hf_state(full, gather=False)))
            del full
        else:
            self.model.init_weights(seed=1)
        ctx = int(cfg.context_length or mc.max_position_embeddings)
        hkv = self.model.layers[0].attn.hkv
        kv_fp8 = str(cfg.kv_cache_dtype) in ("fp8", "fp8_e4m3")
        per_page = _kv_bytes_per_page(mc, len(self.model.layers), hkv, 1 if kv_fp8 else 2)
        if self._num_pages_arg is not None:
            num_pages = int(self._num_pages_arg)
        elif self._kv_cache_gb is not None:
            num_pages = int(self._kv_cache_gb * 1024**3 // per_page)
        elif self.device.type == "cuda":
            free, total = torch.cuda.mem_get_info(self.device)
            budget = max(0, free - (1.0 - float(cfg.mem_fraction_static or 0.9)) * total)
            num_pages = int(budget // per_page)
        else:
            num_pages = 256
        num_pages = max(8, num_pages)
        max_running = int(cfg.max_running_requests or 256)
        self.decode_steps = max(1, int(cfg.num_continuous_decode_steps))
        self.runner = ModelRunner(self.model, num_pages, ctx, max_running, decode_steps=self.decode_steps,
                                  use_graph=not cfg.disable_cuda_graph, seed=int(cfg.random_seed),
                                  buckets=cfg.cuda_graph_bs, kv_dtype=torch.float8_e4m3fn if kv_fp8 else None)
        self.num_pages = num_pages
        self.runner.tune_decode_paths()
        if (self.tp_size > 1 and self.device.type == "cuda" and not cfg.disable_custom_all_reduce
                and self.tp_size <= 8 and mc.n_experts == 0 and self.dtype == torch.bfloat16
                and os.environ.get("AREAL_CUSTOM_ALL_REDUCE", "1") != "0"):
            # decode activations [batch, hidden] go through the peer-memory kernel; all ranks of one server share a
            # node (tp_worker starts them there), which is what mapping each other's memory needs
            from ..ops.ipc_allreduce import IpcAllReduce

            ar = IpcAllReduce(self.tp_group, self.device, min(max_running, 512) * mc.hidden_size)
            if ar.self_check():
                self.runner.ipc_ar = ar
                logger.info(f"tensor-parallel all-reduce of <= {ar.cap} elements over peer memory")
            else:
                ar.close()
                logger.warning("peer-memory all-reduce failed its self-check on this machine: using the group's collective")
        self._free_penalty_slots = list(range(max_running - 1, -1, -1))
        if not self.is_leader:
            return self  # followers execute the leader's plans in `run_follower`
        sc = _runtime.SchedulerConfig()
        sc.num_pages = num_pages
        sc.max_running_requests = max_running
        sc.max_prefill_tokens = int(cfg.max_prefill_tokens)
        sc.context_length = ctx
        sc.decode_steps = self.decode_steps
        sc.enable_prefix_cache = not cfg.disable_radix_cache
        sc.policy = cfg.schedule_policy if cfg.schedule_policy in ("fcfs", "lpm") else "fcfs"
        sc.schedule_conservativeness = float(cfg.schedule_conservativeness)
        self.sched = _runtime.Scheduler(sc)
        self.num_pages = num_pages
        logger.info(f"rollout engine ready: {num_pages} KV pages ({num_pages * per_page / 2**30:.1f} GiB), "
                    f"context {ctx}, max running {max_running}, decode steps/plan {self.decode_steps}")
        self._thread = threading.Thread(target=self._loop, daemon=True, name="rollout-engine-loop")
        self._thread.start()
        return self

    def destroy(self):
        self._stop.set()
        self._wake.set()
        if self._thread is not None:
            self._thread.join(timeout=30)
        self.runner = None
        self.model = None

    # -------------------------------------------------------------------------- step loop
    def _loop(self):
        """One thread drives the scheduler and the device. Decode plans are pipelined: while plan n runs, its
        continuation n+1 (same rows, device state carried over) is scheduled and launched, and only then the
        tokens of plan n are awaited and booked - the device goes from one graph replay straight into the next.
        Anything that changes the batch (a prefill that is due, a pause, an abort, new weights) first lets the
        plan in flight finish."""
        if self.device.type == "cuda":
            torch.cuda.set_device(self.device)
            # the engine owns a stream: decode graphs overlap with whatever the trainer thread runs on
            # its stream (colocated asynchronous training)
            # high priority: decode kernels are small (a few hundred workgroups) and latency-bound; when the
            # trainer's large GEMMs share the GPU they would otherwise queue behind full-chip grids, while with
            # priority they take the CUs they need and the trainer's grids fill the rest
            prio = int(os.environ.get("AREAL_ROLLOUT_STREAM_PRIORITY", "-1"))
            self._stream = torch.cuda.Stream(device=self.device, priority=prio)
            torch.cuda.set_stream(self._stream)
        try:
            while not self._stop.is_set():
                self._drain_control()
                self._poll_pull()
                self._drain_requests()
                if self._inflight is not None:
                    self._advance_pipeline()
                    continue
                if self._paused or not self.sched.has_work():
                    self._deliver()
                    if self.tp_size > 1 and time.monotonic() - self._last_bcast > 30.0:
                        self._tp_bcast("noop")  # keeps the followers' blocking receive inside its timeout
                    self._wake.wait(timeout=0.01)
                    self._wake.clear()
                    continue
                plan = self.sched.schedule()
                kind = plan["kind"]
                if kind == 0:
                    self._deliver()
                    self._wake.wait(timeout=0.002)
                    self._wake.clear()
                    continue
                if kind == 1:
                    t0 = time.perf_counter()
                    self._attach_slot_clears(plan)
                    self._tp_bcast("prefill", plan)
                    self.runner.clear_penalty_slots(plan["clear_penalty_slots"])
                    toks, lps = self.runner.prefill(plan)
                    self.sched.process_prefill(toks, lps, plan["plan_id"])
                    self.stats["prefill_steps"] += 1
                    self.stats["prefill_time"] += time.perf_counter() - t0
                    self._deliver()
                else:
                    self._launch_decode(plan)
        except Exception as e:
            traceback.print_exc()
            # the engine is dead from here on: running requests end as aborted, new ones are refused and
            # `/health` reports it, so a supervisor (launcher/server_wrapper.py) restarts the server
            self.failed = f"{type(e).__name__}: {e}"
            self._fail_all("engine loop crashed")
        finally:
            if self.tp_size > 1:
                try:
                    self._tp_bcast("shutdown")
                except Exception:  # the group may already be gone at interpreter exit
                    pass

    def _attach_slot_clears(self, plan):
        """Count-table rows handed to new requests since the last plan: every rank zeroes them before it runs
        the plan."""
        plan["clear_penalty_slots"] = self._slots_to_clear
        self._slots_to_clear = []

    def _launch_decode(self, plan):
        self._attach_slot_clears(plan)
        self._tp_bcast("decode", plan)
        self.runner.clear_penalty_slots(plan["clear_penalty_slots"])
        self._inflight = dict(plan=plan, handle=self.runner.decode_launch(plan), t0=time.perf_counter())

    def _advance_pipeline(self, allow_successor: bool = True):
        """Book the plan in flight. Unless something needs the batch to change, its continuation is launched
        FIRST, so the device has work queued while the host waits for, and processes, this plan's tokens."""
        cur, nxt = self._inflight, None
        if (allow_successor and self.pipeline_decode and not self._paused and self._ctrl_q.empty()
                and getattr(self, "_pending_pull", None) is None):
            # live rows must still fill more than the next smaller batch bucket, or re-packing pays
            live_floor = self.runner.smaller_bucket(len(cur["plan"]["req_ids"])) + 1
            cont = self.sched.schedule_continuation(cur["plan"]["plan_id"], live_floor)
            if cont["kind"] == 2:
                self._attach_slot_clears(cont)
                self._tp_bcast("decode", cont)
                # a penalised request queued while the batch is full gets its count row here (the followers zero
                # it when they receive this plan): zero it on this rank too, before its prefill samples from it
                self.runner.clear_penalty_slots(cont["clear_penalty_slots"])
                nxt = dict(plan=cont, handle=self.runner.decode_launch(cont), t0=time.perf_counter())
                self.stats["pipelined_plans"] += 1
        toks, lps = self.runner.decode_wait(cur["handle"])
        self.sched.process_decode(toks, lps, cur["plan"]["plan_id"], nxt is not None)
        self.stats["decode_steps"] += 1
        now = time.perf_counter()
        # device time is shared between overlapping plans: count from the later of launch / previous completion
        self.stats["decode_time"] += now - max(cur["t0"], self._last_decode_done)
        self._last_decode_done = now
        self._inflight = nxt
        self._deliver()

    def _finish_inflight(self):
        """Wait for and book whatever is running on the device (no successor is launched)."""
        while self._inflight is not None:
            self._advance_pipeline(allow_successor=False)

    def run_follower(self):
        """Loop of TP ranks > 0: receive the leader's commands and run them on this rank's shard. Sampling
        is keyed by (seed, step counter) and the LM head is replicated, so every rank draws the same
        tokens; the runner additionally broadcasts the leader's tokens so that ranks can never diverge."""
        import torch.distributed as dist

        assert not self.is_leader
        if self.device.type == "cuda":
            torch.cuda.set_device(self.device)
        while True:
            box = [None]
            dist.broadcast_object_list(box, src=self._tp_src, group=self.tp_group)
            cmd, payload = box[0]
            if cmd == "prefill":
                self.runner.clear_penalty_slots(payload.get("clear_penalty_slots", []))
                self.runner.prefill(payload)
            elif cmd == "decode":
                # launched, never awaited: the leader may send the continuation before this plan has finished
                # (its tokens arrive through the per-step broadcast inside the plan, not through the host)
                self.runner.clear_penalty_slots(payload.get("clear_penalty_slots", []))
                self.runner.decode_launch(payload)
            elif cmd == "load_disk":
                hf_io.load_hf_weights(self.model, payload)
            elif cmd == "init_update_group":
                self._join_update_group(payload)
            elif cmd == "recv_weights":
                self._recv_weights(payload)
            elif cmd == "pull_ipc":  # stream order puts the switch behind every plan launched so far
                self._pull_sharded(payload, wait=True)
                name, off, shape = self.model.flat.specs[-1]
                self._flip_staging((off + int(math.prod(shape))) * self.model.flat.flat.element_size())
            elif cmd == "shutdown":
                return
            elif cmd != "noop":
                raise RuntimeError(f"unknown TP command {cmd!r}")

    def _drain_requests(self):
        while True:
            try:
                rid, prompt, sp, mm = self._req_q.get_nowait()
            except queue.Empty:
                return
            sched_prompt = prompt
            if mm is not None and self.runner.is_vl:
                assert self.tp_size == 1, "vision-language serving is single-rank"
                sched_prompt = self.runner.salted_prompt(prompt, mm["pixel_values"], mm["image_grid_thw"])
            slot = -1
            if sp["frequency_penalty"] != 0.0 or sp["presence_penalty"] != 0.0:
                if not self._free_penalty_slots:  # more penalised requests than batch rows: wait for a free row
                    self._req_q.put((rid, prompt, sp, mm))
                    return
                slot = self._free_penalty_slots.pop()
                self._slots_to_clear.append(slot)
            req_id = self.sched.add_request(rid, sched_prompt, sp["max_new_tokens"], sp["min_new_tokens"],
                                            sp["temperature"], sp["top_p"], sp["top_k"], sp["stop_token_ids"],
                                            sp["stop_sequences"], sp["frequency_penalty"], sp["presence_penalty"], slot)
            if sp.get("on_tokens") is not None:
                self._streams[rid] = [req_id, 0, sp["on_tokens"]]
            if mm is not None and self.runner.is_vl:
                self.runner.register_multimodal(req_id, prompt, mm["pixel_values"], mm["image_grid_thw"])
                self._mm_ids[rid] = req_id

    def _drain_control(self):
        while True:
            try:
                fn, done = self._ctrl_q.get_nowait()
            except queue.Empty:
                return
            self._finish_inflight()  # control verbs act on a quiescent device and a fully booked scheduler
            try:
                res = fn()
                done.put((True, res))
            except Exception as e:  # surfaced to the caller
                traceback.print_exc()
                done.put((False, e))

    def _flush_streams(self):
        """Hand streaming requests the tokens they have produced since the last call."""
        for rid, st in self._streams.items():
            p = self.sched.progress(st[0], st[1])
            if p is not None and p[0]:
                st[1] += len(p[0])
                st[2](list(p[0]), list(p[1]), list(p[2]))

    def _deliver(self):
        if self._streams:
            self._flush_streams()
        for f in self.sched.take_finished():
            st = self._streams.pop(f["rid"], None)
            if st is not None and len(f["output_tokens"]) > st[1]:  # the tail produced by the request's last plan
                st[2](list(f["output_tokens"][st[1]:]), list(f["output_logprobs"][st[1]:]),
                      list(f["output_versions"][st[1]:]))
            res = GenerationResult(rid=f["rid"], output_tokens=list(f["output_tokens"]),
                                   output_logprobs=list(f["output_logprobs"]),
                                   output_versions=list(f["output_versions"]), finish_reason=f["finish_reason"],
                                   prompt_len=f["prompt_len"], latency=f["latency"], ttft=f["ttft"], itl=list(f["itl"]))
            self.stats["gen_tokens"] += len(res.output_tokens)
            if f["penalty_slot"] >= 0:
                self._free_penalty_slots.append(int(f["penalty_slot"]))
            if res.rid in self._mm_ids:
                self.runner.release_multimodal(self._mm_ids.pop(res.rid))
            if res.finish_reason == "abort":
                self.stats["aborted"] += 1
            with self._lock:
                w = self._waiters.pop(res.rid, None)
            if w is not None:
                w(res)

    def _fail_all(self, msg):
        self._streams.clear()
        with self._lock:
            ws = list(self._waiters.items())
            self._waiters.clear()
        for rid, w in ws:
            w(GenerationResult(rid=rid, output_tokens=[], output_logprobs=[], output_versions=[],
                               finish_reason="abort"))

    def _control(self, fn: Callable, timeout: float = 600.0):
        """Run `fn` on the loop thread between two steps and return its result."""
        if self._thread is None or not self._thread.is_alive() or threading.current_thread() is self._thread:
            return fn()
        done: "queue.Queue" = queue.Queue()
        self._ctrl_q.put((fn, done))
        self._wake.set()
        ok, res = done.get(timeout=timeout)
        if not ok:
            raise res
        return res

    # ------------------------------------------------------------------------ request API
    def submit(self, rid: str, input_ids: List[int], *, max_new_tokens: int, min_new_tokens: int = 0,
               temperature: float = 1.0, top_p: float = 1.0, top_k: int = 1 << 30,
               stop_token_ids: Optional[List[int]] = None, greedy: bool = False,
               callback: Optional[Callable[[GenerationResult], None]] = None,
               multimodal: Optional[Dict[str, torch.Tensor]] = None,
               stop_sequences: Optional[List[List[int]]] = None, frequency_penalty: float = 0.0,
               presence_penalty: float = 0.0,
               on_tokens: Optional[Callable[[List[int], List[float], List[int]], None]] = None):
        """`on_tokens(tokens, logprobs, versions)`: streaming - called from the engine thread with every batch of
        newly generated tokens (one call per processed plan), before `callback` receives the complete result.
        `multimodal`: {"pixel_values" [n_patches, C*tp*P*P], "image_grid_thw" [n_images, 3]} for prompts
        whose `input_ids` contain image placeholder tokens (vision-language models). `stop_sequences`: token-id
        sequences; generation stops (reason "stop") when the generated tokens end with one of them.
        `frequency_penalty` / `presence_penalty`: logit - f * count - p * (count > 0) over the tokens this
        request has generated so far (OpenAI / SGLang semantics)."""
        sp = dict(max_new_tokens=int(max_new_tokens), min_new_tokens=int(min_new_tokens),
                  temperature=0.0 if greedy else float(temperature), top_p=float(top_p),
                  top_k=int(min(top_k, 1 << 30)), stop_token_ids=[int(x) for x in (stop_token_ids or [])],
                  stop_sequences=[[int(t) for t in seq] for seq in (stop_sequences or []) if len(seq)],
                  frequency_penalty=float(frequency_penalty), presence_penalty=float(presence_penalty),
                  on_tokens=on_tokens)
        if self.failed is not None:
            raise RuntimeError(f"the rollout engine has stopped ({self.failed})")
        with self._lock:
            if rid in self._waiters:
                raise ValueError(f"duplicate request id {rid}")
            self._waiters[rid] = callback or (lambda r: None)
        if multimodal is not None and multimodal.get("pixel_values") is not None:
            multimodal = dict(pixel_values=multimodal["pixel_values"].detach().cpu(),
                              image_grid_thw=torch.as_tensor(multimodal["image_grid_thw"]).cpu().long())
        else:
            multimodal = None
        self._req_q.put((rid, [int(t) for t in input_ids], sp, multimodal))
        self._wake.set()

    async def agenerate_once(self, rid: str, input_ids: List[int], **sampling) -> GenerationResult:
        loop = asyncio.get_running_loop()
        fut = loop.create_future()

        def cb(res):
            loop.call_soon_threadsafe(lambda: fut.done() or fut.set_result(res))

        self.submit(rid, input_ids, callback=cb, **sampling)
        return await fut

    def generate_blocking(self, prompts: List[List[int]], **sampling) -> List[GenerationResult]:
        """Synchronous helper: run all prompts to completion (no interruption handling)."""
        out: List[Optional[GenerationResult]] = [None] * len(prompts)
        ev = threading.Event()
        remaining = [len(prompts)]
        lock = threading.Lock()

        def mk(i):
            def cb(res):
                out[i] = res
                with lock:
                    remaining[0] -= 1
                    if remaining[0] == 0:
                        ev.set()
            return cb

        base = f"blk-{time.monotonic_ns()}"
        for i, p in enumerate(prompts):
            self.submit(f"{base}-{i}", p, callback=mk(i), **sampling)
        if prompts:
            ev.wait()
        return out  # type: ignore

    # ------------------------------------------------------------------------ control API
    def pause(self):
        def f():
            self._paused = True
            self.sched.pause()
        self._control(f)

    def resume(self):
        def f():
            self._paused = False
            self.sched.resume()
        self._control(f)
        self._wake.set()

    def abort_all(self) -> int:
        def f():
            n = self.sched.abort_all()
            self._deliver()
            return n
        return self._control(f)

    def abort(self, rid: str) -> bool:
        """Abort one request (its waiter receives finish reason "abort"); False if the scheduler does not hold it."""
        def f():
            ok = self.sched.abort(rid)
            self._deliver()
            return ok
        return self._control(f)

    def flush_cache(self):
        self._control(lambda: self.sched.flush_cache())

    def get_version(self) -> int:
        return self._version

    def set_version(self, v: int):
        def f():
            self._version = int(v)
            self.sched.set_version(int(v))
        self._control(f)

    # ------------------------------------------------------------------------ weight loading
    def _after_weight_change(self, version: Optional[int]):
        self.sched.flush_cache()
        if version is not None:
            self._version = int(version)
            self.sched.set_version(int(version))

    def load_weights_from_trainer(self, trainer_model: Transformer, version: Optional[int] = None,
                                  abort_running: bool = True):
        """Colocated update: one device-to-device copy of the flat bf16 buffer on a side stream, applied
        between two engine steps; running requests are aborted first (their KV is stale)."""
        assert self.tp_size == 1, "the flat-buffer copy needs identical layouts; TP engines load by name"
        src = trainer_model.flat.flat
        cuda = self.device.type == "cuda"
        ready = done = None
        if cuda:  # the optimizer step that produced `src` runs on the caller's stream
            ready, done = torch.cuda.Event(), torch.cuda.Event()
            ready.record(torch.cuda.current_stream())

        def f():
            if abort_running:
                self.sched.abort_all()
                self._deliver()
            # Same parameters at the same offsets; only the zero padding at the end differs (the trainer
            # pads its buffer to a multiple of the data-parallel size for the ZeRO shards).
            assert trainer_model.flat.specs == self.model.flat.specs, "trainer and rollout parameter layouts differ"
            name, off, shape = self.model.flat.specs[-1]
            n = off + int(math.prod(shape))
            dst, src_ = self.model.flat.flat[:n], src[:n]
            if cuda:
                if self._copy_stream is None:
                    self._copy_stream = torch.cuda.Stream(device=self.device)
                self._copy_stream.wait_stream(torch.cuda.current_stream())
                self._copy_stream.wait_event(ready)
                with torch.cuda.stream(self._copy_stream):
                    dst.copy_(src_, non_blocking=True)
                    done.record(self._copy_stream)
                torch.cuda.current_stream().wait_stream(self._copy_stream)
            else:
                dst.copy_(src_)
            self._after_weight_change(version)
        self._control(f)
        if cuda:  # the caller must not overwrite `src` before the copy has read it
            torch.cuda.current_stream().wait_event(done)

    def load_weights_from_disk(self, path: str, version: Optional[int] = None, abort_running: bool = True):
        def f():
            if abort_running:
                self.sched.abort_all()
                self._deliver()
            self._tp_bcast("load_disk", path)
            hf_io.load_hf_weights(self.model, path)
            self._after_weight_change(version)
        self._control(f)

    def load_named_tensors(self, tensors: Dict[str, torch.Tensor], version: Optional[int] = None,
                           abort_running: bool = True, finalize: bool = True):
        """Assign HF-named full tensors (weight updates received through a collective)."""
        def f():
            if abort_running:
                self.sched.abort_all()
                self._deliver()
            assign_hf_named_tensors(self.model, tensors)
            if finalize:
                self._after_weight_change(version)
        self._control(f)

    # ------------------------------------------------------------ weight updates from peer memory
    def update_weights_from_ipc(self, buffer: Optional[Dict[str, Any]], layout: str, used_bytes: int,
                                version: Optional[int] = None, abort_running: bool = True,
                                buffers: Optional[List[Dict[str, Any]]] = None, src: Optional[Dict[str, Any]] = None,
                                replica: int = 0) -> Dict[str, float]:
        """Pull new weights out of the trainer's exported memory.

        Same layout on both sides (`buffer`: the snapshot of ONE unsharded trainer replica): one copy of the flat
        buffer. Different layouts (`buffers`: the snapshots of ALL trainer ranks, `src`: the trainer's layout -
        tensor- / pipeline-parallel slices, ZeRO-3 shards - or a tensor-parallel server): every rank of this server
        works out which element ranges of which trainer ranks make up ITS flat buffer
        (``parallel/param_realloc.py::derive_pull_plan``) and copies exactly those, one kernel launch per source
        piece; the trainer gathers nothing and every trainer GPU serves only what it owns.

        All device work is issued by the engine's loop thread. Phase 1: between two engine steps the loop maps
        the snapshots (first time) and starts the copies into a staging buffer on a side stream (xGMI between two
        GPUs, local copies when trainer and server share one) - decoding goes on while they run. Phase 2: the loop
        polls the copies' event between steps; once it has fired, running requests are aborted (their KV cache
        belongs to the old weights; the client re-issues them), staging -> live parameters (an on-device copy,
        ~2 ms for a 1.5B model; the live buffer's address is baked into the captured decode graphs, so the buffers
        are not swapped), prefix cache flushed, version bumped. A tensor-parallel server does both phases in one
        go on every rank, between two plans (its ranks must switch weights at the same plan boundary).
        Blocks the caller until the flip is done; returns timings."""
        from ..engine.train_engine import flat_layout_fingerprint

        flat = self.model.flat.flat
        sharded = bool(src) and (int(src.get("tp", 1)) > 1 or int(src.get("pp", 1)) > 1 or int(src.get("zero3", 1)) > 1
                                 or int(src.get("ep", 1) or 1) > 1 or self.tp_size > 1)
        if not sharded:
            if buffer is None:
                buffer = buffers[replica % len(buffers)]
            if layout and layout != flat_layout_fingerprint(self.model):
                raise ValueError("the trainer's flat parameter layout differs from this server's (model config, dtype "
                                 "or tied-embedding setting)")
            n = int(used_bytes) or min(int(buffer["nbytes"]), flat.numel() * flat.element_size())
            assert n <= flat.numel() * flat.element_size(), "trainer buffer holds more parameters than this model"
        else:
            assert buffers, "a sharded pull needs the exported buffers of all trainer ranks"
            name, off, shape = self.model.flat.specs[-1]
            n = (off + int(math.prod(shape))) * flat.element_size()
        done: "queue.Queue" = queue.Queue()
        t0 = time.perf_counter()
        payload = dict(buffers=buffers, src=src, replica=replica)

        def start():
            if self.tp_size > 1:  # every rank switches at this plan boundary: pull and flip now, everywhere
                t1 = time.perf_counter()
                self._tp_bcast("pull_ipc", payload)
                self._pull_sharded(payload, wait=True)
                if abort_running:
                    self.sched.abort_all()
                    self._deliver()
                self._flip_staging(n)
                self._after_weight_change(version)
                t2 = time.perf_counter()
                done.put((True, dict(pull_ms=(t2 - t1) * 1e3, flip_ms=0.0, total_ms=(t2 - t0) * 1e3, bytes=float(n))))
                return
            event = None
            if self.device.type == "cuda":
                if self._copy_stream is None:
                    self._copy_stream = torch.cuda.Stream(device=self.device)
                with torch.cuda.stream(self._copy_stream):
                    self._issue_pull(buffer, n, payload if sharded else None)
                    event = torch.cuda.Event()
                    event.record(self._copy_stream)
            else:
                self._issue_pull(buffer, n, payload if sharded else None)
            self._pending_pull = dict(event=event, n=n, version=version, abort=abort_running, done=done, t0=t0,
                                      t_issue=time.perf_counter())

        self._control(start)
        ok, info = done.get(timeout=600)
        if not ok:
            raise info
        return info

    def _staging(self) -> torch.Tensor:
        if getattr(self, "_ipc_staging", None) is None:
            self._ipc_staging = torch.empty_like(self.model.flat.flat)
        return self._ipc_staging

    def _peer(self, desc: Dict[str, Any]):
        from ..ops import ipc

        peers = self.__dict__.setdefault("_ipc_peers", {})
        if desc["key"] not in peers:
            peers[desc["key"]] = ipc.PeerBuffer(desc)
        return peers[desc["key"]]

    def _issue_pull(self, buffer: Optional[Dict[str, Any]], n: int, payload: Optional[Dict[str, Any]]):
        """Queue the copies of one weight pull into the staging buffer on the current stream."""
        if payload is None:
            self._peer(buffer).copy_into(self._staging(), n)
        else:
            self._pull_sharded(payload, wait=False)

    def _pull_sharded(self, payload: Dict[str, Any], wait: bool):
        """This rank's ranges of the trainer ranks' exported buffers -> staging (current stream). The plan depends
        on the two layouts only: derived once, its device-side range lists kept for every later update."""
        from ..parallel.param_realloc import Layout, derive_expert_parallel_pull_plan, derive_pull_plan

        src, buffers, replica = payload["src"], payload["buffers"], int(payload.get("replica", 0))
        experts = src.get("expert_buffers") or []
        lay_items = sorted((k, v) for k, v in src.items() if k != "expert_buffers")
        key = (repr(lay_items), replica, tuple(b["key"] for b in buffers), tuple(b["key"] for b in experts if b))
        cache = self.__dict__.setdefault("_ipc_plans", {})
        staging = self._staging()
        if key not in cache:
            esz = staging.element_size()
            if int(src.get("ep", 1) or 1) > 1:  # expert-parallel trainer: two exported buffers per rank
                assert len(experts) == len(buffers), "one expert buffer descriptor (or None) per trainer rank"
                plan = derive_expert_parallel_pull_plan(self.model_config, dict(lay_items), self.tp_size, self.tp_rank,
                                                        replica, dtype=self.dtype)
                peers = [self._peer(buffers[r] if which == "flat" else experts[r]) for r, which, _ in plan]
                cache[key] = [(pb, pb.prepare_moves(tri, esz, self.device)) for pb, (_, _, tri) in zip(peers, plan)]
            else:
                lay = Layout(dp=int(src.get("dp", 1)), tp=int(src.get("tp", 1)), pp=int(src.get("pp", 1)),
                             vocab_parallel=bool(src.get("vocab_parallel", int(src.get("tp", 1)) > 1)),
                             vpp=int(src.get("vpp", 1) or 1))
                assert lay.world == len(buffers), f"trainer layout {lay} does not match {len(buffers)} exported buffers"
                plan = derive_pull_plan(self.model_config, lay, self.tp_size, self.tp_rank, replica,
                                        dtype=self.dtype, zero3=int(src.get("zero3", 1)))
                cache[key] = [(self._peer(buffers[r]), self._peer(buffers[r]).prepare_moves(tri, esz, self.device))
                              for r, tri in plan]
        for peer, prepared in cache[key]:
            peer.move_intervals(staging, prepared)
        if wait and self.device.type == "cuda":
            torch.cuda.current_stream().synchronize()

    def _flip_staging(self, n: int):
        flat = self.model.flat.flat
        flat.view(torch.uint8).view(-1)[:n].copy_(self._staging().view(torch.uint8).view(-1)[:n])
        if self.device.type == "cuda":
            torch.cuda.current_stream().synchronize()

    def _poll_pull(self):
        """Loop thread, between steps: finish a peer-memory weight pull whose copy has completed."""
        p = getattr(self, "_pending_pull", None)
        if p is None or (p["event"] is not None and not p["event"].query()):
            return
        self._finish_inflight()
        self._pending_pull = None
        try:
            t1 = time.perf_counter()
            if p["abort"]:
                self.sched.abort_all()
                self._deliver()
            n = p["n"]
            self._flip_staging(n)
            self._after_weight_change(p["version"])
            t2 = time.perf_counter()
            p["done"].put((True, dict(pull_ms=(t1 - p["t_issue"]) * 1e3, flip_ms=(t2 - t1) * 1e3,
                                      total_ms=(t2 - p["t0"]) * 1e3, bytes=float(n))))
        except Exception as e:  # surfaced to the caller
            traceback.print_exc()
            p["done"].put((False, e))

    # ------------------------------------------------------- weight updates over a collective
    def _join_update_group(self, body: Dict[str, Any]):
        from ..utils.distributed import init_custom_process_group

        rank = int(body["rank_offset"]) + self.tp_rank
        group = init_custom_process_group(
            backend=body.get("backend", "nccl"), world_size=int(body["world_size"]), rank=rank,
            init_method=f"tcp://{body['master_address']}:{body['master_port']}", group_name=body["group_name"])
        self.update_groups[body["group_name"]] = group
        self.update_group = group

    def init_weight_update_group(self, body: Dict[str, Any]):
        """Join a weight-update group of the trainer: the sending trainer rank is rank 0, TP rank r of this
        server is rank ``rank_offset + r`` (``areal/engine/sglang_remote.py:452-480`` in the reference). With
        several senders the trainer opens one group per sender, each under its own name."""
        def f():
            self._tp_bcast("init_update_group", body)
            self._join_update_group(body)
        self._control(f)

    def _recv_weights(self, groups: List[Dict[str, Any]]):
        """`groups`: [{group_name, names, dtypes, shapes}]: the chunks of one round, one per sender; their
        broadcasts are all started before the first one is waited for."""
        import torch.distributed as dist

        tensors, works = {}, []
        for g in groups:
            group = self.update_groups.get(g.get("group_name")) or self.update_group
            for n, dt, sh in zip(g["names"], g["dtypes"], g["shapes"]):
                t = torch.empty(sh, dtype=getattr(torch, str(dt).replace("torch.", "")), device=self.device)
                works.append(dist.broadcast(t, src=0, group=group, async_op=True))
                tensors[n] = t
        for w in works:
            w.wait()
        assign_hf_named_tensors(self.model, tensors)

    def update_weights_from_distributed(self, names=None, dtypes=None, shapes=None, version: Optional[int] = None,
                                        abort_running: bool = True, group_name: Optional[str] = None,
                                        groups: Optional[List[Dict[str, Any]]] = None):
        """Receive HF-named full tensors broadcast by the trainer's sender rank(s); every TP rank keeps its
        slices."""
        if groups is None:
            groups = [dict(group_name=group_name, names=names, dtypes=dtypes, shapes=shapes)]

        def f():
            if abort_running:
                self.sched.abort_all()
                self._deliver()
            self._tp_bcast("recv_weights", groups)
            self._recv_weights(groups)
            self._after_weight_change(version)
        self._control(f)

    # ----------------------------------------------------------------------------- metrics
    def metrics(self) -> Dict[str, float]:
        s = self.sched
        total = max(1, s.num_total_pages())
        return {
            "num_running_reqs": s.num_running(), "num_queue_reqs": s.num_waiting(),
            "token_usage": 1.0 - s.num_free_pages() / total, "num_used_pages": total - s.num_free_pages(),
            "cached_pages": s.num_cached_pages(), "gen_tokens_total": s.total_generated(),
            "prompt_tokens_total": s.total_prompt_tokens(), "cached_tokens_total": s.total_cached_tokens(),
            "preemptions_total": s.num_preemptions(), "prefix_cache_collisions": s.num_prefix_collisions(),
            "weight_version": self._version,
            **{k: float(v) for k, v in self.stats.items()},
        }


def _assign_gpt2_named_tensors(model: Transformer, tensors: Dict[str, torch.Tensor], strict: bool):
    """GPT-2 checkpoint names (`models/hf_io.py::_gpt2_state` writes them): linears are stored [in, out], q|k|v
    fused along the output axis of `c_attn`; any subset of the model's tensors may arrive in one call."""
    import re

    cfg = model.cfg
    D, H = cfg.head_dim, cfg.hidden_size
    tp, tr = model.tp_size, model.tp_rank
    params = dict(model.named_parameters())
    local = {g: li for li, g in enumerate(model.layer_ids)}
    v_lo, v_hi = model.vocab_range
    plain = {"ln_1.weight": "ln1", "ln_1.bias": "ln1_bias", "ln_2.weight": "ln2", "ln_2.bias": "ln2_bias",
             "attn.c_proj.bias": "attn.o_bias", "mlp.c_proj.bias": "mlp.proj_bias"}
    left = []
    for name, t in tensors.items():
        n = name[len("transformer."):] if name.startswith("transformer.") else name
        m = re.match(r"h\.(\d+)\.(.+)$", n)
        if m:
            if int(m.group(1)) not in local:  # another pipeline stage's layer
                continue
            pre, key = f"layers.{local[int(m.group(1))]}.", m.group(2)
            blk = model.layers[local[int(m.group(1))]]
            q = blk.attn.hq * D
            if key in plain:
                params[pre + plain[key]].data.copy_(t)
            elif key == "attn.c_attn.weight":
                params[pre + "attn.qkv"].data.copy_(torch.cat([t[:, o + tr * q:o + (tr + 1) * q] for o in (0, H, 2 * H)], 1).t())
            elif key == "attn.c_attn.bias":
                params[pre + "attn.qkv_bias"].data.copy_(torch.cat([t[o + tr * q:o + (tr + 1) * q] for o in (0, H, 2 * H)], 0))
            elif key == "attn.c_proj.weight":
                params[pre + "attn.o"].data.copy_(t[tr * q:(tr + 1) * q].t())
            elif key == "mlp.c_fc.weight":
                f = params[pre + "mlp.fc"].shape[0]
                params[pre + "mlp.fc"].data.copy_(t[:, tr * f:(tr + 1) * f].t())
            elif key == "mlp.c_fc.bias":
                f = params[pre + "mlp.fc_bias"].shape[0]
                params[pre + "mlp.fc_bias"].data.copy_(t[tr * f:(tr + 1) * f])
            elif key == "mlp.c_proj.weight":
                f = params[pre + "mlp.proj"].shape[1]
                params[pre + "mlp.proj"].data.copy_(t[tr * f:(tr + 1) * f].t())
            else:
                left.append(name)
        elif n == "wte.weight":
            if model.has_embed:
                params["embed"].data.copy_(t[v_lo:v_hi])
        elif n == "wpe.weight":
            if model.has_embed:
                params["pos_embed"].data.copy_(t)
        elif n in ("ln_f.weight", "ln_f.bias"):
            if model.has_head:
                params["norm" if n.endswith("weight") else "norm_bias"].data.copy_(t)
        elif n in ("lm_head.weight", "score.weight"):
            if model.has_head and "head" in params:
                params["head"].data.copy_(t if n == "score.weight" else t[v_lo:v_hi])
        else:
            left.append(name)
    if strict and left:
        raise KeyError(f"{len(left)} received tensor(s) map onto no parameter of this model: {sorted(left)[:8]}")


def assign_hf_named_tensors(model: Transformer, tensors: Dict[str, torch.Tensor], strict: bool = True):
    """Copy HF-named FULL tensors into the (possibly tensor-parallel) native parameters. With `strict`, a
    tensor whose name maps onto no parameter of this model raises instead of being dropped (a silent drop
    leaves that part of the rollout model on stale weights)."""
    cfg = model.cfg
    if cfg.family == "gpt2":
        return _assign_gpt2_named_tensors(model, tensors, strict)
    D = cfg.head_dim
    tp, tr = model.tp_size, model.tp_rank
    params = dict(model.named_parameters())
    own_layers = set(model.layer_ids)
    given = set(tensors)

    class _Seen(dict):  # records which of the given names were read
        def __getitem__(self, k):
            given.discard(k)
            return dict.__getitem__(self, k)

    tensors = _Seen(tensors)

    def rows(t, lo, hi):
        return t[lo:hi]

    for li, blk in enumerate(model.layers):
        i = model.layer_ids[li]
        names = hf_io.hf_names_for_layer(cfg, i)
        pre = f"layers.{li}."
        hq, hkv = blk.attn.hq, blk.attn.hkv
        hq_lo = tr * hq
        hkv_lo = (tr * hkv) if cfg.n_kv_heads >= tp else tr // (tp // cfg.n_kv_heads)
        for key, hf_list in names.items():
            p = params[pre + key]
            if key == "attn.qkv" or key == "attn.qkv_bias":
                offs = [0, hq * D, (hq + hkv) * D]
                sizes = [hq * D, hkv * D, hkv * D]
                los = [hq_lo * D, hkv_lo * D, hkv_lo * D]
                for n, o, s, lo in zip(hf_list, offs, sizes, los):
                    if n in tensors:
                        p.data[o:o + s].copy_(tensors[n][lo:lo + s])
            elif key == "attn.o":
                if hf_list[0] in tensors:
                    p.data.copy_(tensors[hf_list[0]][:, hq_lo * D:(hq_lo + hq) * D])
            elif key == "mlp.gate_up":
                f = p.shape[0] // 2
                for j, n in enumerate(hf_list):
                    if n in tensors:
                        p.data[j * f:(j + 1) * f].copy_(tensors[n][tr * f:(tr + 1) * f])
            elif key == "mlp.down":
                if hf_list[0] in tensors:
                    f = p.shape[1]
                    p.data.copy_(tensors[hf_list[0]][:, tr * f:(tr + 1) * f])
            else:
                if hf_list[0] in tensors:
                    p.data.copy_(tensors[hf_list[0]])
        if cfg.n_experts > 0:
            gu, dn = params[pre + "mlp.gate_up"], params[pre + "mlp.down"]
            f = gu.shape[1] // 2
            er = model.etp_rank  # position among the ranks that split an expert (0: whole experts)
            for le in range(gu.shape[0]):
                e = blk.mlp.ep_rank * blk.mlp.n_local + le
                gn, un, dnn = hf_io.hf_expert_names(cfg, i, e)
                if gn in tensors:
                    gu.data[le, :f].copy_(tensors[gn][er * f:(er + 1) * f])
                if un in tensors:
                    gu.data[le, f:].copy_(tensors[un][er * f:(er + 1) * f])
                if dnn in tensors:
                    dn.data[le].copy_(tensors[dnn][:, er * f:(er + 1) * f])
    v_lo, v_hi = model.vocab_range  # the whole vocabulary unless the model shards it over tp
    if model.has_embed and "model.embed_tokens.weight" in tensors:
        params["embed"].data.copy_(tensors["model.embed_tokens.weight"][v_lo:v_hi])
    for n in [k for k in list(given) if k.startswith("visual.")]:  # vision tower: replicated, same names
        if n in params:
            params[n].data.copy_(tensors[n])
    if model.has_head:
        if "model.norm.weight" in tensors:
            params["norm"].data.copy_(tensors["model.norm.weight"])
        if "head" in params:
            if "lm_head.weight" in tensors:
                params["head"].data.copy_(tensors["lm_head.weight"][v_lo:v_hi])
            if "score.weight" in tensors:
                params["head"].data.copy_(tensors["score.weight"])
        elif "lm_head.weight" in tensors:  # tied head: the embedding copy above covers it
            tensors["lm_head.weight"]
    if strict and given:
        import re

        def foreign(n):  # parameters of layers / ends another pipeline stage owns are not ours to consume
            if ".experts." in n:  # routed experts other expert-parallel ranks hold
                return True
            m = re.match(r"model\.layers\.(\d+)\.", n)
            if m:
                return int(m.group(1)) not in own_layers
            if n == "model.embed_tokens.weight" or n.startswith("visual."):
                return not model.has_embed
            return n in ("model.norm.weight", "lm_head.weight", "score.weight") and not model.has_head
        left = sorted(n for n in given if not foreign(n))
        if left:
            raise KeyError(f"{len(left)} received tensor(s) map onto no parameter of this model: {left[:8]}")
