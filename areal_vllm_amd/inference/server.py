"""HTTP front end of the native rollout engine.

Speaks the REST verbs the reference's clients use against SGLang (``areal/engine/sglang_remote.py:83-119,
251-309, 368-480``; ``realhf/system/gserver_manager.py:158-173``):

  GET  /health, /health_generate, /get_model_info, /get_server_info, /metrics (Prometheus text)
  POST /generate {input_ids, sampling_params, return_logprob, stream, rid}   (stream: true -> server-sent events)
       sampling_params: max_new_tokens, min_new_tokens, temperature, top_p, top_k, stop_token_ids,
       stop_sequences (token-id sequences), stop (strings: tokenised here with the server's tokenizer),
       frequency_penalty, presence_penalty, ignore_eos, n (must be 1). Any other key is refused (400): a
       sampling option is either honoured or rejected, never dropped.
  POST /pause_generation, /continue_generation, /flush_cache, /abort_request {rid}
  POST /update_weights_from_disk {model_path, abort_all_requests}
  POST /init_weights_update_group {master_address, master_port, rank_offset, world_size, backend, group_name}
       (once per sending trainer rank; /update_weights_from_distributed then names the group, or lists one
       chunk per group under "groups" when several senders push concurrently)
  POST /update_weights_from_distributed {names, dtypes, shapes, group_name, abort_all_requests}
  POST /set_version {version}

``/generate`` answers with ``meta_info.finish_reason.type`` in {stop, length, abort},
``meta_info.output_token_logprobs`` as ``[(logprob, token_id, None), ...]`` and, beyond SGLang,
``meta_info.output_versions`` (weight version per token).

Run: ``python -m areal_vllm_amd.inference.server --model-path <dir|preset> --host H --port P [flags]``;
flags are the fields of ``InferenceServerConfig``.
"""
import argparse
import asyncio
import dataclasses
import os
import sys
import time
import typing
import uuid
from typing import Any, Dict, List, Optional

import torch
from aiohttp import web

from ..api.cli_args import InferenceServerConfig
from ..utils import logging, stop_strings
from .engine import NativeInferenceEngine

logger = logging.getLogger("inference-server")


SAMPLING_KEYS = {"max_new_tokens", "min_new_tokens", "temperature", "top_p", "top_k", "stop_token_ids",
                 "stop_sequences", "stop", "frequency_penalty", "presence_penalty", "ignore_eos", "n",
                 "skip_special_tokens", "spaces_between_special_tokens", "no_stop_trim"}  # last three: text only
REQUEST_KEYS = {"input_ids", "sampling_params", "return_logprob", "stream", "rid", "image_data", "logprob_start_len",
                "top_logprobs_num", "return_text_in_logprobs"}


class BadRequest(ValueError):
    pass


def parse_sampling_params(sp: Dict[str, Any], stop_to_ids=None) -> Dict[str, Any]:
    """Validate the `sampling_params` of a /generate request and turn them into `NativeInferenceEngine.submit`
    keywords. Unknown keys and options this engine cannot honour raise `BadRequest`. `stop_to_ids(str)` maps a
    stop string to its token ids (None: stop strings are refused)."""
    unknown = sorted(set(sp) - SAMPLING_KEYS)
    if unknown:
        raise BadRequest(f"unsupported sampling parameter(s) {unknown}; supported: {sorted(SAMPLING_KEYS)}")
    if not 1 <= int(1 if sp.get("n") is None else sp["n"]) <= 1024:
        raise BadRequest("n: 1 .. 1024 samples per request")
    temp = float(sp.get("temperature", 1.0))
    top_k = int(sp.get("top_k", -1))
    seqs = [list(map(int, q)) for q in (sp.get("stop_sequences") or [])]
    stops = sp.get("stop") or []
    if isinstance(stops, str):
        stops = [stops]
    if stops:
        if stop_to_ids is None:
            raise BadRequest("stop strings need a tokenizer on the server (start it with --tokenizer-path), or send "
                             "`stop_sequences` (token ids) instead")
        seqs += [ids for ids in (stop_to_ids(x) for x in stops) if ids]
    return dict(max_new_tokens=int(sp.get("max_new_tokens", 128)), min_new_tokens=int(sp.get("min_new_tokens", 0)),
                temperature=temp, top_p=float(sp.get("top_p", 1.0)), top_k=top_k if top_k > 0 else 1 << 30,
                stop_token_ids=[] if sp.get("ignore_eos") else list(sp.get("stop_token_ids") or []),
                stop_sequences=seqs, frequency_penalty=float(sp.get("frequency_penalty", 0.0) or 0.0),
                presence_penalty=float(sp.get("presence_penalty", 0.0) or 0.0), greedy=temp <= 0.0)


def _finish(reason: str) -> Dict[str, Any]:
    if reason == "abort":
        return {"type": "abort", "message": "Abort by weight update or pause"}
    return {"type": reason}


class InferenceServer:
    def __init__(self, engine: NativeInferenceEngine):
        self.engine = engine
        self.update_group = None
        self._image_processor = None
        self._tokenizer = None  # loaded on the first request with stop strings; False: none available
        self.start_time = time.time()
        self.app = web.Application(client_max_size=1024**3)
        r = self.app.router
        r.add_get("/health", self.health)
        r.add_get("/health_generate", self.health)
        r.add_get("/get_model_info", self.model_info)
        r.add_get("/get_server_info", self.server_info)
        r.add_get("/metrics", self.metrics)
        r.add_post("/generate", self.generate)
        r.add_post("/pause_generation", self.pause)
        r.add_post("/continue_generation", self.resume)
        r.add_post("/flush_cache", self.flush_cache)
        r.add_post("/abort_request", self.abort_request)
        r.add_post("/update_weights_from_disk", self.update_from_disk)
        r.add_post("/init_weights_update_group", self.init_update_group)
        r.add_post("/update_weights_from_distributed", self.update_from_distributed)
        r.add_post("/update_weights_from_ipc", self.update_from_ipc)
        r.add_post("/set_version", self.set_version)

    async def health(self, req):
        failed = getattr(self.engine, "failed", None)
        if failed is not None:
            return web.Response(status=503, text=f"engine stopped: {failed}")
        return web.Response(text="OK")

    async def model_info(self, req):
        return web.json_response({"model_path": self.engine.config.model_path, "is_generation": True,
                                  "weight_version": self.engine.get_version()})

    async def server_info(self, req):
        return web.json_response({**dataclasses.asdict(self.engine.config), "version": "native",
                                  "uptime": time.time() - self.start_time, **self.engine.metrics()})

    async def metrics(self, req):
        m = self.engine.metrics()
        lines = []
        alias = {"num_running_reqs": "sglang:num_running_reqs", "num_queue_reqs": "sglang:num_queue_reqs",
                 "token_usage": "sglang:token_usage", "num_used_pages": "sglang:num_used_tokens"}
        for k, v in m.items():
            name = alias.get(k, f"areal:{k}")
            if k == "num_used_pages":
                v = v * 32
            lines.append(f"# TYPE {name} gauge")
            lines.append(f'{name}{{model_name="{self.engine.config.model_path}"}} {float(v)}')
        return web.Response(text="\n".join(lines) + "\n", content_type="text/plain")

    async def generate(self, req):
        body = await req.json()
        ids = body.get("input_ids")
        if ids is None:
            return web.json_response({"error": "this server takes token ids (`input_ids`)"}, status=400)
        rid = body.get("rid") or uuid.uuid4().hex
        try:
            extra = sorted(set(body) - REQUEST_KEYS)
            if extra:
                raise BadRequest(f"unsupported request field(s) {extra}")
            kw = parse_sampling_params(body.get("sampling_params", {}) or {}, self._stop_to_ids())
        except (BadRequest, TypeError, ValueError) as e:
            return web.json_response({"error": str(e)}, status=400)
        multimodal = None
        if body.get("image_data"):
            try:
                multimodal = await self._offload(self._process_images, body["image_data"])
            except Exception as e:
                return web.json_response({"error": f"cannot process image_data: {e!r}"}, status=400)
        n = int((body.get("sampling_params") or {}).get("n", 1) or 1)
        if n > 1 and body.get("stream"):
            return web.json_response({"error": "streamed requests take n = 1"}, status=400)
        if body.get("stream"):
            return await self._generate_stream(req, rid, ids, multimodal, kw)
        stops = (body.get("sampling_params") or {}).get("stop") or []
        stops = [stops] if isinstance(stops, str) else [x for x in stops if x]

        async def one(i):
            engine_rid = f"{rid}-{uuid.uuid4().hex[:8]}"
            if stops and self._tokenizer:
                res = await self._generate_until_text(engine_rid, ids, multimodal, kw, stops)
            else:
                res = await self.engine.agenerate_once(engine_rid, ids, multimodal=multimodal, **kw)
            meta = {
                "id": rid if n == 1 else f"{rid}-{i}", "finish_reason": _finish(res.finish_reason),
                "prompt_tokens": res.prompt_len, "completion_tokens": len(res.output_tokens),
                "output_token_logprobs": [(lp, t, None) for lp, t in zip(res.output_logprobs, res.output_tokens)],
                "output_versions": res.output_versions, "weight_version": self.engine.get_version(),
                "e2e_latency": res.latency, "ttft": res.ttft,
            }
            return {"text": "", "output_ids": res.output_tokens, "meta_info": meta}

        if n == 1:
            return web.json_response(await one(0))
        # n samples of one prompt: n engine requests that share the prompt's pages through the prefix cache (the first
        # one prefills, the others find the pages); the answer is a list, one entry per sample
        return web.json_response(list(await asyncio.gather(*[one(i) for i in range(n)])))

    async def _generate_until_text(self, engine_rid, ids, multimodal, kw, stops: List[str]):
        """Generation that also ends where the DECODED output first contains one of the stop strings. The engine
        matches stop strings as the token-id sequences the tokenizer gives them in isolation; a model can spell the
        same characters with other tokens (merged with a neighbour, split differently), which only the text shows.
        The tokens are watched as they arrive (one batch per decode plan); on a match the request is aborted and the
        result cut after the token that completes the stop string, finish reason `stop`."""
        import dataclasses

        tok = self._tokenizer
        loop = asyncio.get_running_loop()
        q: "asyncio.Queue" = asyncio.Queue()
        self.engine.submit(
            engine_rid, ids, multimodal=multimodal,
            on_tokens=lambda t, l, v: loop.call_soon_threadsafe(q.put_nowait, ("tokens", (t, l, v))),
            callback=lambda res: loop.call_soon_threadsafe(q.put_nowait, ("done", res)), **kw)
        toks: List[int] = []
        lps: List[float] = []
        vers: List[int] = []
        lookback = stop_strings.window(tok, stops)
        cut = None
        while True:
            kind, item = await q.get()
            if kind == "done":
                res = item
                break
            if cut is not None:
                continue  # tokens decoded between the match and the abort
            t, l, v = item
            first_new = len(toks)
            toks += t
            lps += l
            vers += v
            cut = stop_strings.first_cut(tok, toks, stops, first_new, lookback)
            if cut is not None:
                await self._offload(self.engine.abort, engine_rid)
        if cut is None:
            return res
        return dataclasses.replace(res, output_tokens=toks[:cut], output_logprobs=lps[:cut], output_versions=vers[:cut],
                                   finish_reason="stop")

    async def _generate_stream(self, req, rid, ids, multimodal, kw):
        """Server-sent events: one `data: {...}` event per batch of new tokens (`output_ids` = the tokens of this
        event, `meta_info.output_token_logprobs` / `output_versions` likewise, `completion_tokens` cumulative,
        `finish_reason` null), a last event that carries the finish reason and no tokens, then `data: [DONE]`.
        A client that goes away aborts its request."""
        import json

        loop = asyncio.get_running_loop()
        q: "asyncio.Queue" = asyncio.Queue()
        engine_rid = f"{rid}-{uuid.uuid4().hex[:8]}"
        self.engine.submit(
            engine_rid, ids, multimodal=multimodal,
            on_tokens=lambda t, l, v: loop.call_soon_threadsafe(q.put_nowait, ("tokens", (t, l, v))),
            callback=lambda res: loop.call_soon_threadsafe(q.put_nowait, ("done", res)), **kw)
        resp = web.StreamResponse(headers={"Content-Type": "text/event-stream", "Cache-Control": "no-cache"})
        await resp.prepare(req)
        sent = 0
        try:
            while True:
                kind, item = await q.get()
                if kind == "tokens":
                    toks, lps, vers = item
                    sent += len(toks)
                    ev = {"text": "", "output_ids": toks,
                          "meta_info": {"id": rid, "finish_reason": None, "prompt_tokens": len(ids),
                                        "completion_tokens": sent, "output_versions": vers,
                                        "output_token_logprobs": [(lp, t, None) for lp, t in zip(lps, toks)]}}
                else:
                    res = item
                    ev = {"text": "", "output_ids": [],
                          "meta_info": {"id": rid, "finish_reason": _finish(res.finish_reason),
                                        "prompt_tokens": res.prompt_len, "completion_tokens": len(res.output_tokens),
                                        "output_versions": [], "output_token_logprobs": [],
                                        "weight_version": self.engine.get_version(), "e2e_latency": res.latency,
                                        "ttft": res.ttft}}
                await resp.write(f"data: {json.dumps(ev)}\n\n".encode())
                if kind == "done":
                    break
            await resp.write(b"data: [DONE]\n\n")
            await resp.write_eof()
        except (ConnectionResetError, asyncio.CancelledError):
            await self._offload(self.engine.abort, engine_rid)
            raise
        return resp

    def _stop_to_ids(self):
        """str -> token ids with the server's tokenizer, or None when there is none to load."""
        if self._tokenizer is None:
            path = getattr(self.engine.config, "tokenizer_path", None) or self.engine.config.model_path
            self._tokenizer = False
            if path and os.path.isdir(str(path)):
                try:
                    from transformers import AutoTokenizer

                    self._tokenizer = AutoTokenizer.from_pretrained(path)
                except Exception as e:  # no tokenizer files next to the weights
                    logger.warning(f"no tokenizer at {path} ({e!r}): stop strings will be refused")
        if not self._tokenizer:
            return None
        return lambda text: list(self._tokenizer.encode(text, add_special_tokens=False))

    def _process_images(self, image_data: List[str]) -> Dict[str, torch.Tensor]:
        """base64 images -> patch tensor + grids with the checkpoint's own image processor (the request's
        `input_ids` already hold one placeholder token per merged patch, as the HF processor emits them)."""
        if self.engine.model_config.vision is None:
            raise ValueError("this server hosts a text-only model")
        if self._image_processor is None:
            from transformers import AutoProcessor

            path = getattr(self.engine.config, "tokenizer_path", None) or self.engine.config.model_path
            self._image_processor = AutoProcessor.from_pretrained(path).image_processor
        from ..utils.image import base642image

        out = self._image_processor(images=[base642image(x).convert("RGB") for x in image_data], return_tensors="pt")
        return dict(pixel_values=out["pixel_values"], image_grid_thw=out["image_grid_thw"])

    async def _offload(self, fn, *a, **k):
        return await asyncio.get_running_loop().run_in_executor(None, lambda: fn(*a, **k))

    async def pause(self, req):
        await self._offload(self.engine.pause)
        return web.json_response({"success": True})

    async def resume(self, req):
        await self._offload(self.engine.resume)
        return web.json_response({"success": True})

    async def flush_cache(self, req):
        await self._offload(self.engine.flush_cache)
        return web.json_response({"success": True})

    async def abort_request(self, req):
        body = await req.json()
        n = await self._offload(self.engine.abort_all) if not body.get("rid") else 0
        return web.json_response({"success": True, "aborted": n})

    async def set_version(self, req):
        body = await req.json()
        await self._offload(self.engine.set_version, int(body["version"]))
        return web.json_response({"success": True})

    async def update_from_disk(self, req):
        body = await req.json()
        path = body["model_path"]
        abort = bool(body.get("abort_all_requests", body.get("abort_all_request", True)))
        t0 = time.perf_counter()
        try:
            await self._offload(self.engine.load_weights_from_disk, path, body.get("version"), abort)
        except Exception as e:
            return web.json_response({"success": False, "message": repr(e)})
        return web.json_response({"success": True, "message": f"loaded in {time.perf_counter() - t0:.2f}s",
                                  "num_paused_requests": 0})

    async def init_update_group(self, req):
        body = await req.json()
        try:
            await self._offload(self.engine.init_weight_update_group, body)
        except Exception as e:
            return web.json_response({"success": False, "message": repr(e)})
        return web.json_response({"success": True, "message": "Succeeded to initialize custom process group."})

    async def update_from_distributed(self, req):
        body = await req.json()
        abort = bool(body.get("abort_all_requests", True))
        try:
            await self._offload(self.engine.update_weights_from_distributed, body["names"], body["dtypes"],
                                body["shapes"], body.get("version"), abort, body.get("group_name"), body.get("groups"))
        except Exception as e:
            return web.json_response({"success": False, "message": repr(e)})
        return web.json_response({"success": True, "message": "Succeeded to update parameter online."})


    async def update_from_ipc(self, req):
        body = await req.json()
        t0 = time.perf_counter()
        try:
            info = await self._offload(self.engine.update_weights_from_ipc, body.get("buffer"), body.get("layout", ""),
                                       int(body.get("used_bytes", 0)), body.get("version"),
                                       bool(body.get("abort_all_requests", True)), body.get("buffers"),
                                       body.get("src"), int(body.get("replica", 0)))
        except Exception as e:
            return web.json_response({"success": False, "message": repr(e)})
        return web.json_response({"success": True, "message": f"pulled in {time.perf_counter() - t0:.3f}s", **info})


def _add_config_args(parser: argparse.ArgumentParser):
    hints = typing.get_type_hints(InferenceServerConfig)
    for f in dataclasses.fields(InferenceServerConfig):
        name = "--" + f.name.replace("_", "-")
        tp = hints[f.name]
        origin = typing.get_origin(tp)
        args = [a for a in typing.get_args(tp) if a is not type(None)]
        base = args[0] if origin is typing.Union and args else tp
        if base is bool:
            parser.add_argument(name, action="store_true", default=f.default)
        elif typing.get_origin(base) in (list, typing.List):
            parser.add_argument(name, nargs="*", type=int, default=None)
        else:
            parser.add_argument(name, type=base if base in (int, float, str) else str, default=f.default)


def parse_server_args(argv: Optional[List[str]] = None):
    p = argparse.ArgumentParser("native rollout server")
    p.add_argument("--host", default="127.0.0.1")
    p.add_argument("--port", type=int, default=30000)
    p.add_argument("--tp-size", type=int, default=1)
    p.add_argument("--base-gpu-id", type=int, default=0)
    p.add_argument("--nnodes", type=int, default=1)
    p.add_argument("--node-rank", type=int, default=0)
    p.add_argument("--dist-init-addr", default=None)
    p.add_argument("--tokenizer-path", default=None)
    p.add_argument("--device", default=None)
    p.add_argument("--kv-cache-gb", type=float, default=None)
    p.add_argument("--num-pages", type=int, default=None)
    _add_config_args(p)
    args, _unknown = p.parse_known_args(argv)
    cfg_kwargs = {f.name: getattr(args, f.name) for f in dataclasses.fields(InferenceServerConfig)}
    return args, InferenceServerConfig(**cfg_kwargs)


def build_engine(args, cfg: InferenceServerConfig) -> NativeInferenceEngine:
    if args.device:
        dev = torch.device(args.device)
    elif torch.cuda.is_available():
        dev = torch.device("cuda", args.base_gpu_id if args.base_gpu_id < torch.cuda.device_count() else 0)
        torch.cuda.set_device(dev)
    else:
        dev = torch.device("cpu")
    if args.tp_size > 1:
        raise RuntimeError("tensor-parallel serving is started by main() through inference.tp_worker")
    eng = NativeInferenceEngine(cfg, device=dev, num_pages=args.num_pages, kv_cache_gb=args.kv_cache_gb)
    eng.initialize(load_weights=True)
    return eng


def main(argv: Optional[List[str]] = None):
    args, cfg = parse_server_args(argv)
    if args.tp_size > 1:
        from . import tp_worker

        engine = tp_worker.launch(args, cfg, sys.argv[1:] if argv is None else list(argv))
        if engine is None:  # follower rank: served its leader until shutdown
            return
    else:
        engine = build_engine(args, cfg)
    server = InferenceServer(engine)
    logger.info(f"serving on {args.host}:{args.port}")
    web.run_app(server.app, host=args.host, port=args.port, print=None, access_log=None)


if __name__ == "__main__":
    main()
