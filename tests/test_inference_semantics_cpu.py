"""Serving semantics of the native rollout engine on the CPU platform (tiny random model): frequency / presence
penalties, stop sequences, chunked prefill, early exit inside multi-step plans, pipelined decode plans, and the
request validation of the HTTP front end. The sampling fields are those of the reference's
``GenerationHyperparameters`` (``areal/api/cli_args.py:81-133``) as its client sends them
(``areal/engine/sglang_remote.py:121-143``)."""
import asyncio
import json
import socket
import threading

import pytest
import requests
import torch

from areal_vllm_amd.api.cli_args import InferenceServerConfig
from areal_vllm_amd.inference.engine import NativeInferenceEngine


def _engine(**kw):
    cfg = InferenceServerConfig(model_path="tiny", dtype="float32", context_length=256,
                                kv_cache_dtype=kw.pop("kv_cache_dtype", "auto"),
                                max_running_requests=kw.pop("max_running", 8),
                                num_continuous_decode_steps=kw.pop("decode_steps", 4),
                                max_prefill_tokens=kw.pop("max_prefill_tokens", 4096))
    return NativeInferenceEngine(cfg, device=torch.device("cpu"), num_pages=kw.pop("num_pages", 128)).initialize(
        load_weights=False)


@pytest.fixture(scope="module")
def eng():
    e = _engine()
    yield e
    e.destroy()


PROMPT = [5, 9, 17, 33, 64, 7, 21]


def _greedy(e, prompt=PROMPT, n=24, **kw):
    return e.generate_blocking([prompt], max_new_tokens=n, greedy=True, **kw)[0]


def test_frequency_penalty_suppresses_repeats(eng):
    base = _greedy(eng, n=48)
    assert len(set(base.output_tokens)) < len(base.output_tokens), "the tiny model should repeat itself when greedy"
    pen = _greedy(eng, n=48, frequency_penalty=1000.0)
    assert len(set(pen.output_tokens)) == len(pen.output_tokens) == 48  # a token once drawn is never the arg-max again
    # up to the first repeat of the unpenalised run nothing has a count yet: same tokens, same log-probs
    first_repeat = next(i for i, t in enumerate(base.output_tokens) if t in base.output_tokens[:i])
    assert pen.output_tokens[:first_repeat] == base.output_tokens[:first_repeat]
    assert pen.output_logprobs[0] == pytest.approx(base.output_logprobs[0], abs=1e-5)
    # presence penalty: same effect on the arg-max when large
    pres = _greedy(eng, n=48, presence_penalty=1000.0)
    assert len(set(pres.output_tokens)) == 48
    # the count rows went back to the pool and are cleared for their next owner
    assert len(eng._free_penalty_slots) == 8
    again = _greedy(eng, n=48, frequency_penalty=1000.0)
    assert again.output_tokens == pen.output_tokens


def test_penalty_row_reused_while_the_batch_is_full_starts_from_zero():
    """A penalised request that arrives while the batch is at max_running_requests gets its count row attached to a
    CONTINUATION plan (no prefill is possible): the row must be zeroed there too, or the request samples with its
    previous owner's counts."""
    e = _engine(max_running=2, decode_steps=2)
    try:
        first = _greedy(e, n=40, frequency_penalty=1000.0)  # leaves counts behind in its row
        started = threading.Event()
        results = {}
        done = threading.Event()

        def cb(name):
            def f(res):
                results[name] = res
                if len(results) == 3:
                    done.set()
            return f

        for i in range(2):  # fill the batch with unpenalised requests
            e.submit(f"fill-{i}", [11 + i, 3, 90, 41], max_new_tokens=120, greedy=True, callback=cb(f"fill-{i}"),
                     on_tokens=lambda *a: started.set())
        assert started.wait(60)  # decode plans are in flight now, every further plan is a continuation
        before = e.stats["pipelined_plans"]
        e.submit("late", PROMPT, max_new_tokens=40, greedy=True, frequency_penalty=1000.0, callback=cb("late"))
        assert done.wait(120)
        assert e.stats["pipelined_plans"] > before
        assert results["late"].output_tokens == first.output_tokens
    finally:
        e.destroy()


def test_mild_penalty_shifts_the_reported_logprob(eng):
    base = _greedy(eng, n=32)
    first_repeat = next(i for i, t in enumerate(base.output_tokens) if t in base.output_tokens[:i])
    mild = _greedy(eng, n=first_repeat + 1, frequency_penalty=0.05)
    if mild.output_tokens == base.output_tokens[:first_repeat + 1]:  # still the arg-max, but less likely than before
        assert mild.output_logprobs[first_repeat] < base.output_logprobs[first_repeat] - 1e-4


def test_stop_sequence_ends_generation(eng):
    base = _greedy(eng, n=24)
    o = base.output_tokens
    seq = o[5:7]
    j = next(i for i in range(1, len(o)) if o[i - 1:i + 1] == seq)  # first place the pair is complete
    got = _greedy(eng, n=24, stop_sequences=[[1, 2, 3], seq])
    assert got.finish_reason == "stop" and got.output_tokens == o[:j + 1]
    # min_new_tokens postpones it to the next occurrence (or to the length limit)
    late = _greedy(eng, n=24, stop_sequences=[seq], min_new_tokens=j + 2)
    assert len(late.output_tokens) > j + 1 and late.output_tokens[:j + 1] == o[:j + 1]


def test_rows_that_finish_inside_a_plan_do_not_disturb_the_others(eng):
    """Requests with different budgets share multi-step plans; the short ones retire on the device mid-plan."""
    prompts = [[3 + i, 40 + i, 77, 12 + 2 * i] for i in range(6)]
    solo = [eng.generate_blocking([p], max_new_tokens=20, greedy=True)[0].output_tokens for p in prompts]
    budgets = [1, 2, 3, 5, 9, 20]
    out = [None] * 6
    done = threading.Event()
    left = [6]

    def cb(i):
        def f(res):
            out[i] = res
            left[0] -= 1
            if left[0] == 0:
                done.set()
        return f

    for i, p in enumerate(prompts):
        eng.submit(f"mix-{i}", p, max_new_tokens=budgets[i], greedy=True, callback=cb(i))
    assert done.wait(120)
    for i in range(6):
        assert out[i].finish_reason == "length" and out[i].output_tokens == solo[i][:budgets[i]]
    m = eng.metrics()
    assert m["num_running_reqs"] == 0 and m["num_used_pages"] == 0  # nothing leaked


def test_pipelined_decode_matches_plan_by_plan_decode(eng):
    prompts = [[3 + i, 50 + i, 9, 100 - i, 8] for i in range(5)]
    budgets = [30, 30, 30, 7, 30]  # one row dies early: later plans carry it as a dead row
    runs = {}
    for mode in (False, True):
        eng.pipeline_decode = mode
        eng.flush_cache()
        before = eng.stats["pipelined_plans"]
        res = [None] * 5
        done = threading.Event()
        left = [5]

        def cb(i, res=res, left=left, done=done):
            def f(r):
                res[i] = r
                left[0] -= 1
                if left[0] == 0:
                    done.set()
            return f

        eng.pause()  # queue all five so that they share their plans
        for i, p in enumerate(prompts):
            eng.submit(f"pipe-{mode}-{i}", p, max_new_tokens=budgets[i], greedy=True, callback=cb(i))
        eng.resume()
        assert done.wait(120)
        runs[mode] = ([r.output_tokens for r in res], [r.output_logprobs for r in res],
                      eng.stats["pipelined_plans"] - before)
    eng.pipeline_decode = True
    assert runs[False][2] == 0 and runs[True][2] >= 3
    assert runs[True][0] == runs[False][0]
    for a, b in zip(runs[True][1], runs[False][1]):
        assert a == pytest.approx(b, abs=1e-5)
    m = eng.metrics()
    assert m["num_running_reqs"] == 0 and m["num_used_pages"] == 0


def test_abort_during_pipelined_decode_returns_partial_output(eng):
    got = []
    done = threading.Event()
    eng.submit("abort-me", PROMPT, max_new_tokens=200, greedy=True, callback=lambda r: (got.append(r), done.set()))
    for _ in range(2000):
        if eng.stats["decode_steps"] > 0 and eng.metrics()["num_running_reqs"] == 1:
            break
        threading.Event().wait(0.002)
    n = eng.abort_all()
    assert done.wait(60) and n == 1
    r = got[0]
    assert r.finish_reason == "abort" and 0 < len(r.output_tokens) < 200
    assert r.output_tokens == _greedy(eng, n=len(r.output_tokens)).output_tokens
    assert eng.metrics()["num_used_pages"] == 0


def test_chunked_prefill_matches_whole_prompt_prefill():
    prompt = [(7 * i + 3) % 120 + 2 for i in range(90)]
    whole, chunked = _engine(), _engine(max_prefill_tokens=16)
    try:
        a = whole.generate_blocking([prompt, PROMPT], max_new_tokens=12, greedy=True)
        b = chunked.generate_blocking([prompt, PROMPT], max_new_tokens=12, greedy=True)
        assert chunked.stats["prefill_steps"] >= 6 > whole.stats["prefill_steps"]
        for x, y in zip(a, b):
            assert x.output_tokens == y.output_tokens
            assert x.output_logprobs == pytest.approx(y.output_logprobs, abs=1e-4)
    finally:
        whole.destroy()
        chunked.destroy()


def test_fp8_kv_cache_serves_the_same_requests(eng):
    """`kv_cache_dtype=fp8_e4m3`: keys and values rest as e4m3 bytes (half the cache bytes per token), prefill over
    a cached prefix and decode read them widened. The behaviour log-probs stay close to the full-precision cache's,
    a prefix hit reproduces the miss exactly, and both engines hold the same number of pages in half the memory."""
    e8 = _engine(kv_cache_dtype="fp8_e4m3")
    try:
        k8, kf = e8.runner.kv[0][0], eng.runner.kv[0][0]
        assert k8.dtype == torch.float8_e4m3fn and k8.shape == kf.shape
        assert e8.runner.kv_cache_bytes() * kf.element_size() == eng.runner.kv_cache_bytes()
        a = _greedy(e8, n=24)
        b = _greedy(eng, n=24)
        same = next((i for i, (x, y) in enumerate(zip(a.output_tokens, b.output_tokens)) if x != y), 24)
        assert same >= 4  # the arg-max of a random tiny model may flip on a near tie later on
        assert a.output_logprobs[:same] == pytest.approx(b.output_logprobs[:same], abs=0.05)
        again = _greedy(e8, n=24)  # the prompt's pages are found in the prefix cache this time
        assert again.output_tokens == a.output_tokens
        assert again.output_logprobs == pytest.approx(a.output_logprobs, abs=1e-4)
        long = list(range(3, 3 + 70))  # several pages, extended by a second request that shares them
        first = e8.generate_blocking([long], max_new_tokens=4, greedy=True)[0]
        ext = e8.generate_blocking([long + first.output_tokens[:2]], max_new_tokens=2, greedy=True)[0]
        assert ext.output_tokens == first.output_tokens[2:4]
    finally:
        e8.destroy()
    with pytest.raises(ValueError):
        InferenceServerConfig(model_path="tiny", kv_cache_dtype="fp8_e5m2").reject_unsupported()


# ------------------------------------------------------------------------------------ HTTP front end
@pytest.fixture(scope="module")
def server(eng):
    from aiohttp import web

    from areal_vllm_amd.inference.server import InferenceServer

    srv = InferenceServer(eng)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    loop = asyncio.new_event_loop()
    started = threading.Event()

    def run():
        asyncio.set_event_loop(loop)
        runner = web.AppRunner(srv.app)
        loop.run_until_complete(runner.setup())
        loop.run_until_complete(web.TCPSite(runner, "127.0.0.1", port).start())
        started.set()
        loop.run_forever()

    threading.Thread(target=run, daemon=True).start()
    assert started.wait(30)
    yield f"http://127.0.0.1:{port}"
    loop.call_soon_threadsafe(loop.stop)


def _gen(url, sampling, **extra):
    return requests.post(f"{url}/generate", json={"input_ids": PROMPT, "sampling_params": sampling, **extra}, timeout=60)


def test_generate_rejects_what_it_cannot_honour(server):
    bad = _gen(server, {"max_new_tokens": 4, "repetition_penalty": 1.1, "beam_width": 4})
    assert bad.status_code == 400 and "repetition_penalty" in bad.json()["error"] and "beam_width" in bad.json()["error"]
    assert _gen(server, {"max_new_tokens": 4, "n": 2}, stream=True).status_code == 400
    assert _gen(server, {"max_new_tokens": 4, "n": 0}).status_code == 400
    assert _gen(server, {"max_new_tokens": 4}, lora_path="x").status_code == 400
    no_tok = _gen(server, {"max_new_tokens": 4, "stop": ["</answer>"]})  # the preset has no tokenizer to map it with
    assert no_tok.status_code == 400 and "stop_sequences" in no_tok.json()["error"]


def test_generate_returns_n_samples_of_one_prompt(server, eng):
    """`n` samples per request (the list form of the reference servers' answer): sampled ones differ from each other,
    greedy ones equal the single-sample answer, and the prompt is prefilled once (the others hit the prefix cache)."""
    base = _gen(server, {"max_new_tokens": 12, "temperature": 0.0}).json()
    out = _gen(server, {"max_new_tokens": 12, "temperature": 0.0, "n": 3}, rid="grp").json()
    assert isinstance(out, list) and len(out) == 3
    assert [o["meta_info"]["id"] for o in out] == ["grp-0", "grp-1", "grp-2"]
    assert all(o["output_ids"] == base["output_ids"] for o in out)
    hot = _gen(server, {"max_new_tokens": 12, "temperature": 1.5, "n": 4}).json()
    assert len({tuple(o["output_ids"]) for o in hot}) > 1
    assert all(len(o["meta_info"]["output_token_logprobs"]) == len(o["output_ids"]) for o in hot)


def test_generate_honours_penalties_and_stop_sequences_over_http(server, eng):
    base = _gen(server, {"max_new_tokens": 24, "temperature": 0.0}).json()["output_ids"]
    assert base == _greedy(eng, n=24).output_tokens
    pen = _gen(server, {"max_new_tokens": 24, "temperature": 0.0, "frequency_penalty": 1000.0}).json()["output_ids"]
    assert len(set(pen)) == 24
    seq = base[3:5]
    r = _gen(server, {"max_new_tokens": 24, "temperature": 0.0, "stop_sequences": [seq]}).json()
    j = next(i for i in range(1, len(base)) if base[i - 1:i + 1] == seq)
    assert r["output_ids"] == base[:j + 1] and r["meta_info"]["finish_reason"]["type"] == "stop"
    # ignore_eos drops the stop ids
    stop_at = base[2]
    first = base.index(stop_at)
    assert _gen(server, {"max_new_tokens": 8, "temperature": 0.0,
                         "stop_token_ids": [stop_at]}).json()["output_ids"] == base[:first + 1]
    assert _gen(server, {"max_new_tokens": 8, "temperature": 0.0, "stop_token_ids": [stop_at],
                         "ignore_eos": True}).json()["output_ids"] == base[:8]


def test_parse_sampling_params_maps_stop_strings_through_the_tokenizer():
    from areal_vllm_amd.inference.server import BadRequest, parse_sampling_params

    kw = parse_sampling_params({"stop": ["ab", "c"], "stop_sequences": [[9, 9]], "top_k": -1, "temperature": 0.5},
                               stop_to_ids=lambda text: [ord(ch) for ch in text])
    assert kw["stop_sequences"] == [[9, 9], [97, 98], [99]] and kw["top_k"] == 1 << 30 and not kw["greedy"]
    with pytest.raises(BadRequest):
        parse_sampling_params({"stop": "x"}, stop_to_ids=None)


def test_generate_streams_tokens_as_server_sent_events(server, eng):
    """stream=true: tokens arrive in several events while the request runs, their concatenation (tokens,
    log-probs, versions) is the non-streamed answer, the last event carries the finish reason."""
    want = _gen(server, {"max_new_tokens": 40, "temperature": 0.0}).json()
    events = []
    with requests.post(f"{server}/generate", json={"input_ids": PROMPT, "stream": True, "sampling_params":
                                                    {"max_new_tokens": 40, "temperature": 0.0}},
                       stream=True, timeout=60) as r:
        assert r.status_code == 200 and r.headers["Content-Type"].startswith("text/event-stream")
        for line in r.iter_lines():
            if line.startswith(b"data: "):
                events.append(line[6:].decode())
    assert events[-1] == "[DONE]"
    evs = [json.loads(e) for e in events[:-1]]
    chunks = [e for e in evs if e["output_ids"]]
    assert len(chunks) >= 2, "the whole answer arrived in one event"
    assert sum((e["output_ids"] for e in chunks), []) == want["output_ids"]
    lps = [lp for e in chunks for lp, _t, _ in e["meta_info"]["output_token_logprobs"]]
    assert lps == pytest.approx([lp for lp, _t, _ in want["meta_info"]["output_token_logprobs"]], abs=1e-4)
    assert [e["meta_info"]["completion_tokens"] for e in chunks] == list(
        __import__("itertools").accumulate(len(e["output_ids"]) for e in chunks))
    assert all(e["meta_info"]["finish_reason"] is None for e in chunks)
    last = evs[-1]
    assert last["output_ids"] == [] and last["meta_info"]["finish_reason"]["type"] == "length"
    assert last["meta_info"]["completion_tokens"] == 40
    # engine-level callback: same stream without HTTP, a stop token ends it
    got, done = [], threading.Event()
    stop_at = want["output_ids"][5]
    eng.submit("stream-direct", PROMPT, max_new_tokens=40, greedy=True, stop_token_ids=[stop_at],
               on_tokens=lambda t, l, v: got.extend(t), callback=lambda res: (got.append(res), done.set()))
    assert done.wait(60)
    res = got.pop()
    assert got == res.output_tokens == want["output_ids"][:want["output_ids"].index(stop_at) + 1]


class _LetterTokenizer:
    """26 letters, every letter spelled by many token ids (id % 26): what a real vocabulary does with merged and
    split pieces, in the small."""

    def encode(self, text, add_special_tokens=False):
        return [ord(c) - 97 for c in text]

    def decode(self, ids, **kw):
        return "".join(chr(97 + int(i) % 26) for i in ids)


def test_stop_strings_end_generation_where_the_decoded_text_contains_them(eng):
    """The engine matches a stop string as the token ids its tokenizer gives it in isolation; the server also watches
    the decoded text, so a stop string the model spells with other tokens still ends the request there."""
    from aiohttp import web

    from areal_vllm_amd.inference.server import InferenceServer

    srv = InferenceServer(eng)
    srv._tokenizer = _LetterTokenizer()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    loop = asyncio.new_event_loop()
    started = threading.Event()

    def run():
        asyncio.set_event_loop(loop)
        runner = web.AppRunner(srv.app)
        loop.run_until_complete(runner.setup())
        loop.run_until_complete(web.TCPSite(runner, "127.0.0.1", port).start())
        started.set()
        loop.run_forever()

    threading.Thread(target=run, daemon=True).start()
    assert started.wait(30)
    url = f"http://127.0.0.1:{port}"
    try:
        base = _greedy(eng, n=24).output_tokens
        text = srv._tokenizer.decode(base)
        # a two-letter stop string whose first occurrence is spelled with non-canonical ids
        j = next(i for i in range(4, 22) if (base[i] >= 26 or base[i + 1] >= 26) and text.find(text[i:i + 2]) == i)
        stop = text[j:j + 2]
        assert srv._tokenizer.encode(stop) != base[j:j + 2]
        r = _gen(url, {"max_new_tokens": 24, "temperature": 0.0, "stop": [stop]}).json()
        assert r["output_ids"] == base[:j + 2] and r["meta_info"]["finish_reason"]["type"] == "stop"
        assert len(r["meta_info"]["output_token_logprobs"]) == j + 2 and len(r["meta_info"]["output_versions"]) == j + 2
        # a stop string that never shows up leaves the request alone
        absent = next(a + b for a in "abcdefghijklmnopqrstuvwxyz" for b in "abcdefghijklmnopqrstuvwxyz"
                      if a + b not in text)
        r = _gen(url, {"max_new_tokens": 24, "temperature": 0.0, "stop": [absent]}).json()
        assert r["output_ids"] == base and r["meta_info"]["finish_reason"]["type"] == "length"
    finally:
        loop.call_soon_threadsafe(loop.stop)


def test_client_loop_cuts_at_a_stop_string_the_server_did_not_see(eng):
    """`agenerate` of the in-process engine (the same loop serves the HTTP client): the stop string's canonical token
    ids never occur, its text does."""
    from areal_vllm_amd.api.cli_args import GenerationHyperparameters, InferenceEngineConfig
    from areal_vllm_amd.api.io_struct import ModelRequest
    from areal_vllm_amd.engine.local_engine import LocalInferenceEngine

    tok = _LetterTokenizer()
    base = _greedy(eng, n=24).output_tokens
    text = tok.decode(base)
    j = next(i for i in range(4, 22) if (base[i] >= 26 or base[i + 1] >= 26) and text.find(text[i:i + 2]) == i)
    le = LocalInferenceEngine(InferenceEngineConfig(max_concurrent_rollouts=4, consumer_batch_size=1), eng)
    le.initialize(train_data_parallel_size=1)
    try:
        g = GenerationHyperparameters(n_samples=1, max_new_tokens=24, greedy=True, stop=[text[j:j + 2]])
        resp = asyncio.run(le.agenerate(ModelRequest(rid="s", input_ids=PROMPT, gconfig=g, tokenizer=tok)))
        assert resp.output_tokens == base[:j + 2] and resp.stop_reason == "stop"
        assert len(resp.output_logprobs) == j + 2 and len(resp.output_versions) == j + 2
    finally:
        le.destroy()
