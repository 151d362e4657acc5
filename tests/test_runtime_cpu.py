"""C++ host runtime: continuous-batching scheduler, paged block manager, packing helpers."""
import os
import numpy as np
import pytest

from areal_vllm_amd._C import _runtime as rt


def _sched(num_pages=32, max_running=4, decode_steps=1, prefix=True, ctx=256, max_prefill=256):
    sc = rt.SchedulerConfig()
    sc.num_pages, sc.max_running_requests, sc.max_prefill_tokens = num_pages, max_running, max_prefill
    sc.context_length, sc.decode_steps, sc.enable_prefix_cache = ctx, decode_steps, prefix
    return rt.Scheduler(sc)


def _drive(s, token_fn, max_iters=10000):
    """Run the scheduler to completion with a fake model: token_fn(req_id, position) -> token."""
    done = {}
    for _ in range(max_iters):
        plan = s.schedule()
        if plan["kind"] == 0:
            for f in s.take_finished():
                done[f["rid"]] = f
            if not s.has_work():
                break
            continue
        if plan["kind"] == 1:
            n = len(plan["req_ids"])
            cu = plan["cu_q"]
            toks = [token_fn(int(plan["req_ids"][i]), int(plan["positions"][cu[i + 1] - 1]) + 1) for i in range(n)]
            s.process_prefill(toks, [-1.0] * n)
        else:
            n, k = len(plan["req_ids"]), plan["n_steps"]
            toks = [[token_fn(int(plan["req_ids"][i]), int(plan["positions"][i]) + 1 + j) for i in range(n)]
                    for j in range(k)]
            s.process_decode(np.array(toks, dtype=np.int32), np.full((k, n), -1.0, dtype=np.float32))
        for f in s.take_finished():
            done[f["rid"]] = f
    return done


def test_generate_until_length_and_stop_token():
    s = _sched()
    s.add_request("len", [1, 2, 3], 5, 0, 1.0, 1.0, 1 << 30, [])
    s.add_request("stop", [4, 5, 6, 7], 50, 0, 1.0, 1.0, 1 << 30, [99])
    done = _drive(s, lambda rid, pos: 99 if pos == 8 else 10 + pos)
    assert done["len"]["finish_reason"] == "length" and len(done["len"]["output_tokens"]) == 5
    assert done["stop"]["finish_reason"] == "stop"
    assert done["stop"]["output_tokens"][-1] == 99 and len(done["stop"]["output_tokens"]) == 5
    assert s.num_running() == 0 and s.num_waiting() == 0


def test_min_new_tokens_suppresses_stop():
    s = _sched()
    s.add_request("a", [1, 2], 10, 4, 1.0, 1.0, 1 << 30, [99])
    done = _drive(s, lambda rid, pos: 99)
    assert len(done["a"]["output_tokens"]) >= 4 and done["a"]["finish_reason"] == "stop"


def test_multi_step_decode_truncates_after_stop():
    s = _sched(decode_steps=4)
    s.add_request("a", [1, 2, 3], 64, 0, 1.0, 1.0, 1 << 30, [99])
    done = _drive(s, lambda rid, pos: 99 if pos == 6 else 5)
    out = done["a"]["output_tokens"]
    assert out[-1] == 99 and out.count(99) == 1 and len(out) == 4


def test_continuous_batching_respects_max_running():
    s = _sched(max_running=2)
    for i in range(5):
        s.add_request(f"r{i}", [1, 2, 3, 4], 3, 0, 1.0, 1.0, 1 << 30, [])
    peak = 0
    done = {}
    for _ in range(1000):
        plan = s.schedule()
        peak = max(peak, s.num_running())
        if plan["kind"] == 0:
            if not s.has_work():
                break
            continue
        n = len(plan["req_ids"])
        if plan["kind"] == 1:
            s.process_prefill([7] * n, [-1.0] * n)
        else:
            s.process_decode(np.full((plan["n_steps"], n), 7, dtype=np.int32),
                             np.full((plan["n_steps"], n), -1.0, dtype=np.float32))
        for f in s.take_finished():
            done[f["rid"]] = f
    assert len(done) == 5 and peak <= 2


def test_pages_are_returned_and_prefix_cache_hits():
    s = _sched(num_pages=32, prefix=True)
    free0 = s.num_free_pages()
    prompt = list(range(2, 2 + 70))  # two full pages + 6 tokens
    s.add_request("a", prompt, 2, 0, 1.0, 1.0, 1 << 30, [])
    _drive(s, lambda rid, pos: 5)
    assert s.total_cached_tokens() == 0
    s.add_request("b", prompt, 2, 0, 1.0, 1.0, 1 << 30, [])
    plan = s.schedule()
    # the shared two-page prefix is not recomputed
    assert plan["kind"] == 1 and plan["has_prefix"] and len(plan["input_ids"]) == 70 - 64
    s.process_prefill([5], [-1.0])
    _drive(s, lambda rid, pos: 5)
    assert s.total_cached_tokens() == 64
    s.flush_cache()
    assert s.num_free_pages() == free0


def test_abort_all_returns_partial_output_with_versions():
    s = _sched()
    s.add_request("a", [1, 2, 3], 100, 0, 1.0, 1.0, 1 << 30, [])
    plan = s.schedule()
    s.process_prefill([7], [-0.1])
    plan = s.schedule()
    s.process_decode(np.array([[8]], dtype=np.int32), np.array([[-0.2]], dtype=np.float32))
    s.set_version(3)
    plan = s.schedule()
    s.process_decode(np.array([[9]], dtype=np.int32), np.array([[-0.3]], dtype=np.float32))
    assert s.abort_all() == 1
    f = s.take_finished()[0]
    assert f["finish_reason"] == "abort" and list(f["output_tokens"]) == [7, 8, 9]
    assert list(f["output_versions"]) == [0, 0, 3]
    assert np.allclose(f["output_logprobs"], [-0.1, -0.2, -0.3])


def test_pause_stops_scheduling():
    s = _sched()
    s.add_request("a", [1, 2, 3], 4, 0, 1.0, 1.0, 1 << 30, [])
    s.pause()
    assert s.paused() and s.schedule()["kind"] == 0
    s.resume()
    assert s.schedule()["kind"] == 1


def test_preemption_when_cache_is_exhausted():
    # 7 usable pages of 32 tokens: three requests growing to 96+ tokens cannot all stay resident
    s = _sched(num_pages=8, max_running=3, ctx=200)
    for i in range(3):
        s.add_request(f"r{i}", [1 + i] * 30, 90, 0, 1.0, 1.0, 1 << 30, [])
    done = _drive(s, lambda rid, pos: 5)
    assert len(done) == 3 and all(len(d["output_tokens"]) == 90 for d in done.values())
    assert s.num_preemptions() > 0


def test_ffd_and_partition_helpers():
    groups = rt.ffd_allocate([5, 9, 3, 7, 2, 8], 10, 1)
    assert sorted(i for g in groups for i in g) == list(range(6))
    assert all(sum([5, 9, 3, 7, 2, 8][i] for i in g) <= 10 for g in groups)
    assert len(rt.ffd_allocate([1, 1, 1, 1], 100, 3)) >= 3
    bounds = rt.partition_balanced([4, 4, 4, 4, 8, 8], 3, 1)
    assert bounds[0] == 0 and bounds[-1] == 6 and len(bounds) == 4
    sums = [sum([4, 4, 4, 4, 8, 8][bounds[i]:bounds[i + 1]]) for i in range(3)]
    assert max(sums) <= 16
    merged = rt.merge_intervals(np.array([[0, 4], [4, 8], [10, 12], [11, 15]], dtype=np.int64))
    assert np.asarray(merged).reshape(-1, 2).tolist() == [[0, 8], [10, 15]]


def test_scheduler_stress_under_address_and_ub_sanitizers():
    """Thousands of random requests (shared prefixes, aborts, pauses, preemption on a tiny page pool)
    through the C++ scheduler built with -fsanitize=address,undefined: no memory error, no page leak."""
    import subprocess
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import build_ext

    exe = build_ext.build_stress("address,undefined", verbose=False)
    r = subprocess.run([str(exe), "6"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "stress ok" in r.stdout and "runtime error" not in r.stderr
    import re

    counts = {k: int(v) for v, k in re.findall(r"(\d+) (preemptions|partial prefill chunks|continuation plans|"
                                               r"deferred releases|stop-sequence finishes)", r.stdout)}
    # the pool was tight enough to exercise preemption, and every newer code path ran as well
    assert len(counts) == 5 and all(v > 0 for v in counts.values()), r.stdout


# ------------------------------------------------------------------ stop sequences / penalties / chunks
def _add(s, rid, prompt, max_new, **kw):
    return s.add_request(rid, prompt, max_new, kw.get("min_new", 0), 1.0, 1.0, 1 << 30, kw.get("stop_ids", []),
                         kw.get("stop_sequences", []), kw.get("freq", 0.0), kw.get("pres", 0.0), kw.get("slot", -1))


def test_stop_sequence_matches_tail_of_generated_tokens_across_plans():
    s = _sched(decode_steps=2)
    stream = [11, 12, 13, 14, 15, 16, 17, 18]
    _add(s, "a", [13, 14, 3], 50, stop_sequences=[[13, 14], [99]])  # the PROMPT ending in 13 14 does not count
    _add(s, "b", [1, 2, 3], 50, stop_sequences=[[14, 15, 16]])  # straddles a plan boundary (steps 4|5 and 6)
    _add(s, "c", [1, 2, 3], 50, stop_sequences=[[11]], min_new=3)  # too early: min_new_tokens not reached
    done = _drive(s, lambda rid, pos: stream[(pos - 3) % len(stream)])
    assert done["a"]["finish_reason"] == "stop" and done["a"]["output_tokens"] == [11, 12, 13, 14]
    assert done["b"]["finish_reason"] == "stop" and done["b"]["output_tokens"] == [11, 12, 13, 14, 15, 16]
    assert done["c"]["finish_reason"] == "stop" and done["c"]["output_tokens"] == stream + [11]


def test_prompt_longer_than_the_prefill_budget_is_prefilled_in_chunks():
    s = _sched(num_pages=64, max_prefill=32, ctx=512)
    long_prompt = list(range(100, 200))  # 100 tokens -> chunks of 32, 32, 32, 4
    _add(s, "long", long_prompt, 3)
    seen, samples = [], []
    for _ in range(4):
        plan = s.schedule()
        assert plan["kind"] == 1 and len(plan["input_ids"]) <= 32
        seen += plan["input_ids"].tolist()
        samples.append(int(plan["sample"][0]))
        assert plan["positions"].tolist() == list(range(len(seen) - len(plan["input_ids"]), len(seen)))
        assert int(plan["cu_k"][1]) == len(seen) == int(plan["chunk_end"][0])
        # later chunks attend to the earlier ones through the page table
        assert bool(plan["has_prefix"]) == (len(seen) > 32)
        if plan["has_prefix"]:  # the row's whole context so far is reachable through its page table
            assert plan["block_tables"].shape == (1, plan["max_pages"]) and plan["max_pages"] * 32 >= len(seen)
            assert [int(x) // 32 for x in plan["slot_mapping"]] == [
                int(plan["block_tables"][0, int(p) // 32]) for p in plan["positions"]]
        if len(seen) < 100:
            assert s.num_running() == 1 and not s.take_finished()
        s.process_prefill([7], [-0.5], plan["plan_id"])
    assert seen == long_prompt and samples == [0, 0, 0, 1]
    # a short prompt that arrives mid-way shares a plan with the next chunk, within the budget
    _add(s, "long2", list(range(300, 350)), 2)
    p1 = s.schedule()
    s.process_prefill([7], [-0.5], p1["plan_id"])
    _add(s, "short", [5, 6, 7], 2)
    p2 = s.schedule()
    assert p1["kind"] == p2["kind"] == 1 and len(p1["input_ids"]) == 32
    assert len(p2["req_ids"]) == 2 and p2["sample"].tolist() == [1, 1] and len(p2["input_ids"]) == 18 + 3
    s.process_prefill([7, 7], [-0.5, -0.5], p2["plan_id"])
    done = _drive(s, lambda rid, pos: 7)
    assert {k: len(v["output_tokens"]) for k, v in done.items()} == {"long": 3, "long2": 2, "short": 2}
    assert s.num_free_pages() == s.num_total_pages()


def test_continuation_plan_keeps_rows_and_defers_page_release():
    s = _sched(num_pages=64, decode_steps=2, prefix=False)
    _add(s, "a", [1, 2, 3], 40, stop_ids=[99])
    _add(s, "b", [4, 5, 6, 7], 40, stop_ids=[99])
    p = s.schedule()
    s.process_prefill([8, 8], [-1.0, -1.0], p["plan_id"])
    A = s.schedule()
    assert A["kind"] == 2 and not A["continuation"] and A["kill"].tolist() == [0, 0]
    assert A["remaining"].tolist() == [39, 39] and A["stop_ids"][:, 0].tolist() == [99, 99]
    B = s.schedule_continuation(A["plan_id"], 1)  # scheduled BEFORE A's tokens are known
    assert B["kind"] == 2 and B["continuation"] and B["req_ids"].tolist() == A["req_ids"].tolist()
    free = s.num_free_pages()
    # A's tokens: request "a" draws its stop token in the first step
    s.process_decode(np.array([[99, 8], [8, 8]]), np.full((2, 2), -1.0, np.float32), A["plan_id"], True)
    fin = s.take_finished()
    assert [f["rid"] for f in fin] == ["a"] and fin[0]["output_tokens"] == [8, 99]
    # delivered at once, but B (in flight) may still write K/V for that row: its pages are not free yet
    assert s.num_deferred_pages() > 0 and s.num_free_pages() == free
    C = s.schedule_continuation(B["plan_id"], 1)
    assert C["kind"] == 2 and C["kill"].tolist() == [1, 0] and C["req_ids"].tolist() == [-1, int(A["req_ids"][1])]
    assert not C["block_tables"][0].any()
    s.process_decode(np.full((2, 2), 8), np.full((2, 2), -1.0, np.float32), B["plan_id"], True)
    assert s.num_deferred_pages() == 0 and s.num_free_pages() > free  # B is done: now they are
    # a continuation is refused when the batch would fit the next smaller bucket, or a prefill is due
    assert s.schedule_continuation(C["plan_id"], 2)["kind"] == 0
    _add(s, "c", [9, 9, 9], 4)
    assert s.prefill_possible() and s.schedule_continuation(C["plan_id"], 1)["kind"] == 0
    s.process_decode(np.full((2, 2), 8), np.full((2, 2), -1.0, np.float32), C["plan_id"], False)
    done = _drive(s, lambda rid, pos: 8)
    assert len(done["b"]["output_tokens"]) == 40 and s.num_free_pages() == s.num_total_pages()


def test_prefix_cache_hit_is_checked_against_the_stored_tokens():
    s = _sched(num_pages=32, prefix=True)
    prompt = list(range(2, 2 + 70))
    _add(s, "a", prompt, 2)
    _drive(s, lambda rid, pos: 5)
    _add(s, "b", prompt, 2)
    plan = s.schedule()
    assert plan["has_prefix"] and len(plan["input_ids"]) == 6  # both full pages come from the cache
    first_page = int(plan["block_tables"][0, 0])
    s.process_prefill([5], [-1.0], plan["plan_id"])
    _drive(s, lambda rid, pos: 5)
    # same hash key, different tokens behind it (what a 64-bit collision would look like): no hit
    s.corrupt_cached_page_for_test(first_page)
    _add(s, "c", prompt, 2)
    plan = s.schedule()
    assert not plan["has_prefix"] and len(plan["input_ids"]) == 70 and s.num_prefix_collisions() >= 1
    s.process_prefill([5], [-1.0], plan["plan_id"])
    _drive(s, lambda rid, pos: 5)
    assert s.num_free_pages() == s.num_total_pages()


def test_penalties_travel_with_a_count_table_slot():
    s = _sched()
    with pytest.raises(RuntimeError):
        _add(s, "x", [1, 2, 3], 4, freq=0.5)  # a penalty without a slot
    _add(s, "a", [1, 2, 3], 2, freq=0.5, pres=0.25, slot=3)
    _add(s, "b", [1, 2, 3], 2)
    p = s.schedule()
    assert p["penalty_slot"].tolist() == [3, -1] and p["frequency_penalty"].tolist() == [0.5, 0.0]
    assert p["presence_penalty"].tolist() == [0.25, 0.0]
    s.process_prefill([5, 5], [-1.0, -1.0], p["plan_id"])
    d = s.schedule()
    assert d["kind"] == 2 and d["penalty_slot"].tolist() == [3, -1]
    s.process_decode(np.full((1, 2), 5), np.full((1, 2), -1.0, np.float32), d["plan_id"])
    slots = {f["rid"]: f["penalty_slot"] for f in s.take_finished()}
    assert slots == {"a": 3, "b": -1}  # handed back with the result so that the engine can recycle the row


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_device_early_exit_agrees_with_the_scheduler(seed):
    """The bookkeeping the decode graph runs per sampled token (`decode_advance_`, CPU twin of the kernel) retires
    a row exactly where the scheduler ends the request: stop ids after min_new_tokens, max_new_tokens, context cap."""
    import torch

    from areal_vllm_amd import ops

    rng = np.random.RandomState(seed)
    k = 6
    s = _sched(num_pages=128, max_running=8, decode_steps=k, ctx=48, prefix=False)
    n = 8
    for i in range(n):
        _add(s, f"r{i}", [3] * int(rng.randint(2, 30)), int(rng.randint(1, 30)), min_new=int(rng.randint(0, 5)),
             stop_ids=[1, 2] if i % 2 else [1])
    p = s.schedule()
    s.process_prefill(rng.randint(1, 6, n).tolist(), [-1.0] * n, p["plan_id"])
    s.take_finished()
    while s.has_work():
        plan = s.schedule()
        assert plan["kind"] == 2
        B = len(plan["req_ids"])
        st = dict(input_ids=torch.from_numpy(plan["input_ids"].astype(np.int64)), positions=torch.from_numpy(plan["positions"].copy()),
                  context_lens=torch.from_numpy(plan["context_lens"].copy()), slots=torch.from_numpy(plan["slot_mapping"].copy()),
                  block_tables=torch.from_numpy(plan["block_tables"].copy()), active=torch.ones(B, dtype=torch.int32),
                  remaining=torch.from_numpy(plan["remaining"].copy()), min_remaining=torch.from_numpy(plan["min_remaining"].copy()),
                  stop_ids=torch.from_numpy(plan["stop_ids"].copy()))
        toks = rng.randint(1, 6, (k, B))
        alive_after = np.zeros((k, B), dtype=bool)
        for j in range(k):
            ops.decode_advance_(torch.from_numpy(toks[j].astype(np.int64)), st["input_ids"], st["positions"], st["context_lens"],
                                st["slots"], st["block_tables"], st["active"], st["remaining"], st["min_remaining"], st["stop_ids"])
            alive_after[j] = st["active"].numpy().astype(bool)
        ids = plan["req_ids"].tolist()
        s.process_decode(toks, np.full((k, B), -1.0, np.float32), plan["plan_id"])
        gone = {f["rid"]: f for f in s.take_finished()}
        # rows the device still counts as active are exactly the requests the scheduler keeps running
        nxt = s.schedule() if s.has_work() else None
        still = set(nxt["req_ids"].tolist()) if nxt is not None else set()
        assert {ids[i] for i in range(B) if alive_after[-1, i]} == still
        if nxt is not None:  # and their device-side state is what the next plan would upload
            for i in range(B):
                if alive_after[-1, i]:
                    r = nxt["req_ids"].tolist().index(ids[i])
                    assert int(st["positions"][i]) == int(nxt["positions"][r]) and int(st["input_ids"][i]) == int(nxt["input_ids"][r])
                    assert int(st["slots"][i]) == int(nxt["slot_mapping"][r]) and int(st["remaining"][i]) == int(nxt["remaining"][r])
            s.process_decode(np.full((k, len(nxt["req_ids"])), 1), np.full((k, len(nxt["req_ids"])), -1.0, np.float32),
                             nxt["plan_id"])  # token 1 stops everyone whose min_new_tokens allow it
            s.take_finished()
