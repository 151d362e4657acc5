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
    n_preempt = int(r.stdout.strip().split()[-2])
    assert n_preempt > 0  # the pool was tight enough to exercise preemption
