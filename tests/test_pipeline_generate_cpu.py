"""Generation through a trainer's pipeline stages (``parallel/pipeline_generate.py``): a pipeline- (and tensor-)
parallel trainer must generate what the single-process trainer generates from the same checkpoint, and that in
turn what the rollout engine generates (greedy, fp32, gloo)."""
import os
import socket
import sys
import tempfile
import time

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROMPTS = [[5, 6, 7, 8, 9], [11, 12, 13], [100, 200, 300, 400, 17, 18, 19], [3, 4], [250, 251, 252, 253, 254, 255],
           [42] * 40, [77, 78, 79]]
MAX_NEW = 14


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, strategy_kwargs, ckpt_in, out_path, mode):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      LOCAL_RANK=str(rank), GLOO_SOCKET_IFNAME="lo")
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    from areal_vllm_amd.api.alloc_mode import ParallelStrategy
    from areal_vllm_amd.api.cli_args import GenerationHyperparameters, MicroBatchSpec, OptimizerConfig, \
        TrainEngineConfig
    from areal_vllm_amd.api.io_struct import FinetuneSpec, SaveLoadMeta
    from areal_vllm_amd.engine.train_engine import NativeTrainEngine

    cfg = TrainEngineConfig(experiment_name="t", trial_name="t", path=ckpt_in or "tiny", init_from_scratch=ckpt_in is None,
                            dtype="float32", optimizer=OptimizerConfig(lr=1e-3), mb_spec=MicroBatchSpec())
    eng = NativeTrainEngine(cfg)
    strategy_kwargs = dict(strategy_kwargs)
    n_mbs = strategy_kwargs.pop("_mbs", None)
    eng.create_process_group(ParallelStrategy(**strategy_kwargs))
    eng.initialize(None, FinetuneSpec(1, 8, 4))
    if mode == "init":
        eng.save(SaveLoadMeta(path=out_path, weight_format="hf", with_optim=False))
        eng.destroy()
        return
    dp, dpr = eng.data_parallel_world_size, eng.data_parallel_rank
    mine = PROMPTS[dpr::dp]
    if mode == "greedy":
        g = GenerationHyperparameters(max_new_tokens=MAX_NEW, greedy=True)
        res = eng.generate(mine, g, n_micro_batches=n_mbs)
        out = [(r.output_tokens, r.output_logprobs, r.finish_reason) for r in res]
    elif mode == "stop":
        # stop on the token the greedy run produced at index 3 of the first prompt: every stage must leave the
        # loop at the same check and the result must be cut after that token
        greedy = torch.load(os.path.join(os.path.dirname(out_path), "pp1.pt"), weights_only=True)
        stop = greedy[0][0][3]
        res = eng.generate(mine, None, max_new_tokens=MAX_NEW, min_new_tokens=2, greedy=True, stop_token_ids=[stop],
                           check_interval=2)
        out = [(r.output_tokens, r.output_logprobs, r.finish_reason) for r in res] + [stop]
    else:  # sampled: valid tokens, log-probs of a sampling distribution, reproducible shape
        res = eng.generate(mine, None, max_new_tokens=5, temperature=0.8, top_p=0.9, top_k=50)
        out = [(r.output_tokens, r.output_logprobs, r.finish_reason) for r in res]
    torch.save([dpr, out], out_path + f".r{rank}")
    if eng.grid.global_rank == 0:
        torch.save(out, out_path)
    eng.destroy()


def _run(world, strategy, ckpt_in, out_path, mode):
    port = _free_port()
    procs = mp.spawn(_worker, args=(world, port, dict(strategy), ckpt_in, out_path, mode), nprocs=world, join=False)
    deadline = time.time() + 300
    while not procs.join(timeout=5):  # raises when a rank failed; False while ranks are still running
        if time.time() > deadline:
            for p in procs.processes:
                p.kill()
            raise TimeoutError(f"{mode} on {world} ranks did not finish")


@pytest.fixture(scope="module")
def ckpt():
    d = tempfile.mkdtemp(prefix="areal_ppgen_")
    _run(1, dict(), None, os.path.join(d, "init"), "init")
    return d


def test_pipeline_generation_matches_single_process_and_rollout_engine(ckpt):
    init = os.path.join(ckpt, "init")
    _run(1, dict(), init, os.path.join(ckpt, "pp1.pt"), "greedy")
    want = torch.load(os.path.join(ckpt, "pp1.pt"), weights_only=True)
    assert len(want) == len(PROMPTS) and all(len(t) == MAX_NEW and r == "length" for t, _, r in want)

    # the rollout engine on the same weights
    sys.path.insert(0, ROOT)
    from areal_vllm_amd.api.cli_args import InferenceServerConfig
    from areal_vllm_amd.inference.engine import NativeInferenceEngine

    eng = NativeInferenceEngine(InferenceServerConfig(model_path=init, dtype="float32", context_length=128,
                                                      max_running_requests=8, disable_cuda_graph=True),
                                device=torch.device("cpu"), num_pages=64)
    eng.initialize(load_weights=True)
    ref = eng.generate_blocking(PROMPTS, max_new_tokens=MAX_NEW, greedy=True)
    eng.destroy()
    for (t, lp, _), r in zip(want, ref):
        assert t == r.output_tokens
        assert lp == pytest.approx(r.output_logprobs, abs=1e-4)

    for name, world, kw in [("pp2", 2, dict(pipeline_parallel_size=2)), ("pp2mb3", 2, dict(pipeline_parallel_size=2, _mbs=3)),
                            ("pp2tp2", 4, dict(pipeline_parallel_size=2, tensor_parallel_size=2)), ("dp2pp2", 4, dict(pipeline_parallel_size=2, data_parallel_size=2))]:
        out = os.path.join(ckpt, name + ".pt")
        _run(world, kw, init, out, "greedy")
        dp = kw.get("data_parallel_size", 1)
        got = [None] * len(PROMPTS)
        for r in range(world):
            dpr, part = torch.load(out + f".r{r}", weights_only=True)
            for i, item in zip(range(dpr, len(PROMPTS), dp), part):
                assert got[i] is None or got[i] == item  # every rank of a replica returns the same sequences
                got[i] = item
        for (t, lp, fr), (t2, lp2, fr2) in zip(want, got):
            assert t == t2 and fr == fr2, name
            assert lp == pytest.approx(lp2, abs=1e-4), name


def test_pipeline_generation_stops_together_and_samples(ckpt):
    init = os.path.join(ckpt, "init")
    if not os.path.exists(os.path.join(ckpt, "pp1.pt")):
        _run(1, dict(), init, os.path.join(ckpt, "pp1.pt"), "greedy")
    greedy = torch.load(os.path.join(ckpt, "pp1.pt"), weights_only=True)
    out = os.path.join(ckpt, "stop.pt")
    _run(2, dict(pipeline_parallel_size=2), init, out, "stop")
    res = torch.load(out, weights_only=True)
    stop = res.pop()
    for (t, _, fr), (tg, _, _) in zip(res, greedy):
        if stop in tg[2:]:  # the first two tokens may not end the sequence (min_new_tokens)
            k = tg.index(stop, 2)
            assert t == tg[:k + 1] and fr == "stop"
        else:
            assert t == tg and fr == "length"
    assert res[0][2] == "stop" and res[0][0][-1] == stop and len(res[0][0]) <= 4
    out = os.path.join(ckpt, "sampled.pt")
    _run(2, dict(pipeline_parallel_size=2), init, out, "sample")
    res = torch.load(out, weights_only=True)
    assert len(res) == len(PROMPTS)
    for t, lp, fr in res:
        assert len(t) == 5 and all(0 <= x < 512 for x in t) and all(v <= 1e-6 for v in lp)
