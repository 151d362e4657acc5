"""Multi-process (gloo) parity tests: every parallel layout must take the same optimizer steps as the
single-process engine on the same data. Each layout trains two steps from the same HF checkpoint and
writes a HF checkpoint; the test compares the written tensors."""
import os
import socket
import sys
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_batch(seed=0, B=4, vocab=512):
    g = torch.Generator().manual_seed(seed)
    lens = [23, 9, 31, 14][:B]
    S = max(lens)
    ids = torch.zeros(B, S, dtype=torch.long)
    am = torch.zeros(B, S, dtype=torch.bool)
    lm = torch.zeros(B, S, dtype=torch.bool)
    for i, n in enumerate(lens):
        ids[i, :n] = torch.randint(2, vocab, (n,), generator=g)
        am[i, :n] = True
        lm[i, n // 3:n - 1] = True  # already shifted: position t scores token t+1
    return dict(input_ids=ids, attention_mask=am, loss_mask=lm)


def _sft_loss(proxy, mb):
    labels = torch.roll(mb["input_ids"], -1, 0)
    lp = proxy.logprobs(labels)
    m = mb["loss_mask"].float()
    return -(lp * m).sum() / m.sum().clamp(min=1)


def _weight(mb):
    return mb["loss_mask"].float().sum()


def _worker(rank, world, port, strategy_kwargs, ckpt_in, ckpt_out, preset_name, n_mbs):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      LOCAL_RANK=str(rank), GLOO_SOCKET_IFNAME="lo")
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    forced = dict(strategy_kwargs).get("_force", False)
    if forced:  # a one-rank group whose collectives are really issued (read when the package is imported)
        os.environ.update(AREAL_FORCE_COLLECTIVES="1", AREAL_ZERO_BUCKETS="5")
    from areal_vllm_amd.api.alloc_mode import ParallelStrategy
    from areal_vllm_amd.api.cli_args import MicroBatchSpec, OptimizerConfig, TrainEngineConfig
    from areal_vllm_amd.api.io_struct import FinetuneSpec, SaveLoadMeta
    from areal_vllm_amd.engine.train_engine import NativeTrainEngine

    cfg = TrainEngineConfig(experiment_name="t", trial_name="t", path=ckpt_in or preset_name,
                            init_from_scratch=ckpt_in is None, dtype="float32", gradient_checkpointing=False,
                            optimizer=OptimizerConfig(lr=1e-2, weight_decay=0.01, gradient_clipping=1.0,
                                                      warmup_steps_proportion=0.0, lr_scheduler_type="constant"),
                            mb_spec=MicroBatchSpec(n_mbs=n_mbs, max_tokens_per_mb=4096))
    strategy_kwargs = dict(strategy_kwargs)
    strategy_kwargs.pop("_force", None)
    cfg.native.context_parallel_mode = strategy_kwargs.pop("_cp_mode", "ulysses")
    cfg.native.sequence_parallel = strategy_kwargs.pop("_seq_par", False)
    reshard_from = strategy_kwargs.pop("_reshard_from", None)
    z3 = strategy_kwargs.pop("_zero3", None)
    if z3 is not None:
        cfg.native.zero_stage = 3
        cfg.native.reshard_after_forward = z3.get("reshard", True)
        cfg.native.offload_optimizer = z3.get("offload", False)
        cfg.gradient_checkpointing = z3.get("ckpt", False)
    eng = NativeTrainEngine(cfg)
    if "virtual_pipeline_parallel_size" in strategy_kwargs:
        from areal_vllm_amd.api.alloc_mode import MegatronParallelStrategy

        st = MegatronParallelStrategy(**strategy_kwargs)
    else:
        st = ParallelStrategy(**strategy_kwargs)
    eng.create_process_group(st)
    eng.initialize(None, FinetuneSpec(1, 8, 4))
    if ckpt_in is None:
        eng.save(SaveLoadMeta(path=ckpt_out, weight_format="hf", with_optim=False))
        eng.destroy()
        return
    if reshard_from is not None:
        # load a sharded checkpoint written under ANOTHER (tp, pp) layout, write it back as HF
        eng.model.flat.flat.zero_()
        if eng.model.flat_experts is not None:
            eng.model.flat_experts.flat.zero_()
        eng.load(SaveLoadMeta(path=reshard_from, weight_format="dcp", with_optim=False))
        eng.save(SaveLoadMeta(path=ckpt_out, weight_format="hf", with_optim=False))
        eng.destroy()
        return
    batch = _make_batch()
    dp, dpr = eng.data_parallel_world_size, eng.data_parallel_rank
    B = batch["input_ids"].shape[0]
    per = B // dp
    local = {k: v[dpr * per:(dpr + 1) * per] for k, v in batch.items()}
    stats = []
    for _ in range(2):
        stats.append(eng.train_batch(local, _sft_loss, _weight))
        eng.step_lr_scheduler()
    if eng.zero3 is None and eng.grid.pp == 1 and (eng.grid.dp * eng.grid.cp > 1 or forced) \
            and eng.expert_optimizer is None \
            and eng._can_overlap_grad_reduce():
        # both steps sent (almost) all gradient buckets on their way while the last backward was still running
        # (all but the lowest bucket, which holds the embedding table that reports after the backward)
        assert len(eng.optimizer.buckets) > 1
        assert eng.optimizer.overlapped_buckets >= 2 * (len(eng.optimizer.buckets) - 1), \
            (eng.optimizer.overlapped_buckets, len(eng.optimizer.buckets))
    ev = eng.eval_batch(local, _sft_loss, _weight)
    fw = eng.forward(local, post_hook=lambda proxy, mb: proxy.logprobs(torch.roll(mb["input_ids"], -1, 0)))
    eng.save(SaveLoadMeta(path=ckpt_out, weight_format="hf", with_optim=False))
    if rank == 0:
        torch.save(dict(grad_norm=[s["grad_norm"] for s in stats], eval=float(ev)), os.path.join(ckpt_out, "stats.pt"))
    torch.save(fw, os.path.join(ckpt_out, f"fwd_dp{dpr}_r{rank}.pt"))
    # sharded ("dcp") checkpoint round trip in the same layout: weights and optimizer state come back exactly
    shard_dir = os.path.join(ckpt_out, "sharded")
    eng.save(SaveLoadMeta(path=shard_dir, weight_format="dcp", with_optim=True))
    if eng.zero3 is not None:
        z = eng.zero3
        # at rest a rank holds its shards only: no unit buffer has storage
        assert z.resident_param_bytes() == z.param_shard.numel() * z.param_shard.element_size()
        assert z.resident_param_bytes() * z.n <= z.full_param_bytes() + 64 * z.n * len(z.units) * 4
        want_w, want_m = z.param_shard.clone(), z.m.clone()
        z.param_shard.add_(1.0)
        z.m.zero_()
        eng.load(SaveLoadMeta(path=shard_dir, weight_format="dcp", with_optim=True))
        assert torch.equal(z.param_shard, want_w) and torch.equal(z.m, want_m)
        assert torch.equal(z.master, want_w.float())
        eng.destroy()
        return
    want_w, want_m = eng.model.flat.flat.clone(), eng.optimizer.m.clone()
    want_e = eng.model.flat_experts.flat.clone() if eng.model.flat_experts is not None else None
    eng.model.flat.flat.add_(1.0)
    eng.optimizer.m.zero_()
    if want_e is not None:
        eng.model.flat_experts.flat.zero_()
    eng.load(SaveLoadMeta(path=shard_dir, weight_format="dcp", with_optim=True))
    assert torch.equal(eng.model.flat.flat, want_w) and torch.equal(eng.optimizer.m, want_m)
    if want_e is not None:  # every expert-parallel rank gets ITS experts back, not rank 0's
        assert torch.equal(eng.model.flat_experts.flat, want_e)
    assert torch.equal(eng.optimizer.master, torch.cat([want_w[a:b] for a, b in eng.optimizer.owned_ranges()]).float())
    eng.destroy()


def _run(world, strategy_kwargs, ckpt_in, ckpt_out, preset_name="tiny", n_mbs=1):
    port = _free_port()
    if world == 1:
        ctx = mp.get_context("spawn")
        p = ctx.Process(target=_worker, args=(0, 1, port, strategy_kwargs, ckpt_in, ckpt_out, preset_name, n_mbs))
        p.start()
        p.join(600)
        assert p.exitcode == 0
    else:
        mp.spawn(_worker, args=(world, port, strategy_kwargs, ckpt_in, ckpt_out, preset_name, n_mbs), nprocs=world,
                 join=True)


def _load_ckpt(path):
    from safetensors.torch import load_file

    out = {}
    for f in sorted(os.listdir(path)):
        if f.endswith(".safetensors"):
            out.update(load_file(os.path.join(path, f)))
    return out


@pytest.fixture(scope="module")
def dense_reference():
    d = tempfile.mkdtemp(prefix="areal_par_")
    init, ref = os.path.join(d, "init"), os.path.join(d, "ref")
    _run(1, {}, None, init)
    _run(1, {}, init, ref, n_mbs=2)
    return d, init, ref


def _compare(ref, out, atol=2e-5):
    a, b = _load_ckpt(ref), _load_ckpt(out)
    assert set(a) == set(b), set(a) ^ set(b)
    for k in a:
        torch.testing.assert_close(b[k], a[k], atol=atol, rtol=1e-4, msg=lambda m: f"{k}: {m}")
    sa = torch.load(os.path.join(ref, "stats.pt"), weights_only=True)
    sb = torch.load(os.path.join(out, "stats.pt"), weights_only=True)
    assert sa["grad_norm"] == pytest.approx(sb["grad_norm"], rel=1e-3)
    assert sa["eval"] == pytest.approx(sb["eval"], rel=1e-4)


@pytest.mark.parametrize("name,world,kw,n_mbs", [
    ("dp2", 2, dict(data_parallel_size=2), 1),
    ("tp2", 2, dict(tensor_parallel_size=2), 2),
    ("cp2", 2, dict(context_parallel_size=2), 2),
    ("pp2", 2, dict(pipeline_parallel_size=2), 3),
    ("dp2pp2", 4, dict(data_parallel_size=2, pipeline_parallel_size=2), 2),
    ("dp2tp2", 4, dict(data_parallel_size=2, tensor_parallel_size=2), 1),
    ("cp2pp2", 4, dict(context_parallel_size=2, pipeline_parallel_size=2), 2),
    ("tp2sp", 2, dict(tensor_parallel_size=2, _seq_par=True), 2),
    ("dp2tp2sp", 4, dict(data_parallel_size=2, tensor_parallel_size=2, _seq_par=True), 1),
    ("tp2pp2sp", 4, dict(tensor_parallel_size=2, pipeline_parallel_size=2, _seq_par=True), 2),
    ("zigzag2", 2, dict(context_parallel_size=2, _cp_mode="zigzag"), 2),
    ("zigzag4", 4, dict(context_parallel_size=4, _cp_mode="zigzag"), 1),
    ("zigzag2tp2", 4, dict(context_parallel_size=2, tensor_parallel_size=2, _cp_mode="zigzag"), 2),
    ("dp1-collectives-forced", 1, dict(_force=True), 2),
])
def test_layout_matches_single_process(dense_reference, name, world, kw, n_mbs):
    d, init, ref = dense_reference
    out = os.path.join(d, name)
    _run(world, kw, init, out, n_mbs=n_mbs)
    _compare(ref, out)
    # forward() returns the full, correctly ordered per-token result on every rank of a replica
    ref_fw = torch.load(os.path.join(ref, "fwd_dp0_r0.pt"), weights_only=True)
    fws = sorted(f for f in os.listdir(out) if f.startswith("fwd_dp0_"))
    for f in fws:
        fw = torch.load(os.path.join(out, f), weights_only=True)
        # the last token of a sequence scores its packed neighbour's first token: layout dependent
        for i, n in enumerate([23, 9, 31, 14][: fw.shape[0]]):
            torch.testing.assert_close(fw[i, : n - 1], ref_fw[i, : n - 1], atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("src,dst_world,dst_kw", [
    ("tp2", 1, {}),
    ("pp2", 2, dict(tensor_parallel_size=2)),
    ("dp2tp2", 4, dict(tensor_parallel_size=2, pipeline_parallel_size=2)),
])
def test_sharded_checkpoint_loads_into_another_layout(dense_reference, src, dst_world, dst_kw):
    """A sharded checkpoint written as tp2 / pp2 / dp2tp2 is read by an engine laid out differently (every
    rank pulls the intervals that make up its own flat buffer from the source shard files)."""
    d, init, ref = dense_reference
    src_dir = os.path.join(d, src)
    if not os.path.isdir(os.path.join(src_dir, "sharded")):
        kw, world, n_mbs = {"tp2": (dict(tensor_parallel_size=2), 2, 2), "pp2": (dict(pipeline_parallel_size=2), 2, 3),
                            "dp2tp2": (dict(data_parallel_size=2, tensor_parallel_size=2), 4, 1)}[src]
        _run(world, kw, init, src_dir, n_mbs=n_mbs)
    out = os.path.join(d, f"reshard_{src}_to_{dst_world}")
    _run(dst_world, dict(dst_kw, _reshard_from=os.path.join(src_dir, "sharded")), init, out)
    a, b = _load_ckpt(src_dir), _load_ckpt(out)
    assert set(a) == set(b)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_expert_parallel_matches_single_process():
    d = tempfile.mkdtemp(prefix="areal_ep_")
    init, ref, out = (os.path.join(d, x) for x in ("init", "ref", "ep2"))
    _run(1, {}, None, init, preset_name="tiny-moe")
    _run(1, {}, init, ref, preset_name="tiny-moe")
    _run(2, dict(data_parallel_size=2, expert_parallel_size=2), init, out, preset_name="tiny-moe")
    _compare(ref, out, atol=5e-5)


@pytest.mark.parametrize("name,world,kw", [
    ("tp2ep2etp1", 2, dict(tensor_parallel_size=2, expert_parallel_size=2, expert_tensor_parallel_size=1)),
    ("dp2tp2ep2etp1", 4, dict(data_parallel_size=2, tensor_parallel_size=2, expert_parallel_size=2,
                              expert_tensor_parallel_size=1)),
    ("dp2tp2ep4etp1", 4, dict(data_parallel_size=2, tensor_parallel_size=2, expert_parallel_size=4,
                              expert_tensor_parallel_size=1)),
])
def test_whole_experts_inside_a_tensor_parallel_group_match_single_process(name, world, kw):
    """Expert tensor parallel size 1 under attention tp 2: experts are not split, every tp rank routes its slice of
    the tokens, the expert-parallel groups run over (dp x tp) ranks. Same weights after two steps as one process;
    the hf and the sharded checkpoints (each expert set written once) round-trip."""
    d = tempfile.mkdtemp(prefix="areal_etp_")
    init, ref, out = (os.path.join(d, x) for x in ("init", "ref", name))
    _run(1, {}, None, init, preset_name="tiny-moe")
    _run(1, {}, init, ref, preset_name="tiny-moe")
    _run(world, kw, init, out, preset_name="tiny-moe")
    _compare(ref, out, atol=5e-5)


def test_moe_sharded_checkpoint_loads_into_other_layouts():
    """Sharded checkpoints of a mixture-of-experts model cross layouts in every direction: written with expert
    parallelism (experts in their own files) and read without; written by pipeline stages and read by expert-parallel
    ranks that split every expert over a tensor-parallel pair; written with whole experts inside a tensor-parallel
    group and read by two pipeline stages. The hf export of the reader equals the hf export of the writer bit for bit."""
    d = tempfile.mkdtemp(prefix="areal_moe_reshard_")
    init = os.path.join(d, "init")
    _run(1, {}, None, init, preset_name="tiny-moe")
    writers = {"ep2": (2, dict(data_parallel_size=2, expert_parallel_size=2)),
               "pp2": (2, dict(pipeline_parallel_size=2)),
               "tp2ep2etp1": (2, dict(tensor_parallel_size=2, expert_parallel_size=2, expert_tensor_parallel_size=1))}
    for name, (world, kw) in writers.items():
        _run(world, kw, init, os.path.join(d, name), preset_name="tiny-moe")
    readers = [("ep2", "to_single", 1, {}),
               ("ep2", "to_tp2", 2, dict(tensor_parallel_size=2)),
               ("pp2", "to_dp2tp2ep2", 4, dict(data_parallel_size=2, tensor_parallel_size=2, expert_parallel_size=2)),
               ("tp2ep2etp1", "to_pp2", 2, dict(pipeline_parallel_size=2)),
               ("pp2", "to_tp2ep2etp1", 2, dict(tensor_parallel_size=2, expert_parallel_size=2,
                                                expert_tensor_parallel_size=1))]
    for src, name, world, kw in readers:
        out = os.path.join(d, f"{src}_{name}")
        _run(world, dict(kw, _reshard_from=os.path.join(d, src, "sharded")), init, out, preset_name="tiny-moe")
        a, b = _load_ckpt(os.path.join(d, src)), _load_ckpt(out)
        assert set(a) == set(b), (src, name)
        for k in a:
            assert torch.equal(a[k], b[k]), (src, name, k)


def test_qk_norm_weights_under_tensor_parallel():
    """Per-head q/k norm weights are replicated over tp but every rank only sees its own heads: their
    gradients must be summed over the tp group or the replicas drift apart."""
    d = tempfile.mkdtemp(prefix="areal_qkn_")
    init, ref, out = (os.path.join(d, x) for x in ("init", "ref", "tp2"))
    _run(1, {}, None, init, preset_name="tiny-qknorm")
    _run(1, {}, init, ref, preset_name="tiny-qknorm")
    _run(2, dict(tensor_parallel_size=2), init, out, preset_name="tiny-qknorm")
    _compare(ref, out, atol=5e-5)


def test_interleaved_pipeline_stages_match_single_process():
    """Virtual pipeline stages (two layer chunks per rank, five layers over four virtual stages): the interleaved
    1F1B schedule, its all-forward-all-backward form for a micro-batch count that is no multiple of pp, and the
    combination with tensor + sequence parallelism end at the single-process weights; eval and forward agree;
    the sharded checkpoint round-trips in the interleaved layout and loads into / from other layouts."""
    d = tempfile.mkdtemp(prefix="areal_vpp_")
    init, ref = os.path.join(d, "init"), os.path.join(d, "ref")
    _run(1, {}, None, init, preset_name="tiny-deep")
    _run(1, {}, init, ref, preset_name="tiny-deep", n_mbs=2)
    v2 = dict(pipeline_parallel_size=2, virtual_pipeline_parallel_size=2)
    for name, world, kw, n_mbs in (("pp2v2", 2, v2, 4), ("pp2v2odd", 2, v2, 3),
                                   ("tp2pp2v2sp", 4, dict(v2, tensor_parallel_size=2, _seq_par=True), 2)):
        out = os.path.join(d, name)
        _run(world, kw, init, out, preset_name="tiny-deep", n_mbs=n_mbs)
        _compare(ref, out, atol=5e-5)
        ref_fw = torch.load(os.path.join(ref, "fwd_dp0_r0.pt"), weights_only=True)
        for f in sorted(f for f in os.listdir(out) if f.startswith("fwd_dp0_")):
            fw = torch.load(os.path.join(out, f), weights_only=True)
            for i, n in enumerate([23, 9, 31, 14][: fw.shape[0]]):
                torch.testing.assert_close(fw[i, : n - 1], ref_fw[i, : n - 1], atol=1e-4, rtol=1e-4)
    # resharding loads across interleaved layouts: interleaved -> tensor parallel, and that one's -> interleaved
    for name, world, kw, src in (("v2_to_tp2", 2, dict(tensor_parallel_size=2), "pp2v2"),
                                 ("tp2pp2v2_to_v2", 2, v2, "tp2pp2v2sp")):
        out = os.path.join(d, name)
        _run(world, dict(kw, _reshard_from=os.path.join(d, src, "sharded")), init, out, preset_name="tiny-deep")
        a, b = _load_ckpt(os.path.join(d, src)), _load_ckpt(out)
        assert set(a) == set(b)
        for k in a:
            assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("name,world,kw,n_mbs", [
    ("z3dp2", 2, dict(data_parallel_size=2, _zero3={}), 2),
    ("z3dp2keep", 2, dict(data_parallel_size=2, _zero3=dict(reshard=False, ckpt=True)), 1),
    ("z3dp2offload", 2, dict(data_parallel_size=2, _zero3=dict(offload=True, ckpt=True)), 2),
    ("z3cp2", 2, dict(context_parallel_size=2, _zero3={}), 2),
    ("z3dp1", 1, dict(_zero3={}), 2),
    ("z3dp1-collectives-forced", 1, dict(_zero3={}, _force=True), 2),
    ("z3tp2", 2, dict(tensor_parallel_size=2, _zero3={}), 2),
    ("z3tp2sp", 2, dict(tensor_parallel_size=2, _seq_par=True, _zero3=dict(ckpt=True)), 2),
    ("z3dp2tp2", 4, dict(data_parallel_size=2, tensor_parallel_size=2, _zero3={}), 1),
])
def test_zero3_matches_single_process(dense_reference, name, world, kw, n_mbs):
    """ZeRO-3 (weights, gradients, optimizer state sharded; blocks gathered on demand, with and without
    re-sharding after the forward, gradient checkpointing, host-resident optimizer state; alone and under tensor
    parallelism, where every tp rank shards its slice over its own data-parallel group) takes the same two
    optimizer steps as the single-process ZeRO-1 run; eval / forward / hf save / sharded save + load work from
    the sharded state."""
    d, init, ref = dense_reference
    out = os.path.join(d, name)
    _run(world, kw, init, out, n_mbs=n_mbs)
    _compare(ref, out)
    ref_fw = torch.load(os.path.join(ref, "fwd_dp0_r0.pt"), weights_only=True)
    for f in sorted(f for f in os.listdir(out) if f.startswith("fwd_dp0_")):
        fw = torch.load(os.path.join(out, f), weights_only=True)
        for i, n in enumerate([23, 9, 31, 14][: fw.shape[0]]):
            torch.testing.assert_close(fw[i, : n - 1], ref_fw[i, : n - 1], atol=1e-4, rtol=1e-4)


def test_gpt2_family_under_tensor_and_pipeline_parallel():
    """GPT-2 (LayerNorm with bias, learned positions, ungated MLP, biases after the row-parallel
    reductions): two optimizer steps under tp=2 and under pp=2 end at the single-process weights, through
    the Conv1D-layout checkpoint written and re-read in every layout."""
    d = tempfile.mkdtemp(prefix="areal_gpt2_")
    init, ref = os.path.join(d, "init"), os.path.join(d, "ref")
    _run(1, {}, None, init, preset_name="tiny-gpt2")
    assert "transformer.h.0.attn.c_attn.weight" in _load_ckpt(init)
    _run(1, {}, init, ref, preset_name="tiny-gpt2")
    for name, kw in (("tp2", dict(tensor_parallel_size=2)), ("pp2", dict(pipeline_parallel_size=2))):
        out = os.path.join(d, name)
        _run(2, kw, init, out, preset_name="tiny-gpt2", n_mbs=2)
        _compare(ref, out, atol=5e-5)


def _bucket_order_worker(rank, world, port, out_dir):
    """Drive the ZeRO-1 optimizer's overlapped reduction by hand: the ranks report their parameters in DIFFERENT
    orders and rank 1 never reports one of them (a parameter its last micro-batch did not use)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo", AREAL_PLATFORM="cpu")
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from areal_vllm_amd.api.cli_args import OptimizerConfig
    from areal_vllm_amd.models.transformer import FlatParams
    from areal_vllm_amd.parallel.zero import ShardedAdamW

    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    mod = torch.nn.ParameterDict({f"w{i}": torch.nn.Parameter(torch.randn(300 + 64 * i)) for i in range(6)})
    flat = FlatParams(mod, torch.float32, "cpu", pad_to=world)
    flat.alloc_main_grad()
    cfg = OptimizerConfig(lr=1e-2, weight_decay=0.0, gradient_clipping=0.0)
    opt = ShardedAdamW(flat, cfg, dp_group=None, n_buckets=4)
    assert len(opt.buckets) == 4 and sum(b - a for a, b in opt.owned_ranges()) == flat.numel // world
    start = flat.flat.clone()
    g = torch.Generator().manual_seed(7)
    grads = [torch.randn(flat.numel, generator=g) for _ in range(world)]
    names = [n for n, _, _ in flat.specs]
    # which parameter is skipped, and the order of the reports, differ per rank
    order = list(reversed(names)) if rank == 0 else [names[i] for i in (4, 5, 0, 3, 1)]
    for step in range(2):
        flat.main_grad.copy_(grads[rank])
        if rank == 1:  # the unused parameter has no gradient on this rank
            o, numel = flat.offsets[names[2]]
            flat.main_grad[o:o + numel].zero_()
        opt.arm_overlap()
        for n in order:
            opt._param_done(flat._named[n])
        opt.step()
    # reference: plain AdamW on the summed gradients
    want = start.clone()
    total = grads[0].clone()
    g1 = grads[1].clone()
    o, numel = flat.offsets[names[2]]
    g1[o:o + numel].zero_()
    total += g1
    m, v = torch.zeros_like(want), torch.zeros_like(want)
    for step in range(1, 3):
        m.mul_(cfg.beta1).add_(total, alpha=1 - cfg.beta1)
        v.mul_(cfg.beta2).addcmul_(total, total, value=1 - cfg.beta2)
        denom = v.sqrt() / (1 - cfg.beta2 ** step) ** 0.5 + cfg.eps
        want.addcdiv_(m, denom, value=-cfg.lr / (1 - cfg.beta1 ** step))
    torch.testing.assert_close(flat.flat, want, atol=1e-5, rtol=1e-5)
    # a parameter that reports more often than its buckets expect is refused instead of launching a bucket early
    opt.arm_overlap()
    opt._param_done(flat._named[names[5]])
    try:
        opt._param_done(flat._named[names[5]])
        raised = False
    except AssertionError:
        raised = True
    assert raised
    torch.save(flat.flat, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_bucket_reduction_is_order_independent(tmp_path):
    """Gradient buckets leave in one fixed order on every rank, whatever order the parameters finish in and even if a
    rank never sees a gradient for some parameter (reference semantics: DDP bucket order,
    realhf/impl/model/backend/megatron.py:561-695)."""
    mp.spawn(_bucket_order_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = (torch.load(tmp_path / f"r{r}.pt", weights_only=True) for r in range(2))
    assert torch.equal(a, b)
