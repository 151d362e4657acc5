"""Parameter re-allocation between layouts: plan algebra on one process, then a real 4-process gloo run
(pattern of the reference's ``realhf/tests/comm/test_param_realloc.py``: every (dp,tp,pp) -> (dp',tp',pp'))."""
import itertools
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from areal_vllm_amd.models import hf_io
from areal_vllm_amd.models.config import preset
from areal_vllm_amd.models.transformer import Transformer
from areal_vllm_amd.parallel.param_realloc import Layout, derive_realloc_plan, realloc_parameters, shard_pieces
from areal_vllm_amd.parallel.pipeline import stage_layer_range


def _cfg(**kw):
    return preset("tiny", n_layers=3, n_q_heads=4, n_kv_heads=2, head_dim=64, intermediate_size=512, **kw)


def _shard(cfg, lay, tp, pp):
    m = Transformer(cfg, dtype=torch.float32, device="cpu", tp_size=lay.tp, tp_rank=tp,
                    layer_range=stage_layer_range(cfg.n_layers, lay.pp, pp), has_embed=pp == 0,
                    has_head=pp == lay.pp - 1, vocab_parallel=lay.vocab_parallel)
    m.flatten()
    return m


def _apply_locally(plan, src_flats, dst_flats):
    for t in plan:
        for (a, b), (c, d) in zip(t.src_intervals, t.dst_intervals):
            dst_flats[t.dst][c:d] = src_flats[t.src][a:b]


def _reference_shards(cfg, full, lay):
    """Shards produced by the HF-name based loader (independent of the piece formulas)."""
    from areal_vllm_amd.inference.engine import assign_hf_named_tensors

    tensors = dict(hf_io.native_to_hf_state(full, gather=False))
    out = {}
    for pp in range(lay.pp):
        for tp in range(lay.tp):
            m = _shard(cfg, lay, tp, pp)
            assign_hf_named_tensors(m, tensors)
            if cfg.tie_word_embeddings and "head" in dict(m.named_parameters()):
                v_lo, v_hi = m.vocab_range
                m.head.data.copy_(full.embed.data[v_lo:v_hi])
            out[(pp, tp)] = m
    return out


@pytest.mark.parametrize("tied", [True, False])
@pytest.mark.parametrize("dst", [Layout(1, 2, 1), Layout(1, 4, 1), Layout(2, 1, 3), Layout(1, 2, 2),
                                 Layout(1, 2, 2, vocab_parallel=False)])
def test_plan_matches_hf_sharding_and_round_trips(dst, tied):
    cfg = _cfg(tie_word_embeddings=tied)
    full = _shard(cfg, Layout(), 0, 0)
    full.init_weights(seed=3)
    one = Layout()
    plan = derive_realloc_plan(cfg, one, dst, dtype=torch.float32)
    ref = _reference_shards(cfg, full, dst)
    dst_flats = {}
    for pp in range(dst.pp):
        for dp in range(dst.dp):
            for tp in range(dst.tp):
                dst_flats[dst.index_of(pp, dp, tp)] = torch.full((ref[(pp, tp)].flat.numel,), float("nan"))
    _apply_locally(plan, {0: full.flat.flat}, dst_flats)
    for pp in range(dst.pp):
        for dp in range(dst.dp):
            for tp in range(dst.tp):
                got, want = dst_flats[dst.index_of(pp, dp, tp)], ref[(pp, tp)]
                for n, (o, numel) in want.flat.offsets.items():
                    assert torch.equal(got[o:o + numel], want.flat.flat[o:o + numel]), (pp, dp, tp, n)
    # and back to a single shard through an intermediate layout with a different tp / pp
    mid = Layout(1, 2, 3) if dst.tp != 2 or dst.pp != 3 else Layout(1, 4, 1)
    p1 = derive_realloc_plan(cfg, dst, mid, dtype=torch.float32)
    mid_flats = {mid.index_of(pp, 0, tp): torch.full((_shard(cfg, mid, tp, pp).flat.numel,), float("nan"))
                 for pp in range(mid.pp) for tp in range(mid.tp)}
    _apply_locally(p1, dst_flats, mid_flats)
    p2 = derive_realloc_plan(cfg, mid, one, dtype=torch.float32)
    back = {0: torch.full((full.flat.numel,), float("nan"))}
    _apply_locally(p2, mid_flats, back)
    for n, (o, numel) in full.flat.offsets.items():
        assert torch.equal(back[0][o:o + numel], full.flat.flat[o:o + numel]), n


@pytest.mark.parametrize("name,dst", [("tiny-gpt2", Layout(1, 2, 1)), ("tiny-gpt2", Layout(1, 2, 2)),
                                      ("tiny-moe", Layout(1, 2, 2)), ("tiny-moe", Layout(1, 1, 2))])
def test_other_families_match_hf_sharding(name, dst):
    """Ungated MLPs with biases (GPT-2) and routed experts (split over tp, spread over pipeline stages) against
    the shards the HF-name based loader produces."""
    cfg = preset(name)
    full = _shard(cfg, Layout(), 0, 0)
    full.init_weights(seed=4)
    for n, p in full.named_parameters():
        if n.endswith("bias"):
            p.data.normal_(generator=torch.Generator().manual_seed(len(n)))
    ref = _reference_shards(cfg, full, dst)
    flats = {dst.index_of(pp, 0, tp): torch.full((ref[(pp, tp)].flat.numel,), float("nan"))
             for pp in range(dst.pp) for tp in range(dst.tp)}
    _apply_locally(derive_realloc_plan(cfg, Layout(), dst, dtype=torch.float32), {0: full.flat.flat}, flats)
    for (pp, tp), want in ref.items():
        for n, (o, numel) in want.flat.offsets.items():
            assert torch.equal(flats[dst.index_of(pp, 0, tp)][o:o + numel], want.flat.flat[o:o + numel]), (pp, tp, n)
    back = {0: torch.full((full.flat.numel,), float("nan"))}
    _apply_locally(derive_realloc_plan(cfg, dst, Layout(), dtype=torch.float32), flats, back)
    for n, (o, numel) in full.flat.offsets.items():
        assert torch.equal(back[0][o:o + numel], full.flat.flat[o:o + numel]), n


def test_kv_head_replication_when_tp_exceeds_kv_heads():
    cfg = _cfg()
    lay = Layout(1, 4, 1)  # 2 kv heads over 4 tp ranks: neighbouring ranks share a kv head
    pieces = [shard_pieces(_shard(cfg, lay, t, 0)) for t in range(4)]
    k0, k1 = pieces[0]["layers.0.attn.qkv"], pieces[1]["layers.0.attn.qkv"]
    assert np.array_equal(k0[1:, 1:], k1[1:, 1:]) and not np.array_equal(k0[0, 1:], k1[0, 1:])
    plan = derive_realloc_plan(cfg, lay, Layout(1, 2, 1), dtype=torch.float32)
    assert {(t.src, t.dst) for t in plan} >= {(0, 0), (1, 0), (2, 1), (3, 1)}


def _realloc_worker(rank, world, port, q):
    os.environ["GLOO_SOCKET_IFNAME"] = "lo"
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    cfg = _cfg()
    src, dst = Layout(1, 2, 1), Layout(2, 1, 2)
    src_ranks, dst_ranks = [2, 3], [0, 1, 2, 3]  # the two models overlap on ranks 2 and 3
    full = _shard(cfg, Layout(), 0, 0)
    full.init_weights(seed=5)
    src_model = dst_model = None
    if rank in src_ranks:
        src_model = _reference_shards(cfg, full, src)[(0, src_ranks.index(rank))]
    if rank in dst_ranks:
        i = dst_ranks.index(rank)
        pp, dp = i // dst.dp, i % dst.dp
        dst_model = _shard(cfg, dst, 0, pp)
        dst_model.flat.flat.fill_(float("nan"))
        want = _reference_shards(cfg, full, dst)[(pp, 0)]
    plan = derive_realloc_plan(cfg, src, dst, dtype=torch.float32)
    n = realloc_parameters(plan, src_ranks, dst_ranks, src_model.flat.flat if src_model else None,
                           dst_model.flat.flat if dst_model else None)
    ok = n > 0
    for name, (o, numel) in want.flat.offsets.items():
        ok &= bool(torch.equal(dst_model.flat.flat[o:o + numel], want.flat.flat[o:o + numel]))
    # EMA mixing: dst <- 0.25 * src + 0.75 * dst on a buffer of zeros gives a quarter of the weights
    dst_model.flat.flat.zero_()
    realloc_parameters(plan, src_ranks, dst_ranks, src_model.flat.flat if src_model else None, dst_model.flat.flat,
                       eta=0.25)
    o, numel = want.flat.offsets["layers.0.attn.o"]
    ok &= bool(torch.allclose(dst_model.flat.flat[o:o + numel], 0.25 * want.flat.flat[o:o + numel]))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_realloc_over_gloo_with_overlapping_models():
    from areal_vllm_amd.utils.network import find_free_ports

    port = find_free_ports(1)[0]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_realloc_worker, args=(r, 4, port, q)) for r in range(4)]
    [p.start() for p in ps]
    res = dict(q.get(timeout=180) for _ in range(4))
    [p.join(30) for p in ps]
    assert res == {0: True, 1: True, 2: True, 3: True}
