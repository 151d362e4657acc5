"""The fused decode-layer path of the rollout engine (decode GEMMs with split-K slabs + the row kernels that
consume them) against the plain path (library GEMMs + separate norm / RoPE / SwiGLU ops), on the PyTorch
fallbacks of the kernels: same tokens, same log-probs, same KV cache contents."""
import numpy as np
import pytest
import torch

from areal_vllm_amd.inference.model_runner import ModelRunner
from areal_vllm_amd.models.config import preset
from areal_vllm_amd.models.transformer import Transformer
from areal_vllm_amd.ops import decode_gemm as dg
from areal_vllm_amd.ops.attention import PAGE_SIZE


def _plan(B, ctx, steps, V, seed):
    rng = np.random.RandomState(seed)
    pages_per_seq = (ctx + steps + PAGE_SIZE - 1) // PAGE_SIZE + 1
    bt = np.zeros((B, pages_per_seq), dtype=np.int32)
    for b in range(B):
        bt[b] = b * pages_per_seq + np.arange(pages_per_seq)
    lens = rng.randint(ctx // 2, ctx + 1, B).astype(np.int32)
    pos = lens - 1
    slots = bt[np.arange(B), pos // PAGE_SIZE].astype(np.int64) * PAGE_SIZE + pos % PAGE_SIZE
    return dict(req_ids=list(range(B)), n_steps=steps, input_ids=rng.randint(2, V, B).astype(np.int32), positions=pos,
                context_lens=lens, slot_mapping=slots, block_tables=bt, max_pages=pages_per_seq,
                temperature=np.zeros(B, dtype=np.float32), top_p=np.ones(B, dtype=np.float32),
                top_k=np.full(B, 1 << 30, dtype=np.int64))


@pytest.mark.parametrize("B,name", [(48, "tiny"), (70, "tiny"), (48, "tiny-qknorm")])
def test_fused_decode_matches_plain(B, name):
    """`tiny-qknorm`: per-head q / k RMSNorm (Qwen3) runs inside the RoPE + cache-write row kernel."""
    torch.manual_seed(0)
    cfg = preset(name)
    model = Transformer(cfg, dtype=torch.float32, device="cpu")
    model.requires_grad_(False)
    model.eval()
    model.flatten()
    model.init_weights(seed=3)
    for blk in model.layers:  # non-zero projection biases, non-trivial norm weights
        if blk.attn.qkv_bias is not None:
            blk.attn.qkv_bias.data.normal_(0, 0.1)
        if cfg.qk_norm:
            blk.attn.q_norm.data.uniform_(0.5, 1.5)
            blk.attn.k_norm.data.uniform_(0.5, 1.5)
    steps, ctx = 3, 70
    pages = B * ((ctx + steps + PAGE_SIZE - 1) // PAGE_SIZE + 1) + 4
    outs = []
    for fused in (False, True):
        r = ModelRunner(model, pages, ctx + 64, B, decode_steps=steps, use_graph=False, buckets=[B],
                        fused_decode=fused)
        assert r.fused_decode == fused
        g = torch.Generator().manual_seed(5)
        for kc, vc in r.kv:
            kc.copy_(torch.randn(kc.shape, generator=g) * 0.5)
            vc.copy_(torch.randn(vc.shape, generator=g) * 0.5)
        toks, lps = r.decode(_plan(B, ctx, steps, cfg.vocab_size, seed=7))
        outs.append((toks, lps, [(k.clone(), v.clone()) for k, v in r.kv]))
    (t0, l0, kv0), (t1, l1, kv1) = outs
    assert (t0 == t1).all()
    np.testing.assert_allclose(l0, l1, atol=2e-4)
    for (k0, v0), (k1, v1) in zip(kv0, kv1):
        torch.testing.assert_close(k0, k1, atol=1e-4, rtol=1e-4)
        torch.testing.assert_close(v0, v1, atol=1e-4, rtol=1e-4)


def test_pick_config_is_launchable():
    """Every heuristic choice respects the kernel's constraints: LDS budget, loader waves dividing the staging
    pieces, split counts that leave no empty split."""
    for M in (48, 64, 100, 128, 200, 256, 300, 384, 512):
        for N, K, epi in ((2048, 1536, dg.EPI_SLAB), (1536, 1536, dg.EPI_SLAB), (17920, 1536, dg.EPI_SWIGLU),
                          (1536, 8960, dg.EPI_SLAB), (4608, 3584, dg.EPI_SLAB), (37888, 3584, dg.EPI_SWIGLU),
                          (3584, 18944, dg.EPI_SLAB), (151936, 1536, dg.EPI_BF16)):
            cfg, stages, ks, lw, nt = dg.pick_config(M, N, K, epi)
            bm, bn, _ = dg.DG_CONFIGS[cfg]
            assert stages * dg.stage_bytes(cfg) <= 160 * 1024 and 2 <= stages <= 4
            assert lw == 0 or ((bm + bn) // 8) % lw == 0
            assert dg.valid_ksplit(K, ks) == ks and (epi == dg.EPI_SLAB or ks == 1)


def test_splitk_slab_roundtrip_cpu():
    x, w = torch.randn(50, 256), torch.randn(96, 256)
    slab = dg.dgemm(x, w, epilogue=dg.EPI_SLAB, cfg=8, stages=3, ksplit=4)
    assert slab.shape == (4, 50, 96)
    torch.testing.assert_close(slab.sum(0), x @ w.t(), atol=1e-4, rtol=1e-4)
    res = torch.randn(50, 96)
    y, r = dg.splitk_residual_rmsnorm(slab, 4, 50, 96, residual=res, weight=torch.full((96,), 2.0), eps=1e-6,
                                      dtype=torch.float32)
    s = x @ w.t() + res
    torch.testing.assert_close(r, s, atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(y, 2.0 * s * torch.rsqrt(s.pow(2).mean(-1, keepdim=True) + 1e-6), atol=1e-4, rtol=1e-4)
