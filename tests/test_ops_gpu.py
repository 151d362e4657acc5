"""Numerics of every HIP kernel against a plain PyTorch fp32 reference of the same op (GPU only).

Mirrors the reference's native-extension tests (realhf/tests/cpp_extensions/test_cugae.py,
test_interval_ops.py) and extends them to the kernels the reference borrows from external libraries.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import areal_vllm_amd.ops as ops
from areal_vllm_amd.ops import attention as A
from areal_vllm_amd.ops import logprob as LP
from areal_vllm_amd.ops import norm_act as NA
from areal_vllm_amd.ops import optim as OP
from areal_vllm_amd.ops import rl as RL
from areal_vllm_amd.ops import sampling as SM

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _close(a, b, atol, rtol, name="", hard=4.0):
    """At most 0.1 % of the elements may miss the tolerance (bf16 rounding ties), and NO element may miss it by
    more than `hard` times: a localised bug (a wrong row, a wrong tile) fails however small its share is."""
    a, b = a.float(), b.float()
    assert torch.isfinite(a).all(), f"{name}: non-finite values"
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = (err > tol).float().mean().item()
    worst = (err / tol).max().item()
    assert bad < 1e-3, f"{name}: max err {err.max().item():.4g}, frac bad {bad:.4g}"
    assert worst <= hard, f"{name}: an element is off by {worst:.1f}x its tolerance (max err {err.max().item():.4g})"


def test_native_library_loaded():
    assert ops.hip_available()
    with open("/proc/self/maps") as f:
        assert "libareal_hip.so" in f.read()


@pytest.mark.parametrize("T,H", [(7, 1536), (1024, 1536), (333, 3584), (65, 5120), (40, 128)])
@pytest.mark.parametrize("with_res", [False, True])
def test_rmsnorm_fwd_bwd(T, H, with_res):
    torch.manual_seed(0)
    x = torch.randn(T, H, device=DEV).bfloat16().requires_grad_(True)
    r = torch.randn(T, H, device=DEV).bfloat16().requires_grad_(True) if with_res else None
    w = (1 + 0.1 * torch.randn(H, device=DEV)).bfloat16().requires_grad_(True)
    y, s = ops.rmsnorm(x, w, 1e-6, r)
    gy = torch.randn_like(y)
    gs = torch.randn_like(s)
    if with_res:
        (y.float() * gy.float()).sum().add((s.float() * gs.float()).sum()).backward()
    else:
        (y.float() * gy.float()).sum().backward()
    # fp32 reference
    xf = x.detach().float().requires_grad_(True)
    rf = r.detach().float().requires_grad_(True) if with_res else None
    wf = w.detach().float().requires_grad_(True)
    sf = (xf + rf).detach().bfloat16().float() + ((xf + rf) - (xf + rf).detach()) if with_res else xf
    yf = sf * torch.rsqrt(sf.pow(2).mean(-1, keepdim=True) + 1e-6) * wf
    loss = (yf * gy.float()).sum()
    if with_res:
        loss = loss + (sf * gs.float()).sum()
    loss.backward()
    _close(y, yf, 2e-2, 2e-2, "y")
    _close(x.grad, xf.grad, 3e-2, 3e-2, "dx")
    _close(w.grad, wf.grad, 0.05 + 0.02 * math.sqrt(T), 2e-2, "dw")
    if with_res:
        _close(r.grad, rf.grad, 3e-2, 3e-2, "dres")


@pytest.mark.parametrize("T,F", [(5, 8960), (1000, 8960), (77, 256)])
@pytest.mark.parametrize("act", ["silu", "gelu_tanh"])
def test_swiglu(T, F, act):
    torch.manual_seed(1)
    gu = (3.0 * torch.randn(T, 2 * F, device=DEV)).bfloat16().requires_grad_(True)
    out = ops.swiglu(gu, act)
    g = torch.randn_like(out)
    out.backward(g)
    guf = gu.detach().float().requires_grad_(True)
    fn = torch.nn.functional.silu if act == "silu" else (lambda x: torch.nn.functional.gelu(x, approximate="tanh"))
    ref = fn(guf[:, :F]) * guf[:, F:]
    ref.backward(g.float())
    _close(out, ref, 1e-2, 1e-2, "swiglu")
    _close(gu.grad, guf.grad, 2e-2, 2e-2, "dswiglu")


@pytest.mark.parametrize("D", [128, 64])
def test_rope_roundtrip_and_ref(D):
    torch.manual_seed(2)
    T, Hq, Hkv = 300, 12, 2
    tab = ops.RotaryTable(D, 4096, 1e6, DEV)
    x = torch.randn(T, Hq + 2 * Hkv, D, device=DEV).bfloat16()
    pos = torch.randint(0, 4096, (T,), device=DEV, dtype=torch.int32)
    ref = NA._rope_ref(x.clone().cpu(), pos.cpu(), ops.RotaryTable(D, 4096, 1e6, "cpu"), Hq + Hkv, 1.0)
    y = x.clone()
    ops.rope_inplace_(y, pos, tab, Hq + Hkv, 1.0)
    _close(y.cpu(), ref, 2e-2, 2e-2, "rope")
    assert torch.equal(y[:, Hq + Hkv:], x[:, Hq + Hkv:])
    ops.rope_inplace_(y, pos, tab, Hq + Hkv, -1.0)
    _close(y, x, 3e-2, 3e-2, "rope inverse")


def _rand_seqlens(B, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi + 1, (B,), generator=g).tolist()


@pytest.mark.parametrize("Hq,Hkv,lens", [
    (12, 2, [1, 33, 128, 200, 65]),
    (4, 4, [257, 64]),
    (28, 4, [700, 1, 129]),
    (8, 1, [1024]),
])
@pytest.mark.parametrize("D", [128, 256])
def test_flash_attn_fwd_bwd(Hq, Hkv, lens, D):
    torch.manual_seed(3)
    T = sum(lens)
    cu = torch.tensor([0] + np.cumsum(lens).tolist(), dtype=torch.int32, device=DEV)
    qkv = (0.5 * torch.randn(T, Hq + 2 * Hkv, D, device=DEV)).bfloat16().requires_grad_(True)
    out = ops.flash_attn_varlen_qkv(qkv, cu, max(lens), Hq, Hkv)
    g = torch.randn_like(out)
    out.backward(g)
    qf = qkv.detach().float().cpu().requires_grad_(True)
    ref, _ = A.attention_varlen_ref(qf[:, :Hq], qf[:, Hq:Hq + Hkv], qf[:, Hq + Hkv:], cu.cpu(), cu.cpu(),
                                    1.0 / math.sqrt(D), True)
    ref.backward(g.float().cpu())
    _close(out.cpu(), ref, 2e-2, 2e-2, "attn out")
    _close(qkv.grad[:, :Hq].cpu(), qf.grad[:, :Hq], 3e-2, 5e-2, "dq")
    _close(qkv.grad[:, Hq:Hq + Hkv].cpu(), qf.grad[:, Hq:Hq + Hkv], 3e-2, 5e-2, "dk")
    _close(qkv.grad[:, Hq + Hkv:].cpu(), qf.grad[:, Hq + Hkv:], 3e-2, 5e-2, "dv")


@pytest.mark.parametrize("D,Hq,Hkv", [(64, 8, 2), (256, 4, 1), (80, 4, 4), (96, 6, 2), (192, 4, 2), (320, 2, 1)])
def test_attention_head_sizes_other_than_128(D, Hq, Hkv):
    """Head size 256 has MFMA kernels of its own, other multiples of 8 up to 256 run zero-padded on the 128 / 256
    kernels, larger ones take PyTorch's fused attention per sequence; forward, backward and the bottom-right
    aligned mask of the separate-stream entries against the fp32 reference."""
    torch.manual_seed(5)
    lens = [33, 1, 130, 64]
    T = sum(lens)
    cu = torch.tensor([0] + np.cumsum(lens).tolist(), dtype=torch.int32, device=DEV)
    qkv = (0.5 * torch.randn(T, Hq + 2 * Hkv, D, device=DEV)).bfloat16().requires_grad_(True)
    out = ops.flash_attn_varlen_qkv(qkv, cu, max(lens), Hq, Hkv)
    g = torch.randn_like(out)
    out.backward(g)
    qf = qkv.detach().float().cpu().requires_grad_(True)
    ref, _ = A.attention_varlen_ref(qf[:, :Hq], qf[:, Hq:Hq + Hkv], qf[:, Hq + Hkv:], cu.cpu(), cu.cpu(),
                                    1.0 / math.sqrt(D), True)
    ref.backward(g.float().cpu())
    _close(out.cpu(), ref, 2e-2, 2e-2, "attn out")
    _close(qkv.grad.cpu(), qf.grad, 3e-2, 5e-2, "dqkv")
    lq = [5, 1, 40, 64]
    cq = torch.tensor([0] + np.cumsum(lq).tolist(), dtype=torch.int32, device=DEV)
    q = (0.5 * torch.randn(sum(lq), Hq, D, device=DEV)).bfloat16()
    k, v = qkv.detach()[:, Hq:Hq + Hkv], qkv.detach()[:, Hq + Hkv:]
    got, _ = ops.flash_attn_varlen(q, k, v, cq, cu, max(lq))
    want, _ = A.attention_varlen_ref(q.float().cpu(), k.float().cpu(), v.float().cpu(), cq.cpu(), cu.cpu(),
                                     1.0 / math.sqrt(D), True)
    _close(got.cpu(), want, 2e-2, 2e-2, "prefix-extend out")
    q2 = q.clone().requires_grad_(True)
    kv2 = qkv.detach()[:, Hq:].clone().requires_grad_(True)
    o2 = ops.flash_attn_varlen_sep(q2, kv2[:, :Hkv], kv2[:, Hkv:], cq, cu, max(lq), max(lens))
    g2 = torch.randn_like(o2)
    o2.backward(g2)
    qf2 = q.float().cpu().requires_grad_(True)
    kvf2 = kv2.detach().float().cpu().requires_grad_(True)
    r2, _ = A.attention_varlen_ref(qf2, kvf2[:, :Hkv], kvf2[:, Hkv:], cq.cpu(), cu.cpu(), 1.0 / math.sqrt(D), True)
    r2.backward(g2.float().cpu())
    _close(o2.cpu(), r2, 2e-2, 2e-2, "sep out")
    _close(q2.grad.cpu(), qf2.grad, 3e-2, 5e-2, "sep dq")
    _close(kv2.grad.cpu(), kvf2.grad, 3e-2, 5e-2, "sep dkv")


def test_flash_attn_prefix_extend():
    """q shorter than k (cached prefix): causal mask aligned bottom-right."""
    torch.manual_seed(4)
    D, Hq, Hkv = 128, 12, 2
    lq, lk = [5, 70, 1], [100, 70, 300]
    cu_q = torch.tensor([0] + np.cumsum(lq).tolist(), dtype=torch.int32, device=DEV)
    cu_k = torch.tensor([0] + np.cumsum(lk).tolist(), dtype=torch.int32, device=DEV)
    q = torch.randn(sum(lq), Hq, D, device=DEV).bfloat16() * 0.5
    k = torch.randn(sum(lk), Hkv, D, device=DEV).bfloat16() * 0.5
    v = torch.randn(sum(lk), Hkv, D, device=DEV).bfloat16()
    out, lse = ops.flash_attn_varlen(q, k, v, cu_q, cu_k, max(lq))
    ref, lse_ref = A.attention_varlen_ref(q.cpu(), k.cpu(), v.cpu(), cu_q.cpu(), cu_k.cpu(), 1 / math.sqrt(D), True)
    _close(out.cpu(), ref, 2e-2, 2e-2, "extend out")
    _close(lse.cpu(), lse_ref, 2e-2, 1e-2, "lse")


def test_flash_attn_separate_streams_fwd_bwd():
    """q chunk vs longer key ranges out of a shared K/V stream (context parallelism): strided views,
    bottom-right causal mask, gradients for q and for the used slice of k / v."""
    torch.manual_seed(14)
    D, Hq, Hkv = 128, 12, 2
    lq, lk = [130, 257, 64], [300, 257, 200]
    cu_q = torch.tensor([0] + np.cumsum(lq).tolist(), dtype=torch.int32, device=DEV)
    cu_k = torch.tensor([0] + np.cumsum(lk).tolist(), dtype=torch.int32, device=DEV)
    qkv = (0.5 * torch.randn(sum(lq), Hq + 2 * Hkv, D, device=DEV)).bfloat16().requires_grad_(True)
    kv = (0.5 * torch.randn(sum(lk) + 40, 2 * Hkv, D, device=DEV)).bfloat16().requires_grad_(True)
    q, k, v = qkv[:, :Hq], kv[8:8 + sum(lk), :Hkv], kv[8:8 + sum(lk), Hkv:]
    out = ops.flash_attn_varlen_sep(q, k, v, cu_q, cu_k, max(lq), max(lk))
    g = torch.randn_like(out)
    out.backward(g)
    qf = qkv.detach().float().cpu().requires_grad_(True)
    kvf = kv.detach().float().cpu().requires_grad_(True)
    ref, _ = A.attention_varlen_ref(qf[:, :Hq], kvf[8:8 + sum(lk), :Hkv], kvf[8:8 + sum(lk), Hkv:], cu_q.cpu(),
                                    cu_k.cpu(), 1.0 / math.sqrt(D), True)
    ref.backward(g.float().cpu())
    _close(out.cpu(), ref, 2e-2, 2e-2, "sep out")
    _close(qkv.grad[:, :Hq].cpu(), qf.grad[:, :Hq], 3e-2, 5e-2, "sep dq")
    _close(kv.grad.cpu(), kvf.grad, 3e-2, 5e-2, "sep dkv")
    assert float(qkv.grad[:, Hq:].abs().max()) == 0.0 and float(kv.grad[:8].abs().max()) == 0.0


def _blockwise_attention_ref(q, k, v, lens_q, lens_k, g, block=2048):
    """fp32 causal attention (bottom-right aligned) and its gradients for long packed sequences, on the GPU.
    Query blocks of `block` rows bound the score matrix to [Hq, block, L_k]; each block backpropagates on its
    own, the gradients accumulate in the fp32 leaves. -> out, dq, dk, dv (all fp32)."""
    Hq, Hkv, D = q.shape[1], k.shape[1], q.shape[2]
    rep = Hq // Hkv
    qf, kf, vf = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    out = torch.empty_like(qf)
    sq = sk = 0
    for lq, lk in zip(lens_q, lens_k):
        kk = kf[sk:sk + lk].transpose(0, 1).repeat_interleave(rep, 0)  # [Hq, lk, D]
        vv = vf[sk:sk + lk].transpose(0, 1).repeat_interleave(rep, 0)
        for b0 in range(0, lq, block):
            b1 = min(lq, b0 + block)
            qq = qf[sq + b0:sq + b1].transpose(0, 1)  # [Hq, rows, D]
            last = lk - lq + b1  # keys visible to the block's last row
            scores = torch.matmul(qq, kk[:, :last].transpose(1, 2)) / math.sqrt(D)
            rows = torch.arange(b0, b1, device=q.device).unsqueeze(1) + (lk - lq)
            cols = torch.arange(last, device=q.device).unsqueeze(0)
            scores = scores.masked_fill(cols > rows, float("-inf"))
            o = torch.matmul(torch.softmax(scores, -1), vv[:, :last]).transpose(0, 1)
            out[sq + b0:sq + b1] = o.detach()
            (o * g[sq + b0:sq + b1].float()).sum().backward()
        sq, sk = sq + lq, sk + lk
    return out, qf.grad, kf.grad, vf.grad


def _close_long(a, b, name):
    """Long-context bound: bf16 outputs of O(1) values - every element within 0.03 absolute / 3 % relative of the
    fp32 reference, plus the usual <0.1 % allowance at half that tolerance."""
    a, b = a.float(), b.float()
    assert torch.isfinite(a).all(), f"{name}: non-finite values"
    err = (a - b).abs()
    worst = (err / (0.03 + 0.03 * b.abs())).max().item()
    assert worst <= 1.0, f"{name}: max error {err.max().item():.4g} exceeds the hard bound ({worst:.2f}x)"
    bad = (err > 0.015 + 0.015 * b.abs()).float().mean().item()
    assert bad < 1e-3, f"{name}: {bad:.3g} of the elements miss the tight tolerance"


@pytest.mark.parametrize("lens", [[4096], [8192, 300], [16384], [32768], [5000, 12000, 1]])
def test_flash_attn_long_context_fwd_bwd(lens):
    """Packed self-attention at 4k-32k tokens per sequence (the long-generation configurations), forward and
    backward, against the blockwise fp32 reference with a hard per-element bound."""
    torch.manual_seed(21)
    D, Hq, Hkv = 128, 4, 2
    T = sum(lens)
    cu = torch.tensor([0] + np.cumsum(lens).tolist(), dtype=torch.int32, device=DEV)
    qkv = (0.5 * torch.randn(T, Hq + 2 * Hkv, D, device=DEV)).bfloat16().requires_grad_(True)
    out = ops.flash_attn_varlen_qkv(qkv, cu, max(lens), Hq, Hkv)
    g = torch.randn_like(out)
    out.backward(g)
    ref, dq, dk, dv = _blockwise_attention_ref(qkv[:, :Hq], qkv[:, Hq:Hq + Hkv], qkv[:, Hq + Hkv:], lens, lens, g)
    _close_long(out, ref, f"out {lens}")
    _close_long(qkv.grad[:, :Hq], dq, f"dq {lens}")
    _close_long(qkv.grad[:, Hq:Hq + Hkv], dk, f"dk {lens}")
    _close_long(qkv.grad[:, Hq + Hkv:], dv, f"dv {lens}")


@pytest.mark.parametrize("lens", [[4096], [2500, 700, 129], [8192, 300]])
def test_flash_attn_head_size_256_long_context_fwd_bwd(lens):
    """The head-size-256 kernels (Gemma) on multi-thousand-token packed sequences, GQA, against the blockwise fp32
    reference with the hard per-element bound."""
    torch.manual_seed(23)
    D, Hq, Hkv = 256, 4, 2
    T = sum(lens)
    cu = torch.tensor([0] + np.cumsum(lens).tolist(), dtype=torch.int32, device=DEV)
    qkv = (0.35 * torch.randn(T, Hq + 2 * Hkv, D, device=DEV)).bfloat16().requires_grad_(True)
    out = ops.flash_attn_varlen_qkv(qkv, cu, max(lens), Hq, Hkv)
    g = torch.randn_like(out)
    out.backward(g)
    ref, dq, dk, dv = _blockwise_attention_ref(qkv[:, :Hq], qkv[:, Hq:Hq + Hkv], qkv[:, Hq + Hkv:], lens, lens, g)
    _close_long(out, ref, f"out {lens}")
    _close_long(qkv.grad[:, :Hq], dq, f"dq {lens}")
    _close_long(qkv.grad[:, Hq:Hq + Hkv], dk, f"dk {lens}")
    _close_long(qkv.grad[:, Hq + Hkv:], dv, f"dv {lens}")


@pytest.mark.parametrize("lq,lk", [([1024], [16384]), ([4096, 100], [32768, 8292]), ([8192], [8192])])
def test_flash_attn_long_context_q_shorter_than_k_bwd(lq, lk):
    """A query chunk against a longer key range (context-parallel ring steps, chunked prefill) at long context:
    bottom-right aligned causal mask, gradients of q, k and v."""
    torch.manual_seed(22)
    D, Hq, Hkv = 128, 4, 2
    cu_q = torch.tensor([0] + np.cumsum(lq).tolist(), dtype=torch.int32, device=DEV)
    cu_k = torch.tensor([0] + np.cumsum(lk).tolist(), dtype=torch.int32, device=DEV)
    q = (0.5 * torch.randn(sum(lq), Hq, D, device=DEV)).bfloat16().requires_grad_(True)
    k = (0.5 * torch.randn(sum(lk), Hkv, D, device=DEV)).bfloat16().requires_grad_(True)
    v = (0.5 * torch.randn(sum(lk), Hkv, D, device=DEV)).bfloat16().requires_grad_(True)
    out = ops.flash_attn_varlen_sep(q, k, v, cu_q, cu_k, max(lq), max(lk))
    g = torch.randn_like(out)
    out.backward(g)
    ref, dq, dk, dv = _blockwise_attention_ref(q, k, v, lq, lk, g)
    _close_long(out, ref, "out")
    _close_long(q.grad, dq, "dq")
    _close_long(k.grad, dk, "dk")
    _close_long(v.grad, dv, "dv")


@pytest.mark.parametrize("Hq,Hkv,D", [(12, 2, 128), (28, 4, 128), (8, 8, 128), (16, 2, 64), (8, 1, 256), (16, 16, 256)])
@pytest.mark.parametrize("nsplit", [1, 4])
def test_paged_decode_and_cache_write(Hq, Hkv, D, nsplit):
    torch.manual_seed(5)
    B = 9
    ctx = [1, 31, 32, 33, 64, 100, 257, 500, 77]
    max_pages = (max(ctx) + 31) // 32
    num_pages = 1 + sum((c + 31) // 32 for c in ctx)
    kc, vc = ops.alloc_kv_cache(num_pages, Hkv, D, torch.bfloat16, DEV)
    tab = ops.RotaryTable(D, 2048, 1e4, DEV)
    bt = torch.zeros(B, max_pages, dtype=torch.int32, device=DEV)
    nxt = 1
    ks, vs = [], []
    for b, c in enumerate(ctx):
        n = (c + 31) // 32
        pages = torch.arange(nxt, nxt + n, dtype=torch.int32)
        nxt += n
        bt[b, :n] = pages.to(DEV)
        qkv = torch.randn(c, Hq + 2 * Hkv, D, device=DEV).bfloat16()
        pos = torch.arange(c, dtype=torch.int32, device=DEV)
        slots = (pages.long()[pos.cpu().long() // 32] * 32 + pos.cpu().long() % 32).to(DEV)
        ref_qkv = NA._rope_ref(qkv.clone().cpu(), pos.cpu(), ops.RotaryTable(D, 2048, 1e4, "cpu"), Hq + Hkv, 1.0)
        ops.rope_and_cache_write_(qkv, pos, slots, tab, kc, vc, Hq, Hkv)
        _close(qkv.cpu(), ref_qkv, 2e-2, 2e-2, "rope in cache write")
        ks.append(qkv[:, Hq:Hq + Hkv].clone())
        vs.append(qkv[:, Hq + Hkv:].clone())
        kg, vg = ops.cache_gather(kc, vc, slots)
        assert torch.equal(kg, ks[-1]) and torch.equal(vg, vs[-1])
    q = torch.randn(B, Hq, D, device=DEV).bfloat16()
    cl = torch.tensor(ctx, dtype=torch.int32, device=DEV)
    out = ops.paged_attention_decode(q, kc, vc, bt, cl, nsplit=nsplit)
    ref = A.paged_decode_ref(q.cpu(), kc.cpu(), vc.cpu(), bt.cpu(), cl.cpu(), 1 / math.sqrt(D))
    _close(out.cpu(), ref, 2e-2, 2e-2, "paged decode")


@pytest.mark.parametrize("Hq,Hkv,D", [(12, 2, 128), (28, 4, 128), (16, 2, 64), (8, 1, 256)])
@pytest.mark.parametrize("nsplit", [1, 4])
def test_fp8_kv_cache_writes_round_like_torch_and_decode_reads_them(Hq, Hkv, D, nsplit):
    """KV cache of e4m3 bytes: both cache writers (the stand-alone rotate-and-write kernel, the split-K row kernel)
    store exactly ``clamp(x, +-448).to(float8_e4m3fn)`` of the bf16 keys / values, the gather widens them back
    exactly, and decode attention equals an fp32 reference over the widened cache (values beyond +-448 saturate)."""
    torch.manual_seed(5)
    B = 7
    ctx = [1, 32, 33, 100, 257, 500, 77]
    max_pages = (max(ctx) + 31) // 32
    num_pages = 1 + sum((c + 31) // 32 for c in ctx)
    kc, vc = ops.alloc_kv_cache(num_pages, Hkv, D, A.FP8, DEV)
    kb, vb = ops.alloc_kv_cache(num_pages, Hkv, D, torch.bfloat16, DEV)
    assert kc.element_size() == 1 and kc.shape == kb.shape
    tab = ops.RotaryTable(D, 2048, 1e4, DEV)
    bt = torch.zeros(B, max_pages, dtype=torch.int32, device=DEV)
    nxt = 1
    for b, c in enumerate(ctx):
        n = (c + 31) // 32
        pages = torch.arange(nxt, nxt + n, dtype=torch.int32)
        nxt += n
        bt[b, :n] = pages.to(DEV)
        qkv = (3 * torch.randn(c, Hq + 2 * Hkv, D, device=DEV)).bfloat16()
        qkv[0, Hq, :4] = torch.tensor([1000.0, -1000.0, 448.0, 1e-3], device=DEV).bfloat16()  # saturation, tiny values
        pos = torch.arange(c, dtype=torch.int32, device=DEV)
        slots = (pages.long()[pos.cpu().long() // 32] * 32 + pos.cpu().long() % 32).to(DEV)
        q2 = qkv.clone()
        ops.rope_and_cache_write_(qkv, pos, slots, tab, kc, vc, Hq, Hkv, do_rope=b % 2 == 0)
        ops.rope_and_cache_write_(q2, pos, slots, tab, kb, vb, Hq, Hkv, do_rope=b % 2 == 0)
        assert torch.equal(qkv, q2)  # the rotated q / k rows themselves stay bf16
        kg, vg = ops.cache_gather(kc, vc, slots)
        assert kg.dtype == torch.bfloat16
        assert torch.equal(kg, A.to_fp8(qkv[:, Hq:Hq + Hkv]).to(torch.bfloat16))
        assert torch.equal(vg, A.to_fp8(qkv[:, Hq + Hkv:]).to(torch.bfloat16))
    # the whole cache is what torch's conversion makes of the bf16 cache
    assert torch.equal(kc.view(torch.uint8), A.to_fp8(kb).view(torch.uint8))
    assert torch.equal(vc.view(torch.uint8), A.to_fp8(vb).view(torch.uint8))
    q = torch.randn(B, Hq, D, device=DEV).bfloat16()
    cl = torch.tensor(ctx, dtype=torch.int32, device=DEV)
    out = ops.paged_attention_decode(q, kc, vc, bt, cl, nsplit=nsplit)
    ref = A.paged_decode_ref(q.cpu(), kc.cpu(), vc.cpu(), bt.cpu(), cl.cpu(), 1 / math.sqrt(D))
    _close(out.cpu(), ref, 2e-2, 2e-2, "paged decode over an fp8 cache")
    # prefill over cached prefixes: the paged kernel widens the pages itself (head size 128; others gather first)
    rows = [5, 3, 4]  # requests 5, 3, 4 of the batch above with 5 / 40 / 130 query rows at the end of their contexts
    nq = [5, 40, 130]
    cu = torch.tensor([0, 5, 45, 175], dtype=torch.int32, device=DEV)
    cuk = torch.tensor([0] + list(np.cumsum([ctx[r] for r in rows])), dtype=torch.int32, device=DEV)
    qp = torch.randn(175, Hq, D, device=DEV).bfloat16()
    btp = bt[rows].contiguous()
    o8 = ops.flash_attn_paged_prefill(qp, kc, vc, btp, cu, cuk, max(nq))
    kg, vg = ops.cache_gather(kc, vc, A.context_slots(btp, cuk))
    ow, _ = A.attention_varlen_ref(qp.float(), kg.float(), vg.float(), cu, cuk, 1 / math.sqrt(D), True)
    _close(o8, ow, 2e-2, 2e-2, "paged prefill over an fp8 cache")


@pytest.mark.parametrize("Hq,Hkv,D", [(12, 2, 128), (16, 2, 64)])
def test_splitk_row_kernel_writes_an_fp8_cache(Hq, Hkv, D):
    torch.manual_seed(2)
    T, S, P = 75, 4, 40
    Nq = (Hq + 2 * Hkv) * D
    slab = torch.randn(S, T, Nq, device=DEV)
    bias = torch.randn(Nq, device=DEV).bfloat16()
    pos = torch.randint(0, 900, (T,), device=DEV, dtype=torch.int32)
    slots = torch.randperm(P * 32, device=DEV)[:T].long()
    slots[3] = -1
    table = NA.RotaryTable(D, 1024, 1e4, DEV)
    kc, vc = A.alloc_kv_cache(P, Hkv, D, A.FP8, DEV)
    kb, vb = A.alloc_kv_cache(P, Hkv, D, torch.bfloat16, DEV)
    q = DG.splitk_rope_cache_write(slab, S, bias, pos, slots, table, kc, vc, Hq, Hkv, D)
    qb = DG.splitk_rope_cache_write(slab, S, bias, pos, slots, table, kb, vb, Hq, Hkv, D)
    assert torch.equal(q, qb)
    assert torch.equal(kc.view(torch.uint8), A.to_fp8(kb).view(torch.uint8))
    assert torch.equal(vc.view(torch.uint8), A.to_fp8(vb).view(torch.uint8))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("V", [151936, 1000, 517])
def test_logprob_entropy_fwd_bwd(dtype, V):
    torch.manual_seed(6)
    T = 37
    Vp = (V + 7) // 8 * 8
    base = (3 * torch.randn(T, Vp, device=DEV)).to(dtype)
    logits = base[:, :V].detach().requires_grad_(True)
    labels = torch.randint(0, V, (T,), device=DEV)
    temp = 0.7
    logp, ent = ops.gather_logprobs_entropy(logits, labels, temp)
    gl, ge = torch.randn(T, device=DEV), torch.randn(T, device=DEV)
    ((logp * gl).sum() + (ent * ge).sum()).backward()
    lf = logits.detach().float().requires_grad_(True)
    ls = torch.log_softmax(lf / temp, -1)
    ref_lp = ls.gather(-1, labels.unsqueeze(-1)).squeeze(-1)
    ref_ent = -(ls.exp() * ls).sum(-1)
    ((ref_lp * gl).sum() + (ref_ent * ge).sum()).backward()
    _close(logp, ref_lp, 1e-3, 1e-4, "logp")
    _close(ent, ref_ent, 1e-3, 1e-3, "entropy")
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-4
    _close(logits.grad, lf.grad, tol * lf.grad.abs().max().item(), 2e-2, "dlogits")


def test_lm_head_logprobs_matches_unfused():
    torch.manual_seed(7)
    T, H, V = 300, 256, 4096
    h = torch.randn(T, H, device=DEV).bfloat16().requires_grad_(True)
    w = (0.05 * torch.randn(V, H, device=DEV)).bfloat16().requires_grad_(True)
    labels = torch.randint(0, V, (T,), device=DEV)
    lp, ent = ops.lm_head_logprobs(h, w, labels, 1.0, chunk=128)
    g = torch.randn(T, device=DEV)
    (lp * g).sum().backward()
    hf, wf = h.detach().float().requires_grad_(True), w.detach().float().requires_grad_(True)
    lg = (hf @ wf.t()).detach().bfloat16().float() + ((hf @ wf.t()) - (hf @ wf.t()).detach())
    ls = torch.log_softmax(lg, -1)
    ref = ls.gather(-1, labels.unsqueeze(-1)).squeeze(-1)
    (ref * g).sum().backward()
    _close(lp, ref, 2e-3, 1e-3, "fused logp")
    _close(h.grad, hf.grad, 2e-2 * hf.grad.abs().max().item(), 5e-2, "dh")
    _close(w.grad, wf.grad, 2e-2 * wf.grad.abs().max().item(), 5e-2, "dw")


@pytest.mark.parametrize("V,tp", [(151936, 4), (4096, 2), (1000, 3)])
def test_vocab_shard_kernels_merge_to_full_vocab(V, tp):
    """The tensor-parallel loss path on one device: every vocabulary shard (vocab_start != 0, uneven last
    shard) goes through the stats kernel, the [4, T] statistics are merged as the all-gather would, and
    the gradient kernel forms each shard's d(logits) from the GLOBAL log-sum-exp. Compared with an fp32
    log-softmax over the whole vocabulary, forward and backward."""
    from areal_vllm_amd.ops import logprob as LP

    torch.manual_seed(16)
    T, temp = 53, 0.8
    full = (3 * torch.randn(T, V, device=DEV)).bfloat16()
    labels = torch.randint(0, V, (T,), device=DEV)
    labels[0], labels[1] = 0, V - 1
    gl, ge = torch.randn(T, device=DEV), torch.randn(T, device=DEV)
    lf = full.float().requires_grad_(True)
    ls = torch.log_softmax(lf / temp, -1)
    ref_lp = ls.gather(-1, labels.unsqueeze(-1)).squeeze(-1)
    ref_ent = -(ls.exp() * ls).sum(-1)
    ((ref_lp * gl).sum() + (ref_ent * ge).sum()).backward()

    bounds = [ops.vocab_shard_range(V, tp, r) for r in range(tp)]
    shards = [LP._aligned_copy(full[:, lo:hi]) for lo, hi in bounds]
    stats = torch.empty(tp, 4, T, dtype=torch.float32, device=DEV)
    for r, (lo, hi) in enumerate(bounds):
        ops.logprob_stats(shards[r], labels, temp, lo, out=stats[r])
    m = stats[:, 0].max(0).values
    f = torch.exp(stats[:, 0] - m)
    logp, ent, lse, ez = LP._finalize(m, (stats[:, 1] * f).sum(0), (stats[:, 2] * f).sum(0), stats[:, 3].sum(0))
    _close(logp, ref_lp, 1e-3, 1e-4, "merged logp")
    _close(ent, ref_ent, 1e-3, 1e-3, "merged entropy")
    for r, (lo, hi) in enumerate(bounds):
        LP._bwd_inplace(shards[r], labels, temp, lse, ez, gl, ge, lo)
        want = lf.grad[:, lo:hi]
        _close(shards[r], want, 2e-2 * lf.grad.abs().max().item(), 2e-2, f"dlogits of shard {r}")


@pytest.mark.parametrize("max_seqlen", [32, 128, 512, 5000])
@pytest.mark.parametrize("bs", [2, 4, 33])
@pytest.mark.parametrize("gamma,lam", [(0.9, 0.5), (1.0, 1.0), (0.99, 0.95)])
def test_gae_1d_packed(max_seqlen, bs, gamma, lam):
    torch.manual_seed(8)
    seqlens = torch.randint(1, max_seqlen, (bs,), dtype=torch.int32)
    total = int(seqlens.sum())
    rewards = torch.randn(total)
    values = torch.randn(total + bs)
    bootstrap = torch.rand(bs) > 0.5
    cu = torch.nn.functional.pad(seqlens.cumsum(0), (1, 0)).int()
    adv, ret = ops.gae_1d_packed(rewards.to(DEV), values.to(DEV), cu.to(DEV), bootstrap.to(DEV), gamma, lam)
    radv, rret = RL.gae_1d_packed_ref(rewards.double(), values.double(), cu, bootstrap, gamma, lam)
    scale = max(1.0, radv.abs().max().item())
    assert torch.allclose(adv.cpu().double(), radv, atol=2e-4 * scale), (adv.cpu() - radv).abs().max()
    assert torch.allclose(ret.cpu().double(), rret, atol=2e-4 * scale)


def test_gae_padded_and_2d():
    torch.manual_seed(9)
    B, S = 7, 300
    r, v = torch.randn(B, S), torch.randn(B, S)
    ne = torch.rand(B) > 0.5
    adv, ret = ops.gae_padded(r.to(DEV), v.to(DEV), ne.to(DEV), 0.99, 0.9)
    radv, rret = RL.gae_padded_ref(r, v, ne, 0.99, 0.9)
    assert torch.allclose(adv.cpu(), radv, atol=1e-3) and torch.allclose(ret.cpu(), rret, atol=1e-3)
    # segmented
    H = 200
    rewards, values = torch.randn(B, H), torch.randn(B, H + 1)
    done = torch.zeros(B, H + 1, dtype=torch.bool)
    idx, cu, flags = [], [0], []
    for b in range(B):
        ts = sorted(np.random.RandomState(b).choice(np.arange(2, H - 1), size=b % 4, replace=False).tolist())
        for t in ts:
            done[b, t] = True
            idx.append(t)
            flags.append(bool((t + b) % 2))
        cu.append(len(idx))
    idx_t = torch.tensor(idx, dtype=torch.int32)
    cu_t = torch.tensor(cu, dtype=torch.int32)
    fl_t = torch.tensor(flags, dtype=torch.bool)
    for overlapped in (True, False):
        a, rt = ops.gae_2d(rewards.to(DEV), values.to(DEV), done.to(DEV), idx_t.to(DEV), cu_t.to(DEV), fl_t.to(DEV),
                           3, 0.97, 0.8, overlapped)
        ra, rr = RL.gae_2d_ref(rewards, values, done, idx_t, cu_t, fl_t, 0 if overlapped else 1, 0.97, 0.8)
        assert torch.allclose(a.cpu(), ra, atol=1e-3), (a.cpu() - ra).abs().max()
        assert torch.allclose(rt.cpu(), rr, atol=1e-3)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_interval_ops(dtype):
    torch.manual_seed(10)
    n = 1 << 20
    flat = torch.randn(n, device=DEV).to(dtype)
    rs = np.random.RandomState(0)
    starts = np.sort(rs.choice(n - 5000, 700, replace=False))
    iv = np.stack([starts, starts + rs.randint(1, 4000, 700)], 1)
    iv = ops.merge_intervals(iv)
    packed = ops.slice_intervals(flat, iv)
    ref = torch.cat([flat[s:e] for s, e in iv])
    assert torch.equal(packed, ref)
    flat2 = torch.zeros_like(flat)
    ops.set_intervals(packed, flat2, iv)
    for s, e in iv[:50]:
        assert torch.equal(flat2[s:e], flat[s:e])
    mask = torch.zeros(n, dtype=torch.bool, device=DEV)
    for s, e in iv:
        mask[s:e] = True
    assert flat2[~mask].abs().sum() == 0


@pytest.mark.parametrize("gdtype", [torch.bfloat16, torch.float32])
def test_adamw_and_norm(gdtype):
    torch.manual_seed(11)
    n = 1_000_003
    p0 = torch.randn(n, device=DEV)
    g = torch.randn(n, device=DEV).to(gdtype)
    master, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    shadow = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    state = torch.zeros(8, device=DEV)
    ref_p = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref_p], lr=1e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
    for step in range(3):
        state[0] = 0
        ops.sumsq_(g, state)
        norm_ref = g.float().norm()
        ops.adam_prepare_(state, 1.0, 1.0, 0.9, 0.95)
        ops.adamw_(master, m, v, g, shadow, state, 1e-3, 0.9, 0.95, 1e-8, 0.1)
        assert abs(state[1].item() - norm_ref.item()) / norm_ref.item() < 1e-3
        ref_p.grad = g.float() * min(1.0, 1.0 / (norm_ref.item() + 1e-6))
        opt.step()
    _close(master, ref_p.detach(), 1e-5, 1e-4, "adam master")
    _close(shadow, ref_p.detach(), 1e-2, 1e-2, "adam shadow")
    # non-finite gradient: the update is skipped
    before = master.clone()
    state[0] = float("inf")
    ops.adam_prepare_(state, 1.0, 1.0, 0.9, 0.95)
    ops.adamw_(master, m, v, g, shadow, state, 1e-3, 0.9, 0.95, 1e-8, 0.1)
    assert torch.equal(before, master) and state[3].item() == 0.0
    x = torch.randn(1000, device=DEV)
    y = x.clone()
    ops.scale_(y, 0.5)
    assert torch.allclose(y, x * 0.5)


def test_sampling_distribution_and_filters():
    torch.manual_seed(12)
    V = 1000
    B = 512
    row = torch.randn(V, device=DEV) * 2
    logits = row.unsqueeze(0).repeat(B, 1).bfloat16()
    probs = torch.softmax(logits[0].float(), -1)
    step = torch.zeros(1, dtype=torch.int64, device=DEV)
    counts = torch.zeros(V, device=DEV)
    for it in range(40):
        tok, lp = ops.sample(logits, None, None, None, seed=123, step=step)
        counts += torch.bincount(tok, minlength=V).float()
        ref_lp = torch.log_softmax(logits[0].float(), -1)[tok]
        assert torch.allclose(lp, ref_lp, atol=1e-3)
    assert step.item() == 40
    emp = counts / counts.sum()
    assert (emp - probs).abs().max().item() < 0.02
    top = probs.topk(5).indices
    assert ((emp[top] - probs[top]).abs() / probs[top]).max().item() < 0.15
    # greedy
    temp0 = torch.zeros(B, device=DEV)
    tok, _ = ops.sample(logits, temp0, None, None, seed=1, step=step)
    assert (tok == logits[0].float().argmax()).all()
    # top-k = 3 / top-p = 0.5
    t1 = torch.ones(B, device=DEV)
    k3 = torch.full((B,), 3, dtype=torch.int32, device=DEV)
    p1 = torch.ones(B, device=DEV)
    tok, _ = ops.sample(logits, t1, k3, p1, seed=5, step=step)
    assert set(tok.tolist()) <= set(probs.topk(3).indices.tolist())
    kall = torch.full((B,), V, dtype=torch.int32, device=DEV)
    p5 = torch.full((B,), 0.5, device=DEV)
    tok, _ = ops.sample(logits, t1, kall, p5, seed=6, step=step)
    sp, si = torch.sort(probs, descending=True)
    nucleus = set(si[: int(((torch.cumsum(sp, 0) - sp) < 0.5).sum())].tolist())
    assert set(tok.tolist()) <= nucleus


@pytest.mark.parametrize("V,scale", [(151936, 2.0), (50000, 6.0), (9000, 1.0)])
def test_filtered_sampling_on_large_vocabularies(V, scale):
    """top-k / top-p thresholds found by halving over the whole row until the bracket fits LDS, then among the
    candidates: every drawn token lies in the reference's top-k set / nucleus (computed by sorting), the reported
    log-prob is the unfiltered one, and a replay of the same step draws the same tokens."""
    torch.manual_seed(17)
    B = 48
    logits = (scale * torch.randn(B, V, device=DEV)).bfloat16()
    lf = logits.float()
    probs = torch.softmax(lf, -1)
    sp, si = torch.sort(probs, -1, descending=True)
    t1 = torch.ones(B, device=DEV)
    for k, p in ((50, 1.0), (V, 0.9), (200, 0.5), (1, 1.0), (V, 0.999)):
        tk = torch.full((B,), k, dtype=torch.int32, device=DEV)
        tp = torch.full((B,), p, device=DEV)
        # reference with the kernel's treatment of ties (bf16 logits tie often): the top-k set holds every entry that
        # is >= the k-th largest value, the nucleus every entry >= the value at which the descending cumulative mass
        # first reaches p x (mass of the top-k set); one entry of slack at that boundary (summation order)
        in_k = probs >= sp[:, k - 1:k] if k < V else torch.ones_like(probs, dtype=torch.bool)
        mass_k = (probs * in_k).sum(-1, keepdim=True)
        cum = torch.cumsum(sp, -1)
        first = (cum < p * mass_k).sum(-1)  # index of the entry that crosses the target
        last = (first + (1 if p < 1.0 else 0)).clamp(max=V - 1)
        kept = in_k & (probs >= sp.gather(1, last.unsqueeze(1))) if p < 1.0 else in_k
        draws = []
        for rep in range(2):
            step = torch.full((1,), 3, dtype=torch.int64, device=DEV)
            tok, lp = ops.sample(logits, t1, tk, tp, seed=21, step=step)
            draws.append(tok.clone())
            assert kept[torch.arange(B, device=DEV), tok].all(), (k, p)
            ref = torch.log_softmax(lf, -1)[torch.arange(B, device=DEV), tok]
            assert torch.allclose(lp, ref, atol=2e-3)
        assert torch.equal(draws[0], draws[1])
        if k == 1:
            assert torch.equal(lf.gather(1, draws[0].unsqueeze(1)).squeeze(1), lf.max(-1).values)


@pytest.mark.parametrize("V", [1000, 151936, 517])
def test_sampler_penalties_and_inactive_rows(V):
    """Rows with a slot of the token-count table sample from logit - f * count - p * (count > 0) and bump the
    count of what they draw; rows flagged inactive are skipped. Greedy rows make the comparison exact."""
    torch.manual_seed(13)
    B = 12
    logits = (2 * torch.randn(B, V, device=DEV)).bfloat16()
    counts = ops.TokenCounts(8, V, DEV)
    counts.table[:, :V] = (torch.rand(8, V, device=DEV) < 0.3).short() * torch.randint(1, 5, (8, V), device=DEV).short()
    before = counts.table.clone()
    slot = torch.tensor([0, -1, 3, 7, -1, 2, 2, 5, -1, 1, 4, 6], dtype=torch.int32, device=DEV)
    slot[6] = -1
    freq = torch.rand(B, device=DEV) * 2
    pres = torch.rand(B, device=DEV)
    active = torch.ones(B, dtype=torch.int32, device=DEV)
    active[3] = 0
    active[8] = 0
    temp = torch.zeros(B, device=DEV)  # greedy
    step = torch.zeros(1, dtype=torch.int64, device=DEV)
    tok, lp = ops.sample(logits, temp, None, None, seed=3, step=step, active=active, counts=counts, count_slot=slot,
                         freq_pen=freq, pres_pen=pres)
    for b in range(B):
        if not active[b]:
            assert tok[b] == 0 and lp[b] == 0
            continue
        row = logits[b].float()
        if slot[b] >= 0:
            c = before[slot[b], :V].float()
            row = row - freq[b] * c - pres[b] * (c > 0).float()
        want = int(row.argmax())
        assert int(tok[b]) == want
        assert abs(float(lp[b]) - float(torch.log_softmax(row, -1)[want])) < 2e-3
    want_counts = before.clone()
    for b in range(B):
        if active[b] and slot[b] >= 0:
            want_counts[slot[b], tok[b]] += 1
    assert torch.equal(counts.table, want_counts)
    # sampled rows: a token with an enormous count is never drawn, its neighbours keep their relative odds
    hot = logits[0].float().argmax()
    counts.table.zero_()
    counts.table[0, hot] = 30000
    t1 = torch.ones(B, device=DEV)
    all0 = torch.zeros(B, dtype=torch.int32, device=DEV)
    big = torch.full((B,), 10.0, device=DEV)
    same = logits[0:1].repeat(B, 1)
    for _ in range(20):
        counts.table.zero_()
        counts.table[0, hot] = 30000
        tok, lp = ops.sample(same, t1, None, None, seed=9, step=step, counts=counts, count_slot=all0, freq_pen=big,
                             pres_pen=torch.zeros(B, device=DEV))
        assert (tok != hot).all()


@pytest.mark.parametrize("B,V", [(384, 151936), (7, 517), (1500, 1000), (33, 32003)])
def test_split_row_sampler_draws_the_tokens_of_the_one_workgroup_kernel(B, V):
    """Unfiltered batches are reduced by several workgroups per row: same noise per entry, so the same tokens; the
    log-prob may differ by summation order only. Rows on an odd stride take the scalar-load path in both."""
    from areal_vllm_amd.ops import sampling

    torch.manual_seed(15)
    for ld in (V, V + 3):
        buf = (2 * torch.randn(B, ld, device=DEV)).bfloat16()
        logits = buf[:, :V]
        temp = torch.ones(B, device=DEV)
        temp[::5] = 0.0  # some greedy rows
        temp[1::7] = 0.7
        active = torch.ones(B, dtype=torch.int32, device=DEV)
        active[2::9] = 0
        out = {}
        for split in (True, False):
            sampling.SPLIT_UNFILTERED = split
            step = torch.full((1,), 5, dtype=torch.int64, device=DEV)
            try:
                out[split] = ops.sample(logits, temp, None, None, seed=77, step=step, active=active)
            finally:
                sampling.SPLIT_UNFILTERED = True
            assert step.item() == 6
        assert torch.equal(out[True][0], out[False][0])
        assert torch.allclose(out[True][1], out[False][1], atol=1e-4)
        live = active.bool()
        ref = torch.log_softmax(logits.float() / temp.clamp(min=1e-9).unsqueeze(1).where(temp.unsqueeze(1) > 0, torch.ones_like(temp).unsqueeze(1)), -1)
        got = ref.gather(1, out[True][0].unsqueeze(1)).squeeze(1)
        assert torch.allclose(out[True][1][live], got[live], atol=2e-3)
        assert (out[True][0][~live] == 0).all()


@pytest.mark.parametrize("T,E,k", [(1, 4, 2), (7, 60, 4), (256, 128, 8), (300, 8, 2), (5000, 128, 8), (33, 512, 1),
                                   (40000, 64, 6)])
def test_moe_routing_kernels_match_the_library_formulation(T, E, k):
    """Fused softmax + top-k and the counting sort by expert against softmax / topk / stable argsort / bincount:
    same experts, same (stable) order, same row ranges; weights to rounding."""
    torch.manual_seed(16)
    logits = 3 * torch.randn(T, E, device=DEV)
    for norm in (True, False):
        topv, topi = ops.moe_topk(logits, k, norm)
        rv, ri = torch.topk(torch.softmax(logits, -1), k, -1)
        if norm:
            rv = rv / rv.sum(-1, keepdim=True)
        assert torch.equal(topi, ri)
        assert torch.allclose(topv, rv, rtol=2e-5, atol=1e-7)
        # bf16 logits in (read as fp32 inside), bf16 weights out (rounded once, at the end)
        lb = logits.bfloat16()
        bv, bi = ops.moe_topk(lb, k, norm, out_dtype=torch.bfloat16)
        rv2, ri2 = torch.topk(torch.softmax(lb.float(), -1), k, -1)
        if norm:
            rv2 = rv2 / rv2.sum(-1, keepdim=True)
        assert bv.dtype == torch.bfloat16
        assert torch.allclose(bv.float(), rv2.bfloat16().float(), rtol=8e-3, atol=1e-6)
        tie_free = (rv2[:, :-1] != rv2[:, 1:]).all(-1) if k > 1 else torch.ones(T, dtype=torch.bool, device=DEV)
        assert torch.equal(bi[tie_free], ri2[tie_free])
    flat_e = topi.reshape(-1).clone()
    flat_e[::11] = E  # some pairs dropped: the bin behind every expert
    order, inv, src_row, offsets = ops.moe_sort(flat_e, E, k)
    want = torch.argsort(flat_e, stable=True)
    assert torch.equal(order, want)
    assert torch.equal(inv[order], torch.arange(T * k, device=DEV))
    assert torch.equal(src_row, want // k)
    counts = torch.bincount(flat_e, minlength=E + 1)
    assert offsets.dtype == torch.int32 and offsets.numel() == E + 2
    assert torch.equal(offsets[1:].long(), torch.cumsum(counts, 0)) and int(offsets[0]) == 0


@pytest.mark.parametrize("T,hq,hkv,D", [(1000, 16, 8, 128), (37, 4, 2, 64), (513, 2, 1, 256), (4096, 32, 4, 128)])
def test_qk_norm_in_packed_qkv_matches_fp32_reference(T, hq, hkv, D):
    """Per-head RMSNorm of the q / k heads inside the packed projection output, v heads copied, forward and
    backward, against the slice / normalise / concatenate formulation in fp32."""
    torch.manual_seed(18)
    Ht = hq + 2 * hkv
    qkv = (torch.randn(T, Ht * D, device=DEV) * 1.5).bfloat16().requires_grad_(True)
    qw = (1 + 0.2 * torch.randn(D, device=DEV)).bfloat16().requires_grad_(True)
    kw = (1 + 0.2 * torch.randn(D, device=DEV)).bfloat16().requires_grad_(True)
    eps = 1e-6
    out = ops.qk_norm(qkv, qw, kw, eps, hq, hkv, D)
    dout = torch.randn_like(out)
    out.backward(dout)

    x = qkv.detach().float().view(T, Ht, D).requires_grad_(True)
    qf, kf = qw.detach().float().requires_grad_(True), kw.detach().float().requires_grad_(True)

    def norm(v, w):
        return v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + eps) * w

    ref = torch.cat([norm(x[:, :hq], qf), norm(x[:, hq:hq + hkv], kf), x[:, hq + hkv:]], 1).view(T, -1)
    ref.backward(dout.float())
    assert (out.float() - ref).abs().max().item() < 0.04
    assert torch.equal(out.view(T, Ht, D)[:, hq + hkv:], qkv.view(T, Ht, D)[:, hq + hkv:])
    assert (qkv.grad.float() - x.grad.view(T, -1)).abs().max().item() < 0.06
    scale = max(1.0, qf.grad.abs().max().item())
    assert (qw.grad.float() - qf.grad).abs().max().item() < 0.02 * scale
    assert (kw.grad.float() - kf.grad).abs().max().item() < 0.02 * max(1.0, kf.grad.abs().max().item())


def test_decode_advance_and_begin_kernels_match_their_cpu_twins():
    """Device bookkeeping of a multi-step decode plan (early exit on stop ids / budget, kill mask of a
    continuation plan) against the PyTorch twins the CPU engine runs."""
    torch.manual_seed(14)
    B, max_pages, n_stop = 300, 6, 8
    g = torch.Generator().manual_seed(1)
    st = dict(
        input_ids=torch.randint(0, 50, (B,), generator=g), positions=torch.randint(0, 150, (B,), generator=g).int(),
        block_tables=torch.randint(1, 500, (B, max_pages), generator=g).int(), active=(torch.rand(B, generator=g) < 0.8).int(),
        remaining=torch.randint(1, 4, (B,), generator=g).int(), min_remaining=torch.randint(0, 3, (B,), generator=g).int(),
        stop_ids=torch.where(torch.rand(B, n_stop, generator=g) < 0.3, torch.randint(0, 6, (B, n_stop), generator=g),
                             torch.full((B, n_stop), -1)).int())
    st["context_lens"] = st["positions"] + 1
    st["slots"] = torch.randint(0, 5000, (B,), generator=g)
    kill = (torch.rand(B, generator=g) < 0.2).int()
    cpu = {k: v.clone() for k, v in st.items()}
    dev = {k: v.clone().to(DEV) for k, v in st.items()}
    ops.decode_begin_(kill, cpu["active"], cpu["positions"], cpu["context_lens"], cpu["slots"], cpu["block_tables"])
    ops.decode_begin_(kill.to(DEV), dev["active"], dev["positions"], dev["context_lens"], dev["slots"], dev["block_tables"])
    for k in cpu:
        assert torch.equal(dev[k].cpu(), cpu[k]), f"decode_begin: {k}"
    for step in range(4):
        sampled = torch.randint(0, 6, (B,), generator=g)
        for d, smp in ((cpu, sampled), (dev, sampled.to(DEV))):
            ops.decode_advance_(smp, d["input_ids"], d["positions"], d["context_lens"], d["slots"], d["block_tables"],
                                d["active"], d["remaining"], d["min_remaining"], d["stop_ids"])
        for k in cpu:
            assert torch.equal(dev[k].cpu(), cpu[k]), f"decode_advance step {step}: {k}"
    assert 0 < int(cpu["active"].sum()) < B or int(cpu["active"].sum()) == 0


@pytest.mark.gpu
def test_linear_accumulates_wgrad_into_fp32_main_grad():
    """ops.linear: dW is summed by the GEMM into the fp32 main-grad view over two micro-batches."""
    from areal_vllm_amd import ops

    torch.manual_seed(0)
    dev = torch.device("cuda")
    w = torch.nn.Parameter((torch.randn(384, 256, device=dev) * 0.05).bfloat16())
    b = torch.nn.Parameter(torch.zeros(384, device=dev, dtype=torch.bfloat16))
    w.main_grad = torch.zeros(384, 256, device=dev, dtype=torch.float32)
    ref_w = torch.zeros(384, 256, device=dev, dtype=torch.float32)
    ref_b = torch.zeros(384, device=dev, dtype=torch.float32)
    for _ in range(2):
        x = torch.randn(512, 256, device=dev).bfloat16().requires_grad_(True)
        dy = torch.randn(512, 384, device=dev).bfloat16()
        y = ops.linear(x, w, b)
        y.backward(dy)
        assert w.grad is None  # nothing materialised in bf16
        ref_w += dy.float().t() @ x.detach().float()
        ref_b += dy.float().sum(0)
        ref_dx = dy.float() @ w.detach().float()
        assert torch.allclose(x.grad.float(), ref_dx, atol=0.15, rtol=2e-2)
        assert torch.allclose(y.float(), torch.nn.functional.linear(x.detach().float(), w.detach().float()), atol=0.1,
                              rtol=2e-2)
    assert torch.allclose(w.main_grad, ref_w, atol=2e-2, rtol=1e-3), (w.main_grad - ref_w).abs().max()
    assert torch.allclose(b.grad.float(), ref_b, atol=0.5, rtol=2e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("tile_m", [64, 128])
@pytest.mark.parametrize("K", [256, 192])
@pytest.mark.parametrize("N", [192, 256, 384, 512])
@pytest.mark.parametrize("counts", [[64, 0, 130, 1, 61], [5, 300, 7, 0, 0, 200, 33, 64], [129, 128, 127, 0, 513]])
def test_grouped_gemm_matches_reference_fwd_bwd(counts, N, K, tile_m, monkeypatch):
    """Every tile shape of the kernel (64 / 128 rows x 64 / 128 columns), experts with no rows, with fewer rows
    than a tile and with rows just past a tile boundary; forward, dx (the untransposed-weight kernel when K is a
    multiple of 128, else the forward kernel on a transposed copy) and dW (the grouped weight-gradient kernel when
    N and K are multiples of 128, library GEMMs per expert otherwise)."""
    from areal_vllm_amd import ops

    monkeypatch.setenv("AREAL_GROUPED_GEMM_TILE_M", str(tile_m))
    monkeypatch.setenv("AREAL_GROUPED_GEMM", "hip")
    torch.manual_seed(0)
    dev = torch.device("cuda")
    E = len(counts)
    T = sum(counts)
    x = (torch.randn(T, K, device=dev)).bfloat16().requires_grad_(True)
    w = (torch.randn(E, N, K, device=dev) * 0.05).bfloat16().requires_grad_(True)
    dy = torch.randn(T, N, device=dev).bfloat16()
    out = ops.grouped_gemm(x, w, counts)
    xf, wf = x.detach().float().requires_grad_(True), w.detach().float().requires_grad_(True)
    ref = ops.grouped_gemm_ref(xf, wf, counts)
    assert out.shape == ref.shape
    assert torch.allclose(out.float(), ref, atol=3e-2, rtol=2e-2), (out.float() - ref).abs().max()
    out.backward(dy)
    ref.backward(dy.float())
    assert torch.allclose(x.grad.float(), xf.grad, atol=3e-2, rtol=2e-2), (x.grad.float() - xf.grad).abs().max()
    assert torch.allclose(w.grad.float(), wf.grad, atol=0.25, rtol=3e-2), (w.grad.float() - wf.grad).abs().max()


# ------------------------------------------------------------------ decode-batch GEMM + slab row kernels
from areal_vllm_amd.ops import decode_gemm as DG  # noqa: E402

# (N, K) of the linear layers of Qwen2-1.5B and Qwen2.5-7B: qkv, o, gate|up, down
_DG_SHAPES = {"1.5b": [(2048, 1536), (1536, 1536), (17920, 1536), (1536, 8960)],
              "7b": [(4608, 3584), (3584, 3584), (37888, 3584), (3584, 18944)]}


@pytest.mark.parametrize("M", [1, 8, 64, 128, 256, 384, 512])
@pytest.mark.parametrize("model", ["1.5b", "7b"])
def test_decode_gemm_model_shapes(M, model):
    """Heuristic configuration of every layer shape: split-K slabs (qkv / o / down) and the SwiGLU epilogue
    (gate|up) against an fp32 reference of the same bf16 inputs."""
    torch.manual_seed(M)
    for li, (N, K) in enumerate(_DG_SHAPES[model]):
        x = torch.randn(M, K, device=DEV).bfloat16()
        w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).bfloat16()
        ref = x.float() @ w.float().t()
        if li == 2:
            out = DG.dgemm(x, w, epilogue=DG.EPI_SWIGLU)
            f = N // 2
            rb = ref.bfloat16().float()
            _close(out, torch.nn.functional.silu(rb[:, :f]) * rb[:, f:], 2e-2, 2e-2, f"swiglu N={N} K={K}")
        else:
            slab = DG.dgemm(x, w, epilogue=DG.EPI_SLAB)
            _close(slab.sum(0), ref, 2e-3, 2e-3, f"slab N={N} K={K} S={slab.shape[0]}")


@pytest.mark.parametrize("cfg", sorted(DG.DG_CONFIGS))
@pytest.mark.parametrize("lw,nt", [(0, False), (4, True)])
def test_decode_gemm_every_tile_config(cfg, lw, nt):
    """Every tile configuration x pipeline depth x epilogue, on ragged sizes (M and N not multiples of the
    tile), with asymmetric operands so that a transposed or permuted output cannot pass."""
    torch.manual_seed(cfg)
    bm, bn, _ = DG.DG_CONFIGS[cfg]
    if lw and ((bm + bn) // 8) % lw:
        pytest.skip("loader waves do not divide this tile's staging pieces")
    M, N, K = bm + 37, 3 * bn + 32, 704
    x = torch.randn(M, K, device=DEV).bfloat16()
    w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).bfloat16()
    w[5] *= 3.0
    bias = torch.randn(N, device=DEV).bfloat16()
    ref = x.float() @ w.float().t()
    for stages in range(2, DG.max_stages(cfg) + 1):
        out = DG.dgemm(x, w, bias, epilogue=DG.EPI_BF16, cfg=cfg, stages=stages, loader_waves=lw, nt=nt)
        _close(out, ref + bias.float(), 2e-2, 2e-2, f"bf16 cfg={cfg} stages={stages}")
        for ks in (1, 3, 11):
            slab = DG.dgemm(x, w, epilogue=DG.EPI_SLAB, cfg=cfg, stages=stages, ksplit=ks, loader_waves=lw, nt=nt)
            assert slab.shape[0] == DG.valid_ksplit(K, ks)
            _close(slab.sum(0), ref, 2e-3, 2e-3, f"slab cfg={cfg} stages={stages} ks={ks}")
    N2 = 2 * (2 * bn + 32)
    w2 = (torch.randn(N2, K, device=DEV) / math.sqrt(K)).bfloat16()
    rb = (x.float() @ w2.float().t()).bfloat16().float()
    out = DG.dgemm(x, w2, epilogue=DG.EPI_SWIGLU, cfg=cfg, stages=2, loader_waves=lw, nt=nt)
    _close(out, torch.nn.functional.silu(rb[:, :N2 // 2]) * rb[:, N2 // 2:], 2e-2, 2e-2, f"swiglu cfg={cfg}")


@pytest.mark.parametrize("M,N,S", [(1, 1536, 1), (100, 1536, 10), (256, 3584, 3), (384, 5120, 6)])
@pytest.mark.parametrize("with_res,with_w", [(True, True), (False, True), (True, False)])
def test_splitk_residual_rmsnorm(M, N, S, with_res, with_w):
    torch.manual_seed(1)
    slab = torch.randn(S, M, N, device=DEV)
    bias = torch.randn(N, device=DEV).bfloat16()
    res = torch.randn(M, N, device=DEV).bfloat16() if with_res else None
    w = (1 + 0.1 * torch.randn(N, device=DEV)).bfloat16() if with_w else None
    y, r = DG.splitk_residual_rmsnorm(slab, S, M, N, bias=bias, residual=res, weight=w, eps=1e-6)
    s = (slab.sum(0) + bias.float()).bfloat16()
    if with_res:
        s = (s.float() + res.float()).bfloat16()
    _close(r, s, 2e-2, 1e-2, "residual stream")
    if with_w:
        sf = r.float()
        _close(y, sf * torch.rsqrt(sf.pow(2).mean(-1, keepdim=True) + 1e-6) * w.float(), 2e-2, 2e-2, "normed")
    else:
        assert y.data_ptr() == r.data_ptr()


@pytest.mark.parametrize("Hq,Hkv,D", [(12, 2, 128), (28, 4, 128), (16, 2, 64)])
def test_splitk_rope_cache_write_matches_unfused(Hq, Hkv, D):
    torch.manual_seed(2)
    T, S, P = 75, 4, 40
    Nq = (Hq + 2 * Hkv) * D
    slab = torch.randn(S, T, Nq, device=DEV)
    bias = torch.randn(Nq, device=DEV).bfloat16()
    pos = torch.randint(0, 900, (T,), device=DEV, dtype=torch.int32)
    slots = torch.randperm(P * 32, device=DEV)[:T].long()
    slots[3] = -1
    table = NA.RotaryTable(D, 1024, 1e4, DEV)
    kc, vc = A.alloc_kv_cache(P, Hkv, D, torch.bfloat16, DEV)
    kc2, vc2 = A.alloc_kv_cache(P, Hkv, D, torch.bfloat16, DEV)
    q = DG.splitk_rope_cache_write(slab, S, bias, pos, slots, table, kc, vc, Hq, Hkv, D)
    qkv = (slab.sum(0) + bias.float()).bfloat16().view(T, Hq + 2 * Hkv, D).contiguous()
    A.rope_and_cache_write_(qkv, pos, slots, table, kc2, vc2, Hq, Hkv)
    _close(q, qkv[:, :Hq], 2e-2, 1e-2, "q")
    _close(kc, kc2, 2e-2, 1e-2, "k cache")
    _close(vc, vc2, 2e-2, 1e-2, "v cache")


@pytest.mark.parametrize("Hq,Hkv,D,K", [(12, 2, 128, 1536), (28, 4, 128, 3584), (16, 2, 64, 896)])
@pytest.mark.parametrize("cfg,lw", [(None, None), (9, 0), (9, 4), (8, 0), (8, 4), (3, 4), (4, 0)])
@pytest.mark.parametrize("T", [75, 256])
def test_dgemm_qkv_rope_matches_unfused(Hq, Hkv, D, K, cfg, lw, T):
    """The qkv projection with bias + RoPE + KV-cache write in its epilogue against an fp32 GEMM of the same
    bf16 inputs followed by the stand-alone RoPE / cache-write kernel."""
    if cfg == 4 and D < 128:
        pytest.skip("64-column half tiles need D >= 128")
    torch.manual_seed(3)
    P = 40
    Nq = (Hq + 2 * Hkv) * D
    x = torch.randn(T, K, device=DEV).bfloat16()
    w = (torch.randn(Nq, K, device=DEV) / math.sqrt(K)).bfloat16()
    bias = torch.randn(Nq, device=DEV).bfloat16()
    pos = torch.randint(0, 900, (T,), device=DEV, dtype=torch.int32)
    slots = torch.randperm(P * 32, device=DEV)[:T].long()
    slots[3] = -1
    table = NA.RotaryTable(D, 1024, 1e4, DEV)
    kc, vc = A.alloc_kv_cache(P, Hkv, D, torch.bfloat16, DEV)
    kc2, vc2 = A.alloc_kv_cache(P, Hkv, D, torch.bfloat16, DEV)
    for stages in ((None,) if cfg is None else (2, 4, 8)):
        kc.zero_(), vc.zero_()
        q = DG.dgemm_qkv_rope(x, w, bias, pos, slots, table, kc, vc, Hq, Hkv, D, cfg=cfg, stages=stages,
                              loader_waves=lw, nt=bool(lw))
        qkv = (x.float() @ w.float().t() + bias.float()).bfloat16().view(T, Hq + 2 * Hkv, D).contiguous()
        kc2.zero_(), vc2.zero_()
        A.rope_and_cache_write_(qkv, pos, slots, table, kc2, vc2, Hq, Hkv)
        _close(q, qkv[:, :Hq], 3e-2, 2e-2, f"q cfg={cfg} stages={stages}")
        _close(kc, kc2, 3e-2, 2e-2, f"k cache cfg={cfg} stages={stages}")
        _close(vc, vc2, 3e-2, 2e-2, f"v cache cfg={cfg} stages={stages}")


@pytest.mark.parametrize("prefix,chunk", [(0, 70), (31, 5), (32, 200), (4096, 130), (100, 1), (650, 300)])
@pytest.mark.parametrize("Hq,Hkv", [(12, 2), (4, 4)])
def test_paged_prefill_attention_matches_fp32(prefix, chunk, Hq, Hkv):
    """Prefill queries against the paged cache (flash_attn_paged.hip): several requests with different prefix /
    chunk lengths in one launch, pages scattered over the pool, stale non-finite values behind the end of every
    context, against fp32 attention over the gathered context."""
    torch.manual_seed(prefix + chunk)
    dev, D = torch.device("cuda"), 128
    reqs = [(prefix, chunk), (max(0, prefix // 2), chunk + 3), (prefix + 17, max(1, chunk // 2))]
    n_pages = sum((p + c + 31) // 32 for p, c in reqs) + 4
    kc, vc = A.alloc_kv_cache(n_pages, Hkv, D, torch.bfloat16, dev)
    kc.fill_(float("nan"))
    vc.fill_(float("nan"))
    perm = torch.randperm(n_pages)[: n_pages - 4].tolist()
    max_pages = max((p + c + 31) // 32 for p, c in reqs)
    bt = torch.zeros(len(reqs), max_pages, dtype=torch.int32)
    qs, ks, vs, cu_q, cu_k = [], [], [], [0], [0]
    for i, (p, c) in enumerate(reqs):
        n = p + c
        npg = (n + 31) // 32
        pages, perm = perm[:npg], perm[npg:]
        bt[i, :npg] = torch.tensor(pages, dtype=torch.int32)
        k = torch.randn(n, Hkv, D, device=dev).bfloat16()
        v = torch.randn(n, Hkv, D, device=dev).bfloat16()
        t = torch.arange(n)
        pg = torch.tensor(pages)[t // 32].to(dev)
        kc[pg, :, (t % 32).to(dev), :] = k
        vc[pg, :, :, (t % 32).to(dev)] = v
        qs.append(torch.randn(c, Hq, D, device=dev).bfloat16())
        ks.append(k)
        vs.append(v)
        cu_q.append(cu_q[-1] + c)
        cu_k.append(cu_k[-1] + n)
    q = torch.cat(qs)
    cq = torch.tensor(cu_q, dtype=torch.int32, device=dev)
    ck = torch.tensor(cu_k, dtype=torch.int32, device=dev)
    out = A.flash_attn_paged_prefill(q, kc, vc, bt.to(dev), cq, ck, max(c for _, c in reqs))
    want, _ = A.attention_varlen_ref(q.float(), torch.cat(ks).float(), torch.cat(vs).float(), cq, ck, 1.0 / math.sqrt(D))
    _close(out, want, 2e-2, 2e-2, f"prefix={prefix} chunk={chunk}")
    # the slot list the fallback path gathers with names the same tokens
    slots = A.context_slots(bt.to(dev), ck)
    kg, vg = A.cache_gather(kc, vc, slots)
    assert torch.equal(kg, torch.cat(ks)) and torch.equal(vg, torch.cat(vs))


@pytest.mark.gpu
@pytest.mark.parametrize("M,Fc,K,cols,mblocks", [
    (256, 8960, 1536, 36, 1),   # Qwen2-1.5B decode: 249 workgroups
    (384, 8960, 1536, 72, 2),   # two row blocks of 192 rows, ranges paired on one XCD
    (512, 8960, 1536, 72, 2),
    (100, 8960, 1536, 36, 1),   # ragged rows
    (130, 1000, 256, 44, 1),    # last range narrower than the others, few K steps
    (300, 1000, 256, 20, 2),    # more blocks than ranges need (rounded up to the XCD pairing)
    (256, 18944, 3584, 76, 1),  # Qwen2.5-7B shapes: five 16-column tiles per half
    (64, 128, 64, 4, 1),        # a single K step, the narrowest range
])
def test_balanced_swiglu_gemm_matches_fp32(M, Fc, K, cols, mblocks):
    """gate|up GEMM + SwiGLU with one column range of arbitrary width per workgroup (decode_swiglu.hip) against
    fp32, with the rounding points of the unfused bf16 path, for every instantiated row-tile / column-tile count."""
    from areal_vllm_amd.ops import decode_gemm as DG

    torch.manual_seed(M + Fc)
    dev = torch.device("cuda")
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(2 * Fc, K, device=dev) * (1.0 / K ** 0.5)).bfloat16()
    y = x.float() @ w.float().t()
    y = y.bfloat16().float()
    want = F.silu(y[:, :Fc]) * y[:, Fc:]
    for stages in (2, 3, 4):
        out = torch.full((M + 3, Fc), float("nan"), dtype=torch.bfloat16, device=dev)
        DG.dgemm_swiglu_balanced(x, w, out[:M], cols=cols, mblocks=mblocks, stages=stages)
        assert torch.isnan(out[M:].float()).all()  # rows past M untouched
        _close(out[:M], want, 3e-2, 2e-2, f"stages={stages}")
    plan = DG.swiglu_plan(M, Fc)
    if plan is not None:
        _close(DG.dgemm_swiglu_balanced(x, w), want, 3e-2, 2e-2, "default plan")


@pytest.mark.parametrize("M,Fc,K", [(8, 1024, 512), (256, 8960, 1536), (100, 2048, 256)])
def test_gated_mlp_epilogues_with_gelu_gate(M, Fc, K):
    """GeGLU (tanh-approximated GELU on the gate, Gemma) in the epilogue of the tiled and of the balanced decode
    GEMM against fp32 with the rounding points of the unfused path."""
    from areal_vllm_amd.ops import decode_gemm as DG

    torch.manual_seed(M + Fc + 1)
    dev = torch.device("cuda")
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(2 * Fc, K, device=dev) * (1.0 / K ** 0.5)).bfloat16()
    y = (x.float() @ w.float().t()).bfloat16().float()
    want = F.gelu(y[:, :Fc], approximate="tanh") * y[:, Fc:]
    _close(DG.dgemm(x, w, epilogue=DG.EPI_GEGLU), want, 3e-2, 2e-2, "tiled")
    if DG.swiglu_plan(M, Fc) is not None:
        _close(DG.dgemm_swiglu_balanced(x, w, act="gelu_tanh"), want, 3e-2, 2e-2, "balanced")
    # the SiLU gate is untouched by the flag
    _close(DG.dgemm(x, w, epilogue=DG.EPI_SWIGLU), F.silu(y[:, :Fc]) * y[:, Fc:], 3e-2, 2e-2, "silu")


def _ipc_allreduce_worker(rank, world, port, out_dir):
    import os
    import sys

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist

    from areal_vllm_amd.ops.ipc_allreduce import IpcAllReduce

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ar = IpcAllReduce(dist.group.WORLD, dev, 256 * 1536)
    assert ar.self_check()
    results = []
    sizes = [8, 1000, 1536 * 3 + 5, 64 * 1536, 256 * 1536]

    def inputs(n, call):
        g = torch.Generator().manual_seed(1000 * call + n)
        return [torch.randn(n, generator=g).bfloat16() for _ in range(world)]

    call = 0
    for rep in range(4):  # repeated use of both staging slots; the two kernels interleaved on one sequence counter
        for n in sizes:
            xs = inputs(n, call)
            x = xs[rank].to(dev)
            ar.all_reduce_(x, two_shot=(rep + call) % 2 == 1)
            want = torch.stack([t.float() for t in xs]).sum(0).bfloat16()
            got = x.cpu()
            assert torch.equal(got, want), (rank, n, (got.float() - want.float()).abs().max())
            results.append(got)
            call += 1
    # a captured graph replays the collective: the sequence number lives in device memory
    buf = torch.zeros(64 * 1536, dtype=torch.bfloat16, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ar.all_reduce_(buf)  # warm-up launch outside the capture (both ranks: keeps the call sequences aligned)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ar.all_reduce_(buf)
        ar.all_reduce_(buf, two_shot=True)
    for rep in range(3):
        buf.fill_(float(rank + 1 + rep))
        graph.replay()
        torch.cuda.synchronize()
        total = sum(r + 1 + rep for r in range(world))
        assert torch.equal(buf.cpu(), torch.full_like(buf.cpu(), float(total * world))), (rank, rep, buf[:4])
    assert not ar.error()
    torch.save(results, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    ar.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_peer_memory_all_reduce_between_processes(tmp_path, world):
    """Processes on this GPU map each other's staging regions and all-reduce bf16 messages of decode sizes with the
    one-shot and the two-shot kernel interleaved: exact fp32-ordered sums, identical bits on every rank, graph
    replay."""
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_ipc_allreduce_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(150)
    alive = [p for p in procs if p.is_alive()]
    for p in alive:
        p.kill()
    assert not alive and all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    first, *others = (torch.load(tmp_path / f"r{r}.pt", weights_only=True) for r in range(world))
    for other in others:
        assert all(torch.equal(x, y) for x, y in zip(first, other))


@pytest.mark.gpu
@pytest.mark.parametrize("T,k,E,H", [(37, 2, 4, 256), (300, 8, 16, 2048), (5, 1, 3, 64)])
def test_moe_permute_and_combine_match_scatter_formulation(T, k, E, H):
    """Rows into expert order and the weighted recombination as gathers (forward and every gradient) against the
    index / index_add formulation in fp32."""
    from areal_vllm_amd import ops

    torch.manual_seed(T)
    dev = torch.device("cuda")
    x = torch.randn(T, H, device=dev).bfloat16().requires_grad_(True)
    wts = torch.rand(T, k, device=dev).requires_grad_(True)
    experts = torch.stack([torch.randperm(E, device=dev)[:k] if k <= E else torch.arange(k, device=dev) % E
                           for _ in range(T)])
    order = torch.argsort(experts.reshape(-1), stable=True)
    inv = torch.empty_like(order)
    inv[order] = torch.arange(order.numel(), device=dev)
    xs = ops.moe_permute(x, order, inv, k)
    assert torch.equal(xs, x.detach()[order // k])
    scale = torch.randn(T * k, 1, device=dev).bfloat16()  # stand-in for the experts: row-wise scaling
    out = ops.moe_combine(xs * scale, wts, order, inv)
    g = torch.randn_like(out)
    out.backward(g)
    xf, wf = x.detach().float().requires_grad_(True), wts.detach().clone().requires_grad_(True)
    tok = order // k
    ys = xf[tok] * scale.float()
    ref = torch.zeros(T, H, device=dev).index_add_(0, tok, ys * wf.reshape(-1)[order].unsqueeze(-1))
    ref.backward(g.float())
    _close(out, ref, 3e-2, 3e-2, "combine out")
    _close(x.grad, xf.grad, 5e-2, 5e-2, "dx")
    _close(wts.grad, wf.grad, 0.02 * math.sqrt(H), 5e-2, "dw")


@pytest.mark.gpu
@pytest.mark.parametrize("N", [256, 384])
def test_grouped_gemm_weight_gradient_accumulates_into_main_grad(N, monkeypatch):
    """A weight that carries an fp32 `main_grad` view gets its gradient ADDED there by the kernel (two backward
    passes = twice the gradient, experts without rows untouched) and no bf16 gradient tensor."""
    from areal_vllm_amd import ops

    monkeypatch.setenv("AREAL_GROUPED_GEMM", "hip")
    torch.manual_seed(1)
    dev = torch.device("cuda")
    counts, K = [70, 0, 200, 3], 256
    E, T = len(counts), sum(counts)
    x = torch.randn(T, K, device=dev).bfloat16().requires_grad_(True)
    w = (torch.randn(E, N, K, device=dev) * 0.05).bfloat16().requires_grad_(True)
    w.main_grad = torch.full((E, N, K), 0.5, dtype=torch.float32, device=dev)
    folded = []
    w._after_fold = lambda p: folded.append(p)
    dy = torch.randn(T, N, device=dev).bfloat16()
    for _ in range(2):
        ops.grouped_gemm(x, w, counts).backward(dy)
    assert w.grad is None and len(folded) == 2
    xf, wf = x.detach().float(), w.detach().float().requires_grad_(True)
    ops.grouped_gemm_ref(xf, wf, counts).backward(dy.float())
    _close(w.main_grad, 0.5 + 2 * wf.grad, 0.1, 2e-2, "main_grad")
    assert torch.equal(w.main_grad[1], torch.full((N, K), 0.5, device=dev))


@pytest.mark.gpu
@pytest.mark.parametrize("with_main_grad", [False, True])
def test_grouped_gemm_library_loop_for_few_large_experts(with_main_grad, monkeypatch):
    """The dispatch for few, large experts: one library GEMM per expert in forward and backward, dW accumulated
    into the fp32 `main_grad` view when the weight has one."""
    from areal_vllm_amd import ops

    monkeypatch.setenv("AREAL_GROUPED_GEMM", "library")
    torch.manual_seed(2)
    dev = torch.device("cuda")
    counts, N, K = [300, 0, 129, 70], 256, 192
    E, T = len(counts), sum(counts)
    x = torch.randn(T, K, device=dev).bfloat16().requires_grad_(True)
    w = (torch.randn(E, N, K, device=dev) * 0.05).bfloat16().requires_grad_(True)
    if with_main_grad:
        w.main_grad = torch.zeros(E, N, K, dtype=torch.float32, device=dev)
    dy = torch.randn(T, N, device=dev).bfloat16()
    out = ops.grouped_gemm(x, w, counts)
    out.backward(dy)
    xf, wf = x.detach().float().requires_grad_(True), w.detach().float().requires_grad_(True)
    ref = ops.grouped_gemm_ref(xf, wf, counts)
    ref.backward(dy.float())
    _close(out, ref, 3e-2, 2e-2, "out")
    _close(x.grad, xf.grad, 3e-2, 2e-2, "dx")
    if with_main_grad:
        assert w.grad is None
        _close(w.main_grad, wf.grad, 0.1, 2e-2, "main_grad")
    else:
        _close(w.grad, wf.grad, 0.25, 3e-2, "dw")


# ------------------------------------------------------------------------------ dense training GEMMs
@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,bias", [(256, 256, 128, False), (1000, 1536, 1536, True), (4096 + 77, 2048, 1536, True),
                                        (3000, 1284, 256, False), (513, 8960, 1536, False), (2048, 1536, 8960, False)])
@pytest.mark.parametrize("max_blocks", [256, 0, 8])
def test_training_gemm_matches_fp32(M, N, K, bias, max_blocks):
    """8-phase LDS-DMA GEMM C = A . B^T (+ bias): row / column tails, one workgroup per tile, persistent workgroups
    walking several tiles each (8 workgroups: every block streams through many output tiles)."""
    from areal_vllm_amd.ops import train_gemm as tg

    torch.manual_seed(M + N + K)
    dev = torch.device("cuda")
    a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
    b = ((torch.rand(N, K, device=dev) * 2 - 1) / K ** 0.5).bfloat16()
    bv = torch.randn(N, device=dev).bfloat16() if bias else None
    assert tg.supported(a, b, bv)
    got = tg.tgemm_tn(a, b, bv, max_blocks=max_blocks)
    ref = a.float() @ b.float().t() + (bv.float() if bias else 0.0)
    torch.testing.assert_close(got.float(), ref, atol=0.03, rtol=0.02)
    lib = torch.nn.functional.linear(a, b, bv)
    assert float((got.float() - ref).abs().max()) <= 2.0 * float((lib.float() - ref).abs().max()) + 1e-3
    # a strided view of a wider buffer as A, a strided output
    wide = torch.zeros(M, K + 64, device=dev, dtype=torch.bfloat16)
    wide[:, :K] = a
    out = torch.full((M, N + 8), 7.0, device=dev, dtype=torch.bfloat16)
    tg.tgemm_tn(wide[:, :K], b, bv, out=out[:, :N], max_blocks=max_blocks)
    assert torch.equal(out[:, :N], got) and bool((out[:, N:] == 7.0).all())


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (1000, 512, 256), (4096 + 77, 2048, 1536), (8192, 1536, 1536),
                                   (2048, 1792, 1536), (1024, 1920, 512), (640, 128, 256)])
@pytest.mark.parametrize("splits", [0, 1, 3])
def test_weight_gradient_gemm_matches_fp64(M, N, K, splits):
    """acc += dY^T . X with both operands token-major (transposed LDS reads), plain and split over tokens (atomic
    adds), token counts that are not multiples of the 128-token pairs the kernel reduces."""
    from areal_vllm_amd.ops import train_gemm as tg

    torch.manual_seed(M + N)
    dev = torch.device("cuda")
    dy = ((torch.rand(M, N, device=dev) * 2 - 1) / M ** 0.5).bfloat16()
    x = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
    base = torch.randn(N, K, device=dev)
    assert tg.wgrad_supported(dy, x, base)
    got = tg.tgemm_wgrad_(base.clone(), dy, x, splits=splits)
    ref = base.double() + dy.double().t() @ x.double()
    assert float((got.double() - ref).abs().max()) < 2e-5
    # through the autograd function the trainer uses: gradients land in the fp32 main_grad view
    from areal_vllm_amd import ops

    w = torch.nn.Parameter(torch.randn(N, K, device=dev).bfloat16() * 0.02)
    w.main_grad = torch.zeros(N, K, device=dev)
    xin = x.clone().requires_grad_(True)
    ops.linear(xin, w).backward(dy)
    torch.testing.assert_close(w.main_grad.double(), dy.double().t() @ x.double(), atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(xin.grad.float(), dy.float() @ w.float(), atol=2e-2, rtol=2e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("M,Fo,K", [(256, 128, 128), (4096 + 77, 8960, 1536), (1000, 1284, 256), (2048, 4480, 1536)])
def test_fused_gate_up_swiglu_forward_matches_unfused(M, Fo, K):
    """gate|up projection + SwiGLU in one kernel (no-grad forwards) against projection, bf16 round, activation."""
    from areal_vllm_amd.ops import train_gemm as tg

    torch.manual_seed(M + Fo)
    dev = torch.device("cuda")
    x = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
    w = ((torch.rand(2 * Fo, K, device=dev) * 2 - 1) * (3.0 / K ** 0.5)).bfloat16()
    got = tg.tgemm_swiglu(x, w)
    gu = (x.float() @ w.float().t()).bfloat16().float()
    ref = torch.nn.functional.silu(gu[:, :Fo]) * gu[:, Fo:]
    torch.testing.assert_close(got.float(), ref, atol=0.03, rtol=0.03)
    if Fo % 8 == 0:  # the stand-alone activation kernel takes multiples of 8
        unfused = ops.swiglu(torch.nn.functional.linear(x, w), "silu")
        assert float((got.float() - ref).abs().max()) <= 2.0 * float((unfused.float() - ref).abs().max()) + 2e-2
    # the dispatcher: fused without gradients from GATED_MIN_ROWS rows on, otherwise left to the caller
    with torch.no_grad():
        r = tg.gate_up_act(x, w, "silu")
    assert (r is not None) == (M >= tg.GATED_MIN_ROWS) and (r is None or torch.equal(r, got))
    assert tg.gate_up_act(x.requires_grad_(True), w, "silu") is None and tg.gate_up_act(x.detach(), w, "gelu_tanh") is None
