"""Router options of the mixture-of-experts layer (z-loss, capacity factor with the two drop policies, Sinkhorn
routing, input jitter; reference ``realhf/impl/model/modules/moe/router.py:53-161`` and
``realhf/impl/model/utils/moe.py``) and dropout with masks keyed for tensor parallelism and activation recomputation
(role of ``realhf/impl/model/utils/random.py``). No fixture of the reference covers these: parity unpinned, every
check is against a direct PyTorch formulation."""
import dataclasses
import math

import pytest
import torch
import torch.nn.functional as F

from areal_vllm_amd.models.config import preset
from areal_vllm_amd.models.transformer import MoE, Transformer
from areal_vllm_amd.utils import rng


def _moe(**kw):
    cfg = dataclasses.replace(preset("tiny-moe"), **kw)
    torch.manual_seed(0)
    m = MoE(cfg, 1, torch.float32, "cpu")
    for p in m.parameters():
        p.data.normal_(0, 0.2)
    return cfg, m


def _dense_reference(cfg, m, x, topv, topi):
    """Every token through its chosen experts, weight `topv` (zero = dropped pair)."""
    out = torch.zeros_like(x)
    f = m.gate_up.shape[1] // 2
    for t in range(x.shape[0]):
        for j in range(topi.shape[1]):
            if topv[t, j] == 0:
                continue
            e = int(topi[t, j])
            gu = m.gate_up[e] @ x[t]
            out[t] += topv[t, j] * (m.down[e] @ (F.silu(gu[:f]) * gu[f:]))
    return out


@pytest.mark.parametrize("policy", ["probs", "position"])
def test_capacity_factor_drops_pairs_beyond_the_expert_capacity(policy):
    cfg, m = _moe(moe_capacity_factor=0.5, moe_token_drop_policy=policy, router_aux_loss_coef=0.0)
    m.train()
    T, E, k = 24, cfg.n_experts, cfg.n_experts_per_tok
    x = torch.randn(T, cfg.hidden_size)
    y = m(x, None)
    probs = torch.softmax(F.linear(x, m.router).float(), -1)
    topv, topi = torch.topk(probs, k, -1)
    if cfg.norm_topk_prob:
        topv = topv / topv.sum(-1, keepdim=True)
    cap = math.ceil(T * k / E * 0.5)
    keep = torch.zeros(T, k, dtype=torch.bool)
    for e in range(E):
        pairs = [(t, j) for t in range(T) for j in range(k) if int(topi[t, j]) == e]
        if policy == "probs":
            pairs.sort(key=lambda tj: -float(topv[tj].detach()))
        for t, j in pairs[:cap]:
            keep[t, j] = True
    assert (~keep).any() and keep.any()
    want = _dense_reference(cfg, m, x, topv * keep, topi)
    torch.testing.assert_close(y, want, atol=1e-4, rtol=1e-4)
    # dropped pairs carry no gradient into the experts' inputs
    y.sum().backward()
    assert m.gate_up.grad is not None and torch.isfinite(m.gate_up.grad).all()


def test_z_loss_and_load_balancing_loss_add_up():
    cfg, m = _moe(moe_z_loss_coef=0.01, router_aux_loss_coef=0.02)
    m.train()
    x = torch.randn(40, cfg.hidden_size)
    m(x, None)
    logits = F.linear(x, m.router).float()
    probs = torch.softmax(logits, -1)
    topi = torch.topk(probs, cfg.n_experts_per_tok, -1).indices
    frac = torch.bincount(topi.reshape(-1), minlength=cfg.n_experts).float() / topi.numel()
    want = 0.01 * torch.logsumexp(logits, -1).square().mean() + 0.02 * cfg.n_experts * (frac * probs.mean(0)).sum()
    torch.testing.assert_close(m.aux_loss, want, atol=1e-6, rtol=1e-5)
    m.eval()
    m(x, None)  # evaluation leaves the training loss alone and computes none
    cfg2, m2 = _moe(moe_routing_type="none", router_aux_loss_coef=0.02)
    m2.train()
    m2(x, None)
    assert m2.aux_loss is None


def test_sinkhorn_routing_balances_the_experts_while_training():
    cfg, m = _moe(moe_routing_type="sinkhorn", router_aux_loss_coef=0.0)
    m.router.data[0] += 3.0 * m.router.data.abs().mean()  # a router that favours expert 0
    x = torch.randn(256, cfg.hidden_size).abs()
    logits = F.linear(x, m.router).float()
    greedy = torch.bincount(torch.topk(logits, cfg.n_experts_per_tok, -1).indices.reshape(-1), minlength=cfg.n_experts)
    norm = MoE._sinkhorn(logits)
    balanced = torch.bincount(torch.topk(norm, cfg.n_experts_per_tok, -1).indices.reshape(-1), minlength=cfg.n_experts)
    assert balanced.max() < greedy.max() and balanced.float().std() < greedy.float().std()
    torch.testing.assert_close(norm.sum(0), torch.full((cfg.n_experts,), 1.0 / cfg.n_experts), atol=1e-3, rtol=1e-2)
    m.train()
    y = m(x, None)
    y.sum().backward()
    assert torch.isfinite(y).all() and m.router.grad.abs().sum() > 0  # scores keep their gradient
    m.eval()
    assert torch.isfinite(m(x, None)).all()


def test_input_jitter_touches_the_router_only_while_training():
    cfg, m = _moe(moe_input_jitter_eps=0.3, router_aux_loss_coef=0.0)
    x = torch.randn(32, cfg.hidden_size)
    m.eval()
    a, b = m(x, None), m(x, None)
    assert torch.equal(a, b)
    m.train()
    torch.manual_seed(1)
    c = m(x, None)
    torch.manual_seed(2)
    d = m(x, None)
    assert not torch.equal(c, d)


# ----------------------------------------------------------------------------------------------- dropout
def test_dropout_masks_are_keyed_for_tensor_parallel_groups():
    x = torch.ones(64, 32)
    a, b = rng.DropoutKeys(), rng.DropoutKeys()
    a.configure(seed=5, replica=0, tp_rank=0)
    b.configure(seed=5, replica=0, tp_rank=1)
    for s in (a, b):
        s.advance()
    same = rng.dropout(x, 0.5, 3, 1, sharded=False, state=a), rng.dropout(x, 0.5, 3, 1, sharded=False, state=b)
    assert torch.equal(*same)  # replicated activation: the tensor-parallel ranks stay replicas
    apart = rng.dropout(x, 0.5, 3, 1, sharded=True, state=a), rng.dropout(x, 0.5, 3, 1, sharded=True, state=b)
    assert not torch.equal(*apart)  # sharded activation: every rank its own mask
    again = rng.dropout(x, 0.5, 3, 1, sharded=False, state=a)
    assert torch.equal(again, same[0])  # a recomputation of the same micro-batch sees the same mask
    a.advance()
    assert not torch.equal(rng.dropout(x, 0.5, 3, 1, state=a), same[0])  # the next micro-batch does not
    other_site = rng.dropout(x, 0.5, 3, 2, state=a), rng.dropout(x, 0.5, 4, 1, state=a)
    assert not torch.equal(*other_site)
    c = rng.DropoutKeys()
    c.configure(seed=5, replica=1, tp_rank=0)
    c.advance()
    c.advance()
    assert not torch.equal(rng.dropout(x, 0.5, 3, 1, state=c), rng.dropout(x, 0.5, 3, 1, state=a))  # replicas differ
    kept = rng.dropout(x, 0.25, 0, 0, state=a)
    assert kept.unique().tolist() == pytest.approx([0.0, 1.0 / 0.75]) and abs(float((kept > 0).float().mean()) - 0.75) < 0.05
    assert rng.dropout(x, 0.5, 0, 0, training=False) is x and rng.dropout(x, 0.0, 0, 0) is x


def test_recomputed_blocks_see_the_masks_of_the_first_forward():
    """Gradient checkpointing with dropout on: the recomputed forward must reproduce the masks, or the gradients
    belong to another network."""
    cfg = dataclasses.replace(preset("tiny-gpt2"), resid_pdrop=0.2, embd_pdrop=0.1)
    ids = torch.randint(2, cfg.vocab_size, (48,))
    pos = torch.arange(48, dtype=torch.int32)
    cu = torch.tensor([0, 20, 48], dtype=torch.int32)
    pos[20:] -= 20
    grads = []
    for ckpt in (False, True):
        m = Transformer(cfg, dtype=torch.float32, device="cpu", gradient_checkpointing=ckpt)
        m.init_weights(seed=4)
        m.train()
        rng.keys.configure(seed=11)
        rng.keys.counter = 0
        rng.keys.advance()
        h = m(ids, pos, cu, 28)
        h.square().mean().backward()
        grads.append({n: p.grad.clone() for n, p in m.named_parameters()})
    for n in grads[0]:
        torch.testing.assert_close(grads[0][n], grads[1][n], atol=1e-6, rtol=1e-5, msg=n)
    # and dropout is really on: evaluation mode gives another output than training mode
    m.eval()
    with torch.no_grad():
        e = m(ids, pos, cu, 28)
    m.train()
    with torch.no_grad():
        t = m(ids, pos, cu, 28)
    assert not torch.allclose(e, t)


def test_disable_dropout_zeroes_the_config(monkeypatch):
    from areal_vllm_amd.api.cli_args import TrainEngineConfig
    from areal_vllm_amd.api.io_struct import FinetuneSpec
    from areal_vllm_amd.engine import train_engine as te

    monkeypatch.setenv("AREAL_PLATFORM", "cpu")
    monkeypatch.setitem(te.PRESETS, "tiny-gpt2-drop", dict(te.PRESETS["tiny-gpt2"], resid_pdrop=0.1, attn_pdrop=0.1))
    for off in (True, False):
        cfg = TrainEngineConfig(experiment_name="t", trial_name="d", path="tiny-gpt2-drop", init_from_scratch=True,
                                dtype="float32", disable_dropout=off, optimizer=None)
        eng = te.NativeTrainEngine(cfg)
        eng.create_process_group()
        eng.initialize(None, FinetuneSpec(1, 8, 2))
        assert (eng.model_config.resid_pdrop == 0.0) == off and (eng.model_config.attn_pdrop == 0.0) == off
        eng.destroy()
