"""Native decoder-only transformer (Qwen2/2.5/3, Llama, Mistral, Gemma, Mixtral/Qwen-MoE, GPT-2) built on
the HIP ops of this package.

Design points (MI355X-first rather than a port of ``realhf/impl/model/nn/real_llm_api.py:100-834``):

* All parameters are views into ONE flat bf16 buffer (``FlatParams``) with a matching flat fp32
  gradient-accumulation buffer. The ZeRO sharder reduce-scatters / all-gathers those buffers directly,
  weight shipping to rollout engines is a handful of large peer copies, and the interval ops address
  parameters by flat offsets.
* q/k/v and gate/up projections are fused GEMMs; RoPE runs in place on the fused projection and the
  attention kernel reads q, k, v out of it with strides, so nothing is copied between them.
* The residual stream is carried through the fused residual-add RMSNorm kernel.
* Tensor parallelism: column-parallel qkv / gate_up, row-parallel o / down with an all-reduce over the
  TP group (xGMI); sequence-parallel variants live in ``parallel/tensor_parallel.py``.
"""
import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..parallel import comm
from ..utils import rng
from .config import ModelConfig

ALIGN = 128  # elements; keeps every parameter 256-byte aligned inside the flat buffer


def _pad(n: int, a: int = ALIGN) -> int:
    return (n + a - 1) // a * a


class FlatParams:
    """Flat storage for a module's parameters (bf16 or fp32) plus an fp32 main-grad buffer."""

    def __init__(self, module: nn.Module, dtype: torch.dtype, device, pad_to: int = 1, params=None):
        params = [(n, p) for n, p in module.named_parameters()] if params is None else list(params)
        seen = set()
        self.specs: List[Tuple[str, int, torch.Size]] = []
        off = 0
        for n, p in params:
            if id(p) in seen:
                continue
            seen.add(id(p))
            self.specs.append((n, off, p.shape))
            off += _pad(p.numel())
        self.numel = _pad(off, ALIGN * max(1, pad_to))
        self.flat = torch.zeros(self.numel, dtype=dtype, device=device)
        self.main_grad: Optional[torch.Tensor] = None
        named = dict(params)
        for n, o, shape in self.specs:
            p = named[n]
            view = self.flat[o:o + p.numel()].view(shape)
            if p.device.type != "meta":
                view.copy_(p.data)
            p.data = view
        self.offsets: Dict[str, Tuple[int, int]] = {n: (o, int(math.prod(s))) for n, o, s in self.specs}
        self._named = {n: named[n] for n, _, _ in self.specs}

    def alloc_main_grad(self):
        """fp32 gradient accumulation buffer; every parameter gets a `.main_grad` view and a hook that
        folds the bf16 autograd gradient into it (the role of the reference's wgrad_gemm_accum_fp32,
        ``realhf/impl/model/parallelism/tensor_parallel/modules.py:242-402``)."""
        self.main_grad = torch.zeros(self.numel, dtype=torch.float32, device=self.flat.device)
        for n, o, shape in self.specs:
            p = self._named[n]
            p.main_grad = self.main_grad[o:o + p.numel()].view(shape)
            if not getattr(p, "_main_grad_hooked", False):
                p.register_post_accumulate_grad_hook(_fold_grad)
                p._main_grad_hooked = True

    def zero_grad(self):
        if self.main_grad is not None:
            self.main_grad.zero_()
        for p in self._named.values():
            p.grad = None


def _fold_grad(p: torch.Tensor):
    if p.grad is not None:
        p.main_grad.add_(p.grad)
        p.grad = None
        done = getattr(p, "_after_fold", None)
        if done is not None:  # overlapped gradient reduction (parallel/zero.py) counts finished parameters
            done(p)


# ----------------------------------------------------------------------------- TP collectives
class _AllReduceFwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, group):
        if group is not None and dist.get_world_size(group) > 1:
            x = x.contiguous()
            dist.all_reduce(x, group=group)
        return x

    @staticmethod
    def backward(ctx, g):
        return g, None


class _AllReduceBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return x

    @staticmethod
    def backward(ctx, g):
        if ctx.group is not None and dist.get_world_size(ctx.group) > 1:
            g = g.contiguous()
            dist.all_reduce(g, group=ctx.group)
        return g, None


# ------------------------------------------------------------------------------------ layers
class Attention(nn.Module):
    def __init__(self, cfg: ModelConfig, tp_size: int, dtype, device):
        super().__init__()
        assert cfg.n_q_heads % tp_size == 0
        self.cfg = cfg
        self.hq = cfg.n_q_heads // tp_size
        # kv heads are replicated when tp_size exceeds their count
        self.hkv = max(1, cfg.n_kv_heads // tp_size)
        D = cfg.head_dim
        self.qkv = nn.Parameter(torch.empty((self.hq + 2 * self.hkv) * D, cfg.hidden_size, dtype=dtype, device=device))
        self.qkv_bias = (
            nn.Parameter(torch.zeros((self.hq + 2 * self.hkv) * D, dtype=dtype, device=device)) if cfg.qkv_bias else None
        )
        self.o = nn.Parameter(torch.empty(cfg.hidden_size, self.hq * D, dtype=dtype, device=device))
        if cfg.linear_bias:  # added once, after the row-parallel reduction
            self.o_bias = nn.Parameter(torch.zeros(cfg.hidden_size, dtype=dtype, device=device))
        if cfg.qk_norm:
            self.q_norm = nn.Parameter(torch.ones(D, dtype=dtype, device=device))
            self.k_norm = nn.Parameter(torch.ones(D, dtype=dtype, device=device))

    def project_qkv(self, x, reduce_dx_group=None):
        """Fused q|k|v projection, returned 2-D [T, (hq + 2 hkv) * D] (not a view, RoPE runs in place)."""
        T = x.shape[0]
        D = self.cfg.head_dim
        qkv = ops.linear(x, self.qkv, self.qkv_bias, reduce_dx_group=reduce_dx_group)
        if self.cfg.qk_norm:  # q and k heads normalised inside the packed tensor, v heads copied: one pass
            qkv = ops.qk_norm(qkv, self.q_norm, self.k_norm, self.cfg.rms_norm_eps, self.hq, self.hkv, D)
        return qkv

    def forward(self, x, positions, rope, cu_seqlens, max_seqlen, tp_group, sp=None, seq_par=False):
        D = self.cfg.head_dim
        ov = comm.overlap(tp_group) if seq_par else None
        if ov is not None and not (ov.usable(x) and not self.cfg.qk_norm):
            ov = None
        if ov is not None:  # the all-gather travels chunk by chunk under the projection's GEMMs
            from ..parallel import sp_overlap

            qkv = sp_overlap.gather_linear(x, self.qkv, ov, self.qkv_bias)
        elif seq_par:  # token-sharded input: all-gather (backward: reduce-scatter)
            from ..parallel.tensor_parallel import GatherFromSP, ReduceScatterToSP

            qkv = self.project_qkv(GatherFromSP.apply(x, tp_group))
        else:  # replicated input: the input-gradient all-reduce overlaps the weight-gradient GEMM
            qkv = self.project_qkv(x, reduce_dx_group=tp_group)
        T = qkv.shape[0]
        if rope is not None:
            qkv = ops.apply_rope(qkv, positions, rope, self.hq + self.hkv, D)
        qkv = qkv.view(T, self.hq + 2 * self.hkv, D)
        if sp is not None:
            attn = sp.attention(qkv, self.hq, self.hkv, cu_seqlens, max_seqlen)
        elif self.cfg.attn_pdrop > 0 and self.training:
            # dropout on the attention probabilities: the fused kernels have none, the per-sequence library path does
            # (its masks come from the framework's generator, which activation recomputation saves and restores; the
            # heads are sharded over the tensor-parallel group, whose ranks are seeded apart)
            attn = ops.attention_varlen_sdpa(qkv[:, :self.hq], qkv[:, self.hq:self.hq + self.hkv],
                                             qkv[:, self.hq + self.hkv:], cu_seqlens, cu_seqlens, 1.0 / math.sqrt(D),
                                             dropout_p=self.cfg.attn_pdrop)
        else:
            attn = ops.flash_attn_varlen_qkv(qkv, cu_seqlens, max_seqlen, self.hq, self.hkv)
        if ov is not None:
            return sp_overlap.linear_scatter(attn.reshape(T, -1), self.o, ov)
        out = ops.linear(attn.reshape(T, -1), self.o)
        if seq_par:
            return ReduceScatterToSP.apply(out, tp_group)
        out = _AllReduceFwd.apply(out, tp_group)
        return out + self.o_bias if self.cfg.linear_bias else out


class MLP(nn.Module):
    def __init__(self, hidden, inter, tp_size, dtype, device, act: str = "silu"):
        super().__init__()
        assert inter % tp_size == 0
        self.act = act
        f = inter // tp_size
        self.gate_up = nn.Parameter(torch.empty(2 * f, hidden, dtype=dtype, device=device))
        self.down = nn.Parameter(torch.empty(hidden, f, dtype=dtype, device=device))

    def forward(self, x, tp_group, seq_par=False):
        ov = comm.overlap(tp_group) if seq_par else None
        if ov is not None and ov.usable(x):
            from ..parallel import sp_overlap

            h = ops.swiglu(sp_overlap.gather_linear(x, self.gate_up, ov), self.act)
            return sp_overlap.linear_scatter(h, self.down, ov)
        if seq_par:
            from ..parallel.tensor_parallel import GatherFromSP, ReduceScatterToSP

            h = ops.swiglu(ops.linear(GatherFromSP.apply(x, tp_group), self.gate_up), self.act)
            return ReduceScatterToSP.apply(ops.linear(h, self.down), tp_group)
        h = ops.train_gemm.gate_up_act(x, self.gate_up, self.act)  # one kernel in no-grad forwards of many rows
        if h is None:
            h = ops.swiglu(ops.linear(x, self.gate_up, reduce_dx_group=tp_group), self.act)
        return _AllReduceFwd.apply(ops.linear(h, self.down), tp_group)


class DenseMLP(nn.Module):
    """fc -> activation -> proj without a gate, biases on both (GPT-2): column-parallel fc, row-parallel
    proj whose bias is added once after the reduction."""

    def __init__(self, hidden, inter, tp_size, dtype, device, act: str = "gelu_tanh"):
        super().__init__()
        assert inter % tp_size == 0 and act in ("gelu_tanh", "silu")
        self.act = act
        f = inter // tp_size
        self.fc = nn.Parameter(torch.empty(f, hidden, dtype=dtype, device=device))
        self.fc_bias = nn.Parameter(torch.zeros(f, dtype=dtype, device=device))
        self.proj = nn.Parameter(torch.empty(hidden, f, dtype=dtype, device=device))
        self.proj_bias = nn.Parameter(torch.zeros(hidden, dtype=dtype, device=device))

    def forward(self, x, tp_group, seq_par=False):
        assert not seq_par, "sequence parallelism is implemented for the gated MLP"
        h = ops.linear(x, self.fc, self.fc_bias, reduce_dx_group=tp_group)
        h = F.gelu(h, approximate="tanh") if self.act == "gelu_tanh" else F.silu(h)
        return _AllReduceFwd.apply(ops.linear(h, self.proj), tp_group) + self.proj_bias


class MoE(nn.Module):
    """Top-k routed experts (Mixtral / Qwen-MoE). Tokens are sorted by expert and every expert runs its
    fused gate|up and down GEMMs on a contiguous slab (grouped GEMM by expert-sorted dispatch; the
    reference uses grouped_gemm at ``realhf/impl/model/modules/moe/experts.py:195-210``). With expert
    parallelism (``ep_group``) each rank owns n_experts/ep experts and tokens travel by all-to-all.

    `tp_size` is the EXPERT tensor-parallel size: how many ranks split every expert's intermediate dimension.
    `token_shard` (expert tensor parallel size 1 inside a larger attention tp group, Megatron's
    ``expert_tensor_parallel_size``, ``areal/api/alloc_mode.py:35-214``): experts are whole, and since the layer's
    input is replicated over the tp group each tp rank routes only its slice of the tokens - the tp ranks act as
    additional expert-parallel / expert-data-parallel ranks, and the result is all-gathered."""

    def __init__(self, cfg: ModelConfig, tp_size, dtype, device, ep_size: int = 1, ep_rank: int = 0,
                 token_shard: bool = False):
        super().__init__()
        self.cfg = cfg
        self.ep_size, self.ep_rank = ep_size, ep_rank
        self.token_shard = token_shard
        assert cfg.n_experts % ep_size == 0
        self.n_local = cfg.n_experts // ep_size
        f = cfg.moe_intermediate_size // tp_size
        self.router = nn.Parameter(torch.empty(cfg.n_experts, cfg.hidden_size, dtype=dtype, device=device))
        self.gate_up = nn.Parameter(torch.empty(self.n_local, 2 * f, cfg.hidden_size, dtype=dtype, device=device))
        self.down = nn.Parameter(torch.empty(self.n_local, cfg.hidden_size, f, dtype=dtype, device=device))
        self.aux_loss = None

    def _experts(self, xs, counts):
        """All local experts on their expert-sorted token slabs: two grouped GEMMs around SwiGLU."""
        if xs.shape[0] == 0:
            return xs.new_zeros(0, self.cfg.hidden_size)
        h = ops.swiglu(ops.grouped_gemm(xs, self.gate_up, counts))
        return ops.grouped_gemm(h, self.down, counts)

    def forward(self, x, tp_group, ep_group=None):
        if not self.token_shard:
            x = _AllReduceBwd.apply(x, tp_group)
            return _AllReduceFwd.apply(self._route(x, self.router, ep_group, x.shape[0], 1.0), tp_group)
        from ..parallel import tensor_parallel as tpar

        # tokens split over the tp group (zero rows pad the last slice), whole experts, result re-assembled; the
        # router sees only this rank's slice, so its gradient is summed over the group
        T, n = x.shape[0], tpar._size(tp_group)
        pad = (-T) % n
        xs = tpar.ScatterToSP.apply(F.pad(x, (0, 0, 0, pad)) if pad else x, tp_group)
        c = xs.shape[0]
        r = torch.distributed.get_rank(tp_group) if n > 1 else 0
        valid = max(0, min(c, T - r * c))
        out = self._route(xs, _AllReduceBwd.apply(self.router, tp_group), ep_group, valid, 1.0 / n)
        return tpar.GatherForReplicatedUse.apply(out, tp_group)[:T]

    @staticmethod
    def _sinkhorn(logits: torch.Tensor, iters: int = 64, tol: float = 1e-4) -> torch.Tensor:
        """Sinkhorn-Knopp normalisation of exp(logits) [T, E]: rows and columns are rescaled alternately until every
        token carries weight 1/T and every expert receives 1/E. Carried out on log-potentials (a router that
        strongly favours one expert does not underflow the others); the convergence test runs every eight sweeps
        (one host synchronisation each) instead of after every sweep."""
        lk = logits.detach().float()
        T, E = lk.shape
        f = torch.zeros(T, device=lk.device)
        g = torch.zeros(E, device=lk.device)
        for it in range(iters):
            prev = g
            f = -torch.logsumexp(lk + g.unsqueeze(0), 1) - math.log(T)
            g = -torch.logsumexp(lk + f.unsqueeze(1), 0) - math.log(E)
            if it % 8 == 7 and float((prev - g).abs().max()) <= tol:
                break
        return torch.exp(lk + f.unsqueeze(1) + g.unsqueeze(0))

    def _route(self, x, router, ep_group, n_valid: int, aux_scale: float):
        """Route the rows of `x` (the first `n_valid` are real tokens) to their experts and combine the outputs.
        Router options: see `ModelConfig.moe_*` (reference ``realhf/impl/model/modules/moe/router.py:53-161``,
        ``realhf/impl/model/utils/moe.py:54-100,310-390``)."""
        cfg = self.cfg
        T, H = x.shape
        E, k = cfg.n_experts, cfg.n_experts_per_tok
        xr = x
        if cfg.moe_input_jitter_eps and self.training:  # the router alone sees the jittered input
            eps = float(cfg.moe_input_jitter_eps)
            xr = x * torch.empty_like(x).uniform_(1.0 - eps, 1.0 + eps)
        raw = F.linear(xr, router)
        plain = cfg.moe_routing_type in ("aux_loss", "none")
        fused_route = plain and not torch.is_grad_enabled() and not self.training
        logits = None if fused_route else raw.float()
        aux = None
        if cfg.moe_z_loss_coef > 0 and self.training and n_valid > 0:
            aux = aux_scale * cfg.moe_z_loss_coef * torch.logsumexp(logits[:n_valid], -1).square().mean()
        probs = None if fused_route else torch.softmax(logits, -1)
        if fused_route:
            # scoring / serving pass: softmax (fp32 arithmetic on the projection's output), top-k and
            # re-normalisation in one launch; weights in the dtype the combine multiplies in
            topv, topi = ops.moe_topk(raw, k, cfg.norm_topk_prob, out_dtype=x.dtype)
        elif cfg.moe_routing_type == "sinkhorn":
            act = torch.sigmoid(logits) if k == 1 else probs
            if self.training:
                with torch.no_grad():
                    topi = torch.topk(self._sinkhorn(logits.detach()), k, -1).indices
                topv = torch.gather(act, 1, topi)
            else:
                topv, topi = torch.topk(act, k, -1)
        else:
            assert cfg.moe_routing_type in ("aux_loss", "none"), f"unknown MoE routing type {cfg.moe_routing_type}"
            topv, topi = torch.topk(probs, k, -1)
            if cfg.norm_topk_prob:
                topv = topv / topv.sum(-1, keepdim=True)
        if cfg.moe_routing_type == "aux_loss" and cfg.router_aux_loss_coef > 0 and self.training:
            ti, pr = topi[:n_valid], probs[:n_valid]
            frac = torch.zeros(E, device=x.device).index_add_(
                0, ti.reshape(-1), torch.ones(ti.numel(), device=x.device)) / max(1, ti.numel())
            mean_p = pr.mean(0) if n_valid > 0 else probs.sum(0) * 0.0
            lb = aux_scale * cfg.router_aux_loss_coef * E * (frac * mean_p).sum()
            aux = lb if aux is None else aux + lb
        if self.training:
            self.aux_loss = aux
        flat_e = topi.reshape(-1)
        n_drop = 0
        if cfg.moe_capacity_factor is not None:
            # at most `cap` (token, choice) pairs per expert; the rest are dropped: they get weight zero and the
            # sentinel expert E, which sorts them behind every real expert's slab, where no expert touches them
            cap = max(1, math.ceil(max(1, n_valid) * k / E * float(cfg.moe_capacity_factor)))
            if cfg.moe_token_drop_policy == "probs":
                prio = torch.argsort(topv.detach().reshape(-1), descending=True, stable=True)
            else:
                assert cfg.moe_token_drop_policy == "position", cfg.moe_token_drop_policy
                prio = torch.arange(T * k, device=x.device)
            by_e = prio[torch.argsort(flat_e[prio], stable=True)]  # pairs grouped by expert, best first inside
            e_sorted = flat_e[by_e]
            start = torch.searchsorted(e_sorted, torch.arange(E, device=x.device))
            rank_in_e = torch.arange(T * k, device=x.device) - start[e_sorted]
            dropped = torch.zeros(T * k, dtype=torch.bool, device=x.device)
            dropped[by_e] = rank_in_e >= cap
            if n_valid < T:  # padding rows never occupy capacity
                dropped[n_valid * k:] = True
            topv = topv.masked_fill(dropped.view(T, k), 0.0)
            flat_e = flat_e.masked_fill(dropped, E)
        # expert-sorted position -> (token, choice) pair t*k + j (stable), its inverse, the source row of every
        # position and the experts' row ranges: one counting sort (``ops/moe.py::moe_sort``)
        order, inv, src_row, offsets = ops.moe_sort(flat_e, E, k)
        xs = ops.moe_permute(x, order, inv, k, src_row)
        if (cfg.moe_capacity_factor is None and not torch.is_grad_enabled() and not (ep_group is not None and self.ep_size > 1)
                and ops.grouped_gemm_offsets_ok(xs, self.gate_up) and ops.grouped_gemm_offsets_ok(xs, self.down)):
            # inference: the experts' row ranges stay on the device (the grouped kernels read them there), so the
            # layer has no host synchronisation and can sit inside a captured decode graph
            offsets = offsets[:E + 1]
            h = ops.swiglu(ops.grouped_gemm_offsets(xs, self.gate_up, offsets))
            return ops.moe_combine(ops.grouped_gemm_offsets(h, self.down, offsets), topv, order, inv)
        counts = (offsets[1:] - offsets[:-1]).long()  # E experts + the bin of dropped pairs
        if cfg.moe_capacity_factor is not None:
            n_drop = int(counts[E])
        counts = counts[:E]
        n_keep = T * k - n_drop
        if ep_group is not None and self.ep_size > 1:
            from ..parallel.expert_parallel import ep_dispatch_combine

            ys = ep_dispatch_combine(self, xs[:n_keep], counts, ep_group)
        else:
            ys = self._experts(xs[:n_keep], counts.tolist())
        if n_drop:
            ys = torch.cat([ys, ys.new_zeros(n_drop, ys.shape[1])])
        # every token gathers its k expert outputs (no scatter-add: each output row is written once)
        return ops.moe_combine(ys, topv, order, inv)


def add_norm(cfg: ModelConfig, h, weight, bias, residual):
    """(norm(h + residual), h + residual): the fused residual-add RMSNorm kernel, or LayerNorm with bias
    (fp32 statistics) for the families that use it."""
    if cfg.norm_type == "layer":
        residual = h if residual is None else h + residual
        x = F.layer_norm(residual.float(), (residual.shape[-1],), weight.float(), bias.float(), cfg.rms_norm_eps)
        return x.to(h.dtype), residual
    return ops.rmsnorm(h, weight, cfg.rms_norm_eps, residual, cfg.gemma_norm)


class Block(nn.Module):
    def __init__(self, cfg: ModelConfig, tp_size, dtype, device, ep_size=1, ep_rank=0, etp_size=None):
        super().__init__()
        self.cfg = cfg
        self.layer_id = 0  # global layer index (keys the dropout masks); set by the Transformer
        etp_size = tp_size if etp_size is None else etp_size
        self.ln1 = nn.Parameter(torch.ones(cfg.hidden_size, dtype=dtype, device=device))
        self.attn = Attention(cfg, tp_size, dtype, device)
        self.ln2 = nn.Parameter(torch.ones(cfg.hidden_size, dtype=dtype, device=device))
        if cfg.norm_type == "layer":
            self.ln1_bias = nn.Parameter(torch.zeros(cfg.hidden_size, dtype=dtype, device=device))
            self.ln2_bias = nn.Parameter(torch.zeros(cfg.hidden_size, dtype=dtype, device=device))
        if cfg.n_experts > 0:
            self.mlp = MoE(cfg, etp_size, dtype, device, ep_size, ep_rank, token_shard=etp_size != tp_size)
        elif cfg.mlp_gated:
            self.mlp = MLP(cfg.hidden_size, cfg.intermediate_size, tp_size, dtype, device, cfg.hidden_act)
        else:
            self.mlp = DenseMLP(cfg.hidden_size, cfg.intermediate_size, tp_size, dtype, device, cfg.hidden_act)

    def forward(self, h, residual, positions, rope, cu_seqlens, max_seqlen, tp_group, sp=None, ep_group=None,
                seq_par=False):
        cfg = self.cfg
        x, residual = add_norm(cfg, h, self.ln1, getattr(self, "ln1_bias", None), residual)
        h = self.attn(x, positions, rope, cu_seqlens, max_seqlen, tp_group, sp, seq_par)
        if cfg.resid_pdrop > 0:  # the residual stream is token-sharded under sequence parallelism, replicated otherwise
            h = rng.dropout(h, cfg.resid_pdrop, self.layer_id, 1, sharded=seq_par, training=self.training)
        x, residual = add_norm(cfg, h, self.ln2, getattr(self, "ln2_bias", None), residual)
        if cfg.n_experts > 0:
            h = self.mlp(x, tp_group, ep_group)
        else:
            h = self.mlp(x, tp_group, seq_par)
        if cfg.resid_pdrop > 0:
            h = rng.dropout(h, cfg.resid_pdrop, self.layer_id, 2, sharded=seq_par, training=self.training)
        return h, residual


class Transformer(nn.Module):
    """Packed-sequence decoder. `layer_range` selects a pipeline stage's slice of the layers; `layer_chunks`
    (interleaved pipeline schedules) several slices, one per virtual stage this rank hosts: `forward(chunk=c)`
    then runs only slice c. `layer_ids[i]` is the global index of local layer i."""

    def __init__(self, cfg: ModelConfig, dtype=torch.bfloat16, device="cpu", tp_size: int = 1, tp_rank: int = 0,
                 tp_group=None, layer_range: Optional[Tuple[int, int]] = None, has_embed: bool = True,
                 has_head: bool = True, ep_size: int = 1, ep_rank: int = 0, ep_group=None,
                 gradient_checkpointing: bool = False, sequence_parallel: bool = False,
                 vocab_parallel: bool = False, layer_chunks: Optional[List[Tuple[int, int]]] = None,
                 etp_size: Optional[int] = None):
        super().__init__()
        # expert tensor parallelism: the routed experts are split over `etp_size` ranks - the whole tp group
        # (default) or 1 (whole experts, tokens split over the tp group instead, see MoE)
        etp_size = tp_size if etp_size is None or cfg.n_experts == 0 else int(etp_size)
        assert etp_size in (1, tp_size), "expert tensor parallel size is the tensor parallel size or 1"
        self.etp_size = etp_size
        self.etp_rank = tp_rank if etp_size == tp_size else 0
        self.etp_group = tp_group if etp_size == tp_size else None
        # vocab-parallel: the token embedding and the LM head hold only this tp rank's vocabulary rows
        # (`vocab_range`); lookups are summed over the group and the loss merges per-shard softmax statistics
        self.vocab_parallel = bool(vocab_parallel) and tp_size > 1
        self.vocab_range = ops.vocab_shard_range(cfg.vocab_size, tp_size if self.vocab_parallel else 1, tp_rank)
        v_rows = self.vocab_range[1] - self.vocab_range[0]
        # Megatron-style sequence parallelism: between the tensor-parallel regions the residual stream,
        # norms and their activations hold only T/tp tokens per rank (all-gather before a column-parallel
        # GEMM, reduce-scatter after a row-parallel one: the same bytes over xGMI as the all-reduce pair)
        self.seq_par = bool(sequence_parallel) and tp_size > 1
        assert not (self.seq_par and (cfg.n_experts > 0 or not cfg.mlp_gated or cfg.linear_bias)), \
            "sequence parallelism is implemented for gated dense MLPs without output biases"
        self.cfg = cfg
        self.dtype = dtype
        self.tp_size, self.tp_rank, self.tp_group = tp_size, tp_rank, tp_group
        self.ep_group = ep_group
        if layer_chunks is not None:
            assert layer_range is None and len(layer_chunks) >= 1
            self.layer_chunks = [tuple(c) for c in layer_chunks]
        else:
            self.layer_chunks = [tuple(layer_range or (0, cfg.n_layers))]
        # contiguous stages: the slice itself; interleaved stages: first .. last layer hosted (not contiguous)
        self.layer_range = (self.layer_chunks[0][0], self.layer_chunks[-1][1])
        self.layer_ids: List[int] = [i for lo, hi in self.layer_chunks for i in range(lo, hi)]
        bounds, at = [], 0
        for lo, hi in self.layer_chunks:
            bounds.append((at, at + hi - lo))
            at += hi - lo
        self.chunk_bounds = bounds  # local layer index range of every chunk
        self.has_embed, self.has_head = has_embed, has_head
        self.gradient_checkpointing = gradient_checkpointing
        self.sp = None  # Ulysses sequence-parallel context (parallel/ulysses.py)
        if has_embed:
            self.embed = nn.Parameter(torch.empty(v_rows, cfg.hidden_size, dtype=dtype, device=device))
            if cfg.pos_embed == "learned":
                self.pos_embed = nn.Parameter(torch.empty(cfg.max_position_embeddings, cfg.hidden_size, dtype=dtype,
                                                          device=device))
        self.layers = nn.ModuleList(
            [Block(cfg, tp_size, dtype, device, ep_size, ep_rank, etp_size) for _ in self.layer_ids]
        )
        for blk, gid in zip(self.layers, self.layer_ids):
            blk.layer_id = gid
        if has_head:
            self.norm = nn.Parameter(torch.ones(cfg.hidden_size, dtype=dtype, device=device))
            if cfg.norm_type == "layer":
                self.norm_bias = nn.Parameter(torch.zeros(cfg.hidden_size, dtype=dtype, device=device))
            if cfg.is_critic:
                self.head = nn.Parameter(torch.empty(1, cfg.hidden_size, dtype=dtype, device=device))
            elif not (cfg.tie_word_embeddings and has_embed):
                self.head = nn.Parameter(torch.empty(v_rows, cfg.hidden_size, dtype=dtype, device=device))
        if cfg.vision is not None and has_embed:
            from .vision import VisionConfig, VisionTower

            self.visual = VisionTower(VisionConfig.from_dict(cfg.vision), dtype=dtype, device=device)
        self._rope: Optional[ops.RotaryTable] = None
        self.flat: Optional[FlatParams] = None
        self.flat_experts: Optional[FlatParams] = None

    # ---------------------------------------------------------------------------- set-up
    def init_weights(self, seed: int = 0, std: float = 0.02):
        g = torch.Generator(device="cpu").manual_seed(seed)
        for n, p in self.named_parameters():
            if n.startswith("visual.") and p.dim() == 1:
                p.data.fill_(1.0 if n.endswith("weight") else 0.0)  # LayerNorm weights / biases of the tower
            elif p.dim() >= 2:
                # generate in fp32 on host in slabs to bound memory, same values on every device type
                flat = p.data.view(-1)
                step = 1 << 24
                for s in range(0, flat.numel(), step):
                    e = min(flat.numel(), s + step)
                    flat[s:e].copy_(torch.randn(e - s, generator=g) * std)
            elif n.endswith("bias"):
                p.data.zero_()
            else:
                p.data.fill_(0.0 if self.cfg.gemma_norm else 1.0)

    def flatten(self, pad_to: int = 1, expert_pad_to: Optional[int] = None) -> FlatParams:
        """Move parameters into flat storage. With `expert_pad_to` (expert parallelism) the routed-expert
        weights get their own buffer `flat_experts`: they are replicated over a different group (expert
        data parallel) than the dense weights, so their gradients are reduced separately."""
        dev = next(self.parameters()).device
        if expert_pad_to is None or self.cfg.n_experts == 0:
            self.flat = FlatParams(self, self.dtype, dev, pad_to)
            self.flat_experts = None
        else:
            named = list(self.named_parameters())
            is_exp = lambda n: n.endswith("mlp.gate_up") or n.endswith("mlp.down")  # noqa: E731
            self.flat = FlatParams(self, self.dtype, dev, pad_to, [(n, q) for n, q in named if not is_exp(n)])
            self.flat_experts = FlatParams(self, self.dtype, dev, expert_pad_to, [(n, q) for n, q in named if is_exp(n)])
        return self.flat

    def rope(self, device) -> ops.RotaryTable:
        if self._rope is None or self._rope.cos.device != torch.device(device):
            self._rope = ops.RotaryTable(self.cfg.head_dim, self.cfg.max_position_embeddings, self.cfg.rope_theta,
                                         device, self.cfg.rope_scaling)
        return self._rope

    @property
    def head_weight(self) -> torch.Tensor:
        if hasattr(self, "head"):
            return self.head
        return self.embed

    # --------------------------------------------------------------------------- forward
    def embed_tokens(self, input_ids, positions=None, scatter: bool = False):
        """Token (+ learned position) embeddings [T, H]; with `scatter` (sequence parallelism) this rank's
        token slice [T/tp, H]. Vocab-parallel shards look up their own rows (zero for foreign ids) and the
        partial results are summed: an all-reduce, or a reduce-scatter when the result is token-sharded
        anyway (reference ``realhf/impl/model/parallelism/tensor_parallel/modules.py:57-139``)."""
        from ..parallel import tensor_parallel as tpar

        if self.vocab_parallel:
            lo, hi = self.vocab_range
            foreign = (input_ids < lo) | (input_ids >= hi)
            h = ops.embedding((input_ids - lo).masked_fill(foreign, 0), self.embed)
            h = h.masked_fill(foreign.unsqueeze(-1), 0.0)
            h = (tpar.ReduceScatterToSP if scatter else tpar.ReduceFromTP).apply(h, self.tp_group)
        else:
            h = ops.embedding(input_ids, self.embed)
            if scatter:
                h = tpar.ScatterToSP.apply(h, self.tp_group)
        if self.cfg.gemma_norm:
            h = h * math.sqrt(self.cfg.hidden_size)
        if self.cfg.pos_embed == "learned":
            h = h + ops.embedding(positions.long(), self.pos_embed)
        if self.cfg.embd_pdrop > 0:
            h = rng.dropout(h, self.cfg.embd_pdrop, -1, 0, sharded=scatter, training=self.training)
        return h

    def embed_multimodal(self, input_ids: torch.Tensor, pixel_values: Optional[torch.Tensor],
                         grid_thw: Optional[torch.Tensor], mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Token embeddings with the image placeholder positions (`mask`, default: every placeholder id)
        replaced, in order, by the vision tower's outputs for `pixel_values` (all images of the packed
        batch concatenated)."""
        h = self.embed_tokens(input_ids)
        if pixel_values is None or pixel_values.numel() == 0:
            return h
        img = self.visual(pixel_values.to(h.device), grid_thw.to(h.device)).to(h.dtype)
        if mask is None:
            mask = (input_ids == self.cfg.image_token_id)
            if self.cfg.video_token_id is not None:
                mask = mask | (input_ids == self.cfg.video_token_id)
        mask = mask.to(h.device)
        n = int(mask.sum())
        assert n == img.shape[0], f"{n} image placeholder tokens but {img.shape[0]} image embeddings"
        return h.masked_scatter(mask.unsqueeze(-1), img)

    def mrope(self, positions3: torch.Tensor) -> ops.RotaryTable:
        """Per-token rotary table for 3-D positions [3, T] (index it with arange(T))."""
        from .vision import mrope_table

        tab = ops.RotaryTable.__new__(ops.RotaryTable)
        tab.head_dim, tab.max_pos = self.cfg.head_dim, positions3.shape[1]
        tab.cos, tab.sin = mrope_table(positions3, self.cfg.head_dim, self.cfg.rope_theta,
                                       self.cfg.mrope_section or [self.cfg.head_dim // 2, 0, 0])
        return tab

    def forward(self, input_ids: Optional[torch.Tensor], positions: torch.Tensor, cu_seqlens: torch.Tensor,
                max_seqlen: int, hidden_in: Optional[torch.Tensor] = None,
                residual_in: Optional[torch.Tensor] = None, chunk: Optional[int] = None):
        """Returns the final-norm hidden states [T, H] on the last stage, else (h, residual). `chunk` selects
        one of the rank's layer slices (interleaved pipeline): only the first slice of the first stage embeds,
        only the last slice of the last stage applies the final norm."""
        n_chunks = len(self.chunk_bounds)
        first_chunk = chunk is None or chunk == 0
        last_chunk = chunk is None or chunk == n_chunks - 1
        layers = self.layers if chunk is None else self.layers[self.chunk_bounds[chunk][0]:self.chunk_bounds[chunk][1]]
        if self.has_embed and first_chunk and hidden_in is None:
            h = self.embed_tokens(input_ids, positions, scatter=self.seq_par)
        else:
            h = hidden_in
        residual = residual_in
        if self.cfg.pos_embed != "rope":
            rope = None
        elif positions.dim() == 2:  # M-RoPE: [3, T] (temporal, row, column) positions of a vision-language batch
            rope = self.mrope(positions)
            positions = torch.arange(positions.shape[1], dtype=torch.int32, device=positions.device)
        else:
            rope = self.rope(h.device)
        for blk in layers:
            if self.gradient_checkpointing and self.training and torch.is_grad_enabled():
                from torch.utils.checkpoint import checkpoint

                h, residual = checkpoint(blk, h, residual, positions, rope, cu_seqlens, max_seqlen, self.tp_group,
                                         self.sp, self.ep_group, self.seq_par, use_reentrant=False)
            else:
                h, residual = blk(h, residual, positions, rope, cu_seqlens, max_seqlen, self.tp_group, self.sp,
                                  self.ep_group, self.seq_par)
        if not (self.has_head and last_chunk):
            return h, residual
        x, _ = add_norm(self.cfg, h, self.norm, getattr(self, "norm_bias", None), residual)
        if self.seq_par:  # the LM head / value head sees all tokens again
            from ..parallel.tensor_parallel import GatherForReplicatedUse, GatherFromSP

            # a vocab-sharded LM head returns a partial d(hidden) per rank: the gather's backward sums them
            # (reduce-scatter); a replicated head returns the same gradient everywhere: keep the own slice
            sharded_head = self.vocab_parallel and not self.cfg.is_critic
            x = (GatherFromSP if sharded_head else GatherForReplicatedUse).apply(x, self.tp_group)
        return x

    def logits(self, hidden: torch.Tensor) -> torch.Tensor:
        """Full-vocabulary logits [T, V] (vocab shards are all-gathered along the vocabulary)."""
        if not self.vocab_parallel:
            return F.linear(hidden, self.head_weight)
        from ..parallel import tensor_parallel as tpar

        if not self.seq_par:
            hidden = tpar.CopyToTP.apply(hidden, self.tp_group)
        return tpar.GatherVocab.apply(F.linear(hidden, self.head_weight), self.tp_group, self.cfg.vocab_size)

    def values(self, hidden: torch.Tensor) -> torch.Tensor:
        assert self.cfg.is_critic
        return F.linear(hidden, self.head).squeeze(-1).float()

    def logprobs(self, hidden, labels, temperature=1.0, chunk=8192, keep_logits=True):
        """Fused LM head + log-prob + entropy; never materialises fp32 [T, V]."""
        if self.vocab_parallel:
            return ops.lm_head_logprobs(hidden, self.head_weight, labels, temperature, chunk, keep_logits,
                                        vocab_start=self.vocab_range[0], group=self.tp_group,
                                        reduce_dh=not self.seq_par)
        return ops.lm_head_logprobs(hidden, self.head_weight, labels, temperature, chunk, keep_logits)

    def moe_aux_loss(self):
        tot = None
        for blk in self.layers:
            a = getattr(blk.mlp, "aux_loss", None)
            if a is not None:
                tot = a if tot is None else tot + a
        return tot
