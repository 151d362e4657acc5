"""Fused temperature / top-k / top-p sampling returning (token, logprob).

GPU: ``csrc/hip/sampling.hip`` (Gumbel-max, bisection thresholds, Philox keyed by a device-resident
step counter so captured graphs draw fresh noise on replay). CPU: PyTorch reference.
Counterparts: ``realhf/impl/model/utils/logits_warper.py:52-125,207-225`` and ``genstep`` in
``realhf/impl/model/nn/real_llm_generate.py:30-146``.
"""
from typing import Optional, Tuple

import torch

from . import _lib as L


def sample_ref(logits, temperature, top_k, top_p, generator=None):
    B, V = logits.shape
    toks = torch.empty(B, dtype=torch.int64, device=logits.device)
    lps = torch.empty(B, dtype=torch.float32, device=logits.device)
    for b in range(B):
        t = float(temperature[b]) if temperature is not None else 1.0
        row = logits[b].float()
        if t <= 0:
            tok = int(row.argmax())
            lps[b] = torch.log_softmax(row, -1)[tok]
            toks[b] = tok
            continue
        z = row / t
        full_lp = torch.log_softmax(z, -1)
        k = int(top_k[b]) if top_k is not None else V
        p = float(top_p[b]) if top_p is not None else 1.0
        zz = z.clone()
        if 1 <= k < V:
            kth = torch.topk(zz, k).values[-1]
            zz[zz < kth] = float("-inf")
        if p < 1.0:
            sp, si = torch.sort(torch.softmax(zz, -1), descending=True)
            cum = torch.cumsum(sp, 0)
            drop = (cum - sp) >= p
            zz[si[drop]] = float("-inf")
        tok = int(torch.multinomial(torch.softmax(zz, -1), 1, generator=generator))
        toks[b] = tok
        lps[b] = full_lp[tok]
    return toks, lps


def sample(logits: torch.Tensor, temperature: Optional[torch.Tensor], top_k: Optional[torch.Tensor],
           top_p: Optional[torch.Tensor], seed: int, step: Optional[torch.Tensor] = None, bump_step: bool = True,
           out_tokens: Optional[torch.Tensor] = None, out_logprobs: Optional[torch.Tensor] = None
           ) -> Tuple[torch.Tensor, torch.Tensor]:
    """logits [B, V] (bf16/fp32, row stride arbitrary); temperature f32 [B] (<= 0 -> greedy), top_k int32 [B],
    top_p f32 [B]; `step` is a 1-element int64 device tensor (incremented after sampling when bump_step)."""
    B, V = logits.shape
    if not L.use_hip(logits):
        g = torch.Generator(device=logits.device)
        g.manual_seed(int(seed) + (int(step[0]) if step is not None else 0))
        t, l = sample_ref(logits, temperature, top_k, top_p, g)
        if step is not None and bump_step:
            step += 1
        if out_tokens is not None:
            out_tokens.copy_(t)
            out_logprobs.copy_(l)
            return out_tokens, out_logprobs
        return t, l
    assert logits.stride(1) == 1
    if out_tokens is None:
        out_tokens = torch.empty(B, dtype=torch.int64, device=logits.device)
    if out_logprobs is None:
        out_logprobs = torch.empty(B, dtype=torch.float32, device=logits.device)
    L.check(
        L.lib().areal_sample(L.ptr(logits), L.c_int(L.dtype_flag(logits)), L.c_int(B), L.c_int(V),
                             L.c_int64(logits.stride(0)), L.ptr(temperature), L.ptr(top_k), L.ptr(top_p),
                             L.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), L.ptr(step), L.c_int(int(bump_step)),
                             L.ptr(out_tokens), L.ptr(out_logprobs), L.stream_ptr()),
        "sample",
    )
    return out_tokens, out_logprobs


def advance_step_(sampled, input_ids, positions, context_lens, slots, block_tables, active=None):
    """Device-side bookkeeping between two decode steps (see paged_attn.hip::advance_step_kernel)."""
    B = input_ids.numel()
    if L.use_hip(input_ids):
        L.check(
            L.lib().areal_advance_step(L.ptr(sampled), L.ptr(input_ids), L.ptr(positions), L.ptr(context_lens),
                                       L.ptr(slots), L.ptr(block_tables), L.ptr(active), L.c_int(B),
                                       L.c_int(block_tables.shape[1]), L.stream_ptr()),
            "advance_step",
        )
        return
    from .attention import PAGE_SIZE

    act = torch.ones(B, dtype=torch.bool) if active is None else active.bool()
    input_ids[act] = sampled[act]
    positions[act] += 1
    context_lens[act] = positions[act] + 1
    pg = (positions // PAGE_SIZE).long().clamp(max=block_tables.shape[1] - 1)
    new_slots = block_tables.gather(1, pg.unsqueeze(1)).squeeze(1).long() * PAGE_SIZE + positions.long() % PAGE_SIZE
    slots[act] = new_slots[act]
