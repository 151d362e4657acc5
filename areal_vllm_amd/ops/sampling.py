"""Fused temperature / top-k / top-p sampling returning (token, logprob).

GPU: ``csrc/hip/sampling.hip`` (Gumbel-max, bisection thresholds, Philox keyed by a device-resident
step counter so captured graphs draw fresh noise on replay). CPU: PyTorch reference.
Counterparts: ``realhf/impl/model/utils/logits_warper.py:52-125,207-225`` and ``genstep`` in
``realhf/impl/model/nn/real_llm_generate.py:30-146``.
"""
import os
from typing import Optional, Tuple

import torch

from . import _lib as L

# unfiltered batches (no top-k / top-p) split every row over several workgroups; 0 keeps the one-workgroup-per-row
# kernel for A/B timing
SPLIT_UNFILTERED = os.environ.get("AREAL_SAMPLE_SPLIT", "1") != "0"


class TokenCounts:
    """Token-count table for frequency / presence penalties: int16 [slots, ld] on the sampling device, one
    row per request that asked for a penalty (the scheduler hands the row index out as `penalty_slot`).
    The sampler reads a row to adjust the logits and bumps the count of the token it draws."""

    def __init__(self, slots: int, vocab: int, device):
        self.ld = (vocab + 7) // 8 * 8  # 16-byte aligned rows
        self.table = torch.zeros(slots, self.ld, dtype=torch.int16, device=device)

    def clear(self, slots):
        if len(slots):
            self.table[torch.as_tensor(list(slots), dtype=torch.int64, device=self.table.device)] = 0


def sample_ref(logits, temperature, top_k, top_p, generator=None, active=None, counts=None, count_slot=None,
               freq_pen=None, pres_pen=None):
    B, V = logits.shape
    toks = torch.zeros(B, dtype=torch.int64, device=logits.device)
    lps = torch.zeros(B, dtype=torch.float32, device=logits.device)
    for b in range(B):
        if active is not None and not int(active[b]):
            continue
        t = float(temperature[b]) if temperature is not None else 1.0
        row = logits[b].float()
        slot = int(count_slot[b]) if counts is not None and count_slot is not None else -1
        if slot >= 0:
            c = counts.table[slot, :V].float()
            row = row - float(freq_pen[b]) * c - float(pres_pen[b]) * (c > 0).float()
        if t <= 0:
            tok = int(row.argmax())
            lps[b] = torch.log_softmax(row, -1)[tok]
            toks[b] = tok
            if slot >= 0:
                counts.table[slot, tok] += 1
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
        if slot >= 0:
            counts.table[slot, tok] += 1
    return toks, lps


def sample(logits: torch.Tensor, temperature: Optional[torch.Tensor], top_k: Optional[torch.Tensor],
           top_p: Optional[torch.Tensor], seed: int, step: Optional[torch.Tensor] = None, bump_step: bool = True,
           out_tokens: Optional[torch.Tensor] = None, out_logprobs: Optional[torch.Tensor] = None,
           active: Optional[torch.Tensor] = None, counts: Optional[TokenCounts] = None,
           count_slot: Optional[torch.Tensor] = None, freq_pen: Optional[torch.Tensor] = None,
           pres_pen: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """logits [B, V] (bf16/fp32, row stride arbitrary); temperature f32 [B] (<= 0 -> greedy), top_k int32 [B],
    top_p f32 [B]; `step` is a 1-element int64 device tensor (incremented after sampling when bump_step).
    `active` int32 [B]: rows with 0 are skipped (token 0). `counts` + `count_slot` int32 [B] (-1: none) +
    `freq_pen` / `pres_pen` f32 [B]: rows with a slot sample from logit - freq * count - pres * (count > 0)
    and bump the count of the drawn token."""
    B, V = logits.shape
    if not L.use_hip(logits):
        g = torch.Generator(device=logits.device)
        g.manual_seed(int(seed) + (int(step[0]) if step is not None else 0))
        t, l = sample_ref(logits, temperature, top_k, top_p, g, active, counts, count_slot, freq_pen, pres_pen)
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
    if top_k is None and top_p is None and SPLIT_UNFILTERED:
        # no filtering anywhere in the batch: every row is reduced by several workgroups (same tokens as below)
        lib = L.lib()
        S = int(lib.areal_sample_splits(L.c_int(B)))
        ws = torch.empty(B * S * 4, dtype=torch.float32, device=logits.device)
        L.check(
            lib.areal_sample_unfiltered(L.ptr(logits), L.c_int(L.dtype_flag(logits)), L.c_int(B), L.c_int(V),
                                        L.c_int64(logits.stride(0)), L.ptr(temperature),
                                        L.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), L.ptr(step),
                                        L.c_int(int(bump_step)), L.ptr(out_tokens), L.ptr(out_logprobs), L.ptr(active),
                                        L.ptr(counts.table if counts is not None else None),
                                        L.c_int64(counts.ld if counts is not None else 0), L.ptr(count_slot),
                                        L.ptr(freq_pen), L.ptr(pres_pen), L.ptr(ws), L.c_int(S), L.stream_ptr()),
            "sample_unfiltered",
        )
        return out_tokens, out_logprobs
    L.check(
        L.lib().areal_sample(L.ptr(logits), L.c_int(L.dtype_flag(logits)), L.c_int(B), L.c_int(V),
                             L.c_int64(logits.stride(0)), L.ptr(temperature), L.ptr(top_k), L.ptr(top_p),
                             L.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), L.ptr(step), L.c_int(int(bump_step)),
                             L.ptr(out_tokens), L.ptr(out_logprobs), L.ptr(active),
                             L.ptr(counts.table if counts is not None else None),
                             L.c_int64(counts.ld if counts is not None else 0), L.ptr(count_slot), L.ptr(freq_pen),
                             L.ptr(pres_pen), L.stream_ptr()),
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


def decode_advance_(sampled, input_ids, positions, context_lens, slots, block_tables, active, remaining,
                    min_remaining, stop_ids):
    """Bookkeeping after one sampled token of a multi-step decode plan, with early exit (see
    paged_attn.hip::decode_advance_kernel): a row whose token budget is used up, or that drew one of its stop
    ids once min_new_tokens are produced, retires (active 0, slot -1, context length 1); the others advance."""
    B = input_ids.numel()
    if L.use_hip(input_ids):
        L.check(
            L.lib().areal_decode_advance(L.ptr(sampled), L.ptr(input_ids), L.ptr(positions), L.ptr(context_lens),
                                         L.ptr(slots), L.ptr(block_tables), L.ptr(active), L.ptr(remaining),
                                         L.ptr(min_remaining), L.ptr(stop_ids), L.c_int(stop_ids.shape[1]), L.c_int(B),
                                         L.c_int(block_tables.shape[1]), L.stream_ptr()),
            "decode_advance",
        )
        return
    from .attention import PAGE_SIZE

    act = active.bool()
    remaining[act] -= 1
    min_remaining[act] = (min_remaining[act] - 1).clamp(min=0)
    hit = (stop_ids == sampled.to(stop_ids.dtype).unsqueeze(1)).any(1) & (min_remaining == 0)
    stop = act & ((remaining <= 0) | hit)
    go = act & ~stop
    active[stop] = 0
    slots[stop] = -1
    context_lens[stop] = 1
    input_ids[go] = sampled[go]
    positions[go] += 1
    context_lens[go] = positions[go] + 1
    pg = (positions // PAGE_SIZE).long()
    inside = pg < block_tables.shape[1]
    page = block_tables.gather(1, pg.clamp(max=block_tables.shape[1] - 1).unsqueeze(1)).squeeze(1).long()
    new_slots = torch.where(inside, page * PAGE_SIZE + positions.long() % PAGE_SIZE, torch.full_like(page, -1))
    slots[go] = new_slots[go]


def decode_begin_(kill, active, positions, context_lens, slots, block_tables):
    """Start of a decode plan that continues the previous one on the same rows: rows flagged in `kill` retire,
    live rows re-derive their KV slot from the refreshed page table."""
    B = active.numel()
    if L.use_hip(active):
        L.check(
            L.lib().areal_decode_begin(L.ptr(kill), L.ptr(active), L.ptr(positions), L.ptr(context_lens), L.ptr(slots),
                                       L.ptr(block_tables), L.c_int(B), L.c_int(block_tables.shape[1]), L.stream_ptr()),
            "decode_begin",
        )
        return
    from .attention import PAGE_SIZE

    active[kill.bool()] = 0
    dead = ~active.bool()
    slots[dead] = -1
    context_lens[dead] = 1
    pg = (positions // PAGE_SIZE).long()
    inside = pg < block_tables.shape[1]
    page = block_tables.gather(1, pg.clamp(max=block_tables.shape[1] - 1).unsqueeze(1)).squeeze(1).long()
    new_slots = torch.where(inside, page * PAGE_SIZE + positions.long() % PAGE_SIZE, torch.full_like(page, -1))
    live = active.bool()
    slots[live] = new_slots[live]
