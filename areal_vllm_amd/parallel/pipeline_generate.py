"""Generation through the stages of a pipeline-parallel model (and its tensor-parallel ranks).

A batch of prompts is cut into micro-batches that travel round the ring of stages: stage 0 embeds the tokens of
micro-batch m and runs its layers, every stage passes one stacked ``[2, T, H]`` tensor (block output + residual
stream) to its successor, the last stage applies the final norm and the LM head, samples, and sends the ``[b]``
new tokens back to stage 0, which feeds them into the next decode step of that micro-batch. With at least as
many micro-batches as stages every stage is busy while a token makes its way round - the schedule the
reference encodes as ``GenerateSchedule`` instructions (``realhf/impl/model/parallelism/pipeline_parallel/
static_schedule.py:199-280``) and executes in ``realhf/impl/model/backend/pipe_runner.py:420-646``; here it
is a plain loop, identical on every stage, over (decode step, micro-batch).

Every stage keeps the keys / values of ITS layers in the paged cache layout of the rollout engine
(``ops.alloc_kv_cache``), pages handed out sequentially per sequence, so prefill runs the varlen MFMA attention
and decode the split-KV paged kernel. Sends are asynchronous on per-link two-rank communicators
(``ProcessGrid._build_pipeline_links``): on every link the order of sends equals the order of receives, and no
rank ever waits for a send to be matched, so the ring cannot dead-lock. Whether every sequence has finished is
decided by the last stage and broadcast over the pipeline group every ``check_interval`` steps; rows that
finish in between keep decoding and are cut afterwards.

The rollout servers never generate this way (one MI355X holds any model this framework trains, larger ones are
served tensor-parallel); this path lets a pipeline-parallel TRAINER generate with the weights it holds, e.g.
the synchronous PPO experiment with ``actor`` on a ``p2`` layout.
"""
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
import torch.nn.functional as F

from .. import ops
from ..ops.attention import PAGE_SIZE


@dataclass
class GeneratedSequence:
    output_tokens: List[int] = field(default_factory=list)
    output_logprobs: List[float] = field(default_factory=list)
    finish_reason: str = "length"  # "stop" when a stop token ended the sequence


class _Links:
    """Asynchronous sends / blocking receives of one stage: `fwd` = activations to the next stage,
    `ring` = tokens from the last stage to the first."""

    def __init__(self, grid):
        self.g = grid
        self.pending: List[Tuple[Any, torch.Tensor]] = []

    def send_next(self, t: torch.Tensor):
        t = t.contiguous()
        self.pending.append((dist.isend(t, dst=self.g.ring_next, group=self.g.links["send_fwd"]), t))

    def recv_prev(self, shape, dtype, device) -> torch.Tensor:
        buf = torch.empty(shape, dtype=dtype, device=device)
        dist.irecv(buf, src=self.g.ring_prev, group=self.g.links["recv_fwd"]).wait()
        return buf

    def reap(self):
        """Forget sends that have completed (keeps the list short over a long generation)."""
        self.pending = [(w, t) for w, t in self.pending if not w.is_completed()]

    def drain(self):
        for w, _t in self.pending:
            w.wait()
        self.pending.clear()


class PipelineGenerator:
    """Generates with the model slice of ONE rank; every rank of a (pipeline x tensor)-parallel replica calls
    `generate` with the same prompts and gets the same result."""

    def __init__(self, model, grid, device, seed: int = 1):
        cfg = model.cfg
        if cfg.norm_type != "rms" or cfg.pos_embed != "rope" or not cfg.mlp_gated or cfg.linear_bias:
            raise NotImplementedError(f"pipelined generation covers the RMSNorm / rotary / gated-MLP families, "
                                      f"not {cfg.family}")
        if len(model.layer_chunks) != 1:
            raise NotImplementedError("pipelined generation needs contiguous pipeline stages (no virtual stages)")
        self.model, self.grid, self.device, self.seed = model, grid, torch.device(device), int(seed)
        self.cfg = cfg
        self.pp = grid.pp
        self.stage = grid.coord.pp
        self.first, self.last = self.stage == 0, self.stage == grid.pp - 1
        self.hq = model.layers[0].attn.hq if len(model.layers) else cfg.n_q_heads // model.tp_size
        self.hkv = model.layers[0].attn.hkv if len(model.layers) else max(1, cfg.n_kv_heads // model.tp_size)
        self.links = None
        if self.pp > 1:
            if not grid.links:  # collective over the world: every rank constructs its generator at the same point
                grid._build_pipeline_links()
            self.links = _Links(grid)
        self.step_counter = torch.zeros(1, dtype=torch.int64, device=self.device)

    # ------------------------------------------------------------------------------ model pieces
    def _layer(self, blk, h, residual, attn_fn):
        cfg, m = self.cfg, self.model
        x, residual = ops.rmsnorm(h, blk.ln1, cfg.rms_norm_eps, residual, cfg.gemma_norm)
        attn = attn_fn(blk.attn.project_qkv(x))
        h = F.linear(attn.reshape(attn.shape[0], -1), blk.attn.o)
        if m.tp_size > 1:
            dist.all_reduce(h, group=m.tp_group)
        x, residual = ops.rmsnorm(h, blk.ln2, cfg.rms_norm_eps, residual, cfg.gemma_norm)
        if cfg.n_experts > 0:
            return blk.mlp(x, m.tp_group, m.ep_group), residual
        h = F.linear(ops.swiglu(F.linear(x, blk.mlp.gate_up), blk.mlp.act), blk.mlp.down)
        if m.tp_size > 1:
            dist.all_reduce(h, group=m.tp_group)
        return h, residual

    def _stage_in(self, ids: torch.Tensor, T: int):
        if self.first:
            return self.model.embed_tokens(ids), None
        buf = self.links.recv_prev((2, T, self.cfg.hidden_size), self.model.dtype, self.device)
        return buf[0], buf[1]

    def _sample(self, h, residual, temp, top_k, top_p, filt) -> Tuple[torch.Tensor, torch.Tensor]:
        cfg, m = self.cfg, self.model
        x, _ = ops.rmsnorm(h, m.norm, cfg.rms_norm_eps, residual, cfg.gemma_norm)
        logits = m.logits(x)
        tok, lp = ops.sample(logits, temp, top_k if filt else None, top_p if filt else None, self.seed,
                             self.step_counter)
        if m.tp_size > 1:  # identical logits and keys on every tp rank; the broadcast makes agreement unconditional
            src = dist.get_global_rank(m.tp_group, 0)
            dist.broadcast(tok, src=src, group=m.tp_group)
            dist.broadcast(lp, src=src, group=m.tp_group)
        return tok, lp

    # ---------------------------------------------------------------------------------- generate
    @torch.no_grad()
    def generate(self, prompts: Sequence[Sequence[int]], max_new_tokens: int, min_new_tokens: int = 0,
                 temperature: float = 1.0, top_p: float = 1.0, top_k: int = 1 << 30, greedy: bool = False,
                 stop_token_ids: Sequence[int] = (), n_micro_batches: Optional[int] = None,
                 check_interval: int = 4) -> List[GeneratedSequence]:
        dev, cfg, m = self.device, self.cfg, self.model
        B = len(prompts)
        assert B > 0 and max_new_tokens >= 1 and all(len(p) > 0 for p in prompts)
        n_mb = max(1, min(B, n_micro_batches or self.pp))
        bounds = [(B * i) // n_mb for i in range(n_mb + 1)]
        V = cfg.vocab_size
        filt = (not greedy) and (top_k < V or top_p < 1.0)
        stops = torch.tensor(sorted(set(int(s) for s in stop_token_ids)), dtype=torch.int64, device=dev)
        hq, hkv, D = self.hq, self.hkv, cfg.head_dim
        rope = m.rope(dev)

        # ---- per micro-batch state: sequential pages per sequence, the same on every stage
        mbs: List[Dict[str, Any]] = []
        page_base = 0
        for i in range(n_mb):
            rows = list(range(bounds[i], bounds[i + 1]))
            lens = [len(prompts[r]) for r in rows]
            pps = [(n + max_new_tokens + PAGE_SIZE - 1) // PAGE_SIZE for n in lens]
            bt = torch.zeros(len(rows), max(pps), dtype=torch.int32)
            for j, n in enumerate(pps):
                bt[j, :n] = page_base + torch.arange(n, dtype=torch.int32)
                page_base += n
            b = len(rows)
            mbs.append(dict(rows=rows, lens=torch.tensor(lens, dtype=torch.int32, device=dev), bt=bt.to(dev), b=b,
                            temp=torch.full((b,), 0.0 if greedy else float(temperature), device=dev),
                            top_k=torch.full((b,), min(int(top_k), V), dtype=torch.int32, device=dev),
                            top_p=torch.full((b,), float(top_p), device=dev),
                            toks=[], lps=[]))
        kv = [ops.alloc_kv_cache(page_base, hkv, D, m.dtype, dev) for _ in m.layers]

        def slots_of(mb, pos):  # pos int32 [b] -> int64 slot ids
            page = mb["bt"].gather(1, (pos // PAGE_SIZE).long().unsqueeze(1)).squeeze(1).long()
            return page * PAGE_SIZE + (pos % PAGE_SIZE).long()

        # ---- prefill: whole prompts, packed
        for mb in mbs:
            lens = mb["lens"]
            T = int(lens.sum())
            cu = torch.zeros(mb["b"] + 1, dtype=torch.int32, device=dev)
            cu[1:] = torch.cumsum(lens, 0)
            ids = torch.tensor([t for r in mb["rows"] for t in prompts[r]], dtype=torch.long, device=dev)
            pos = torch.cat([torch.arange(int(n), dtype=torch.int32, device=dev) for n in lens.tolist()])
            row_of = torch.repeat_interleave(torch.arange(mb["b"], device=dev), lens.long())
            page = mb["bt"][row_of, (pos // PAGE_SIZE).long()].long()
            slots = page * PAGE_SIZE + (pos % PAGE_SIZE).long()
            max_q = int(lens.max())
            h, residual = self._stage_in(ids, T)
            for li, blk in enumerate(m.layers):
                kc, vc = kv[li]

                def attn_fn(qkv2d, kc=kc, vc=vc):
                    qkv = qkv2d.view(T, hq + 2 * hkv, D).contiguous()
                    ops.rope_and_cache_write_(qkv, pos, slots, rope, kc, vc, hq, hkv)
                    out, _ = ops.flash_attn_varlen(qkv[:, :hq], qkv[:, hq:hq + hkv], qkv[:, hq + hkv:], cu, cu, max_q)
                    return out

                h, residual = self._layer(blk, h, residual, attn_fn)
            if self.last:
                last_tok = (cu[1:] - 1).long()
                res_last = residual[last_tok] if residual is not None else None
                tok, lp = self._sample(h[last_tok], res_last, mb["temp"], mb["top_k"], mb["top_p"], filt)
                mb["toks"].append(tok)
                mb["lps"].append(lp)
                if self.pp > 1 and max_new_tokens > 1:
                    self.links.send_next(tok)  # the last stage's successor on the ring is stage 0
            else:
                self.links.send_next(torch.stack([h, residual if residual is not None else torch.zeros_like(h)]))

        # ---- decode
        done = torch.zeros(1, dtype=torch.int32, device=dev)
        last_rank = self.grid.ranks[self.grid.index_of(self.pp - 1, self.grid.coord.dp, self.grid.coord.cp,
                                                       self.grid.coord.tp)] if self.pp > 1 else None
        finished = [torch.zeros(mb["b"], dtype=torch.bool, device=dev) for mb in mbs] if self.last else None
        n_generated = 1
        for t in range(1, max_new_tokens):
            if self.last and t % check_interval == 0:
                for mb, fin in zip(mbs, finished):
                    if t > min_new_tokens and stops.numel():
                        # tokens 0 .. t-1 exist; those at index >= min_new_tokens may stop the sequence
                        recent = torch.stack(mb["toks"][max(min_new_tokens, t - check_interval):t], 1)
                        fin |= torch.isin(recent, stops).any(1)
                done.fill_(int(all(bool(f.all()) for f in finished)))
            if self.pp > 1 and t % check_interval == 0:
                dist.broadcast(done, src=last_rank, group=self.grid.group("pp"))
            if t % check_interval == 0 and int(done):  # the only host synchronisation of the decode loop
                break
            for mb in mbs:
                b = mb["b"]
                pos = mb["lens"] + (t - 1)  # position of the token generated at step t - 1
                ctx = pos + 1
                slots = slots_of(mb, pos)
                if self.first:
                    ids = mb["toks"][-1] if self.last else self.links.recv_prev((b,), torch.int64, dev)
                else:
                    ids = None
                h, residual = self._stage_in(ids, b)
                for li, blk in enumerate(m.layers):
                    kc, vc = kv[li]

                    def attn_fn(qkv2d, kc=kc, vc=vc):
                        qkv = qkv2d.view(b, hq + 2 * hkv, D).contiguous()
                        ops.rope_and_cache_write_(qkv, pos, slots, rope, kc, vc, hq, hkv)
                        return ops.paged_attention_decode(qkv[:, :hq].contiguous(), kc, vc, mb["bt"], ctx)

                    h, residual = self._layer(blk, h, residual, attn_fn)
                if self.last:
                    tok, lp = self._sample(h, residual, mb["temp"], mb["top_k"], mb["top_p"], filt)
                    mb["toks"].append(tok)
                    mb["lps"].append(lp)
                    if self.pp > 1 and t + 1 < max_new_tokens:
                        self.links.send_next(tok)
                else:
                    self.links.send_next(torch.stack([h, residual]))
            n_generated = t + 1
            if self.links is not None:
                self.links.reap()

        # ---- results: assembled on the last stage, handed to the other stages as a host object
        out: List[Optional[GeneratedSequence]] = [None] * B
        if self.last:
            stop_set = set(stops.tolist())
            for mb in mbs:
                toks = torch.stack(mb["toks"], 1).tolist()
                lps = torch.stack(mb["lps"], 1).tolist()
                for j, r in enumerate(mb["rows"]):
                    n, reason = len(toks[j]), "length"
                    for k in range(min_new_tokens, len(toks[j])):
                        if toks[j][k] in stop_set:
                            n, reason = k + 1, "stop"
                            break
                    out[r] = GeneratedSequence(toks[j][:n], lps[j][:n], reason)
        if self.pp > 1:
            if self.links is not None:
                self._finish_ring(mbs, n_generated, max_new_tokens)
            box = [out if self.last else None]
            dist.broadcast_object_list(box, src=last_rank, group=self.grid.group("pp"))
            out = box[0]
        return out  # type: ignore[return-value]

    def _finish_ring(self, mbs, n_generated: int, max_new_tokens: int):
        """Receive what is still on its way: when the loop ends early the last stage has already sent the tokens of
        the final step to stage 0, which will not feed them into another step."""
        if self.first and not self.last and n_generated < max_new_tokens:
            for mb in mbs:
                self.links.recv_prev((mb["b"],), torch.int64, self.device)
        self.links.drain()
