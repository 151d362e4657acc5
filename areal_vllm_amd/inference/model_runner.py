"""GPU execution of scheduler plans for the native rollout engine.

* prefill: packed varlen forward (flash kernel) with fused RoPE + paged-cache write, then sampling of
  the first token of every request;
* decode: `n_steps` tokens per plan; one step = embedding -> L x (RMSNorm, fused qkv GEMM, fused
  RoPE + cache write, paged-KV MFMA attention, o GEMM, RMSNorm, gate|up GEMM, SwiGLU, down GEMM) ->
  final norm -> LM head -> fused sampler (penalties from the token-count table) -> device-side
  bookkeeping with early exit (`decode_advance_`: rows that used up their budget or drew a stop id
  retire inside the plan). The whole multi-step body is captured once per batch-size bucket into a
  hipGraph (``torch.cuda.CUDAGraph``) and replayed, so a decode plan costs one host launch (the
  reference relies on SGLang's CUDA-graph decode and on ``realhf/impl/model/utils/cuda_graph.py:69-220``
  for its own generator).
* decode plans are issued in two halves, `decode_launch` (uploads + graph replay + copy of the results
  into pinned memory + event) and `decode_wait`: the engine launches the CONTINUATION of a plan - same
  rows, device state carried over, only page tables and the kill mask uploaded - before it waits for
  the plan's own tokens, so the device never idles between two replays.

All block tables have the engine-wide maximum width, so graphs only depend on the batch bucket.
"""
import math
import os
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .. import ops
from ..models.config import ModelConfig
from ..models.transformer import Transformer
from ..ops.attention import PAGE_SIZE
from ..utils import logging

logger = logging.getLogger("model-runner")

DEFAULT_BUCKETS = [1, 2, 4, 8, 16, 24, 32, 48, 64, 96, 128, 160, 192, 256, 320, 384, 512, 768, 1024]


class ModelRunner:
    def __init__(self, model: Transformer, num_pages: int, max_context: int, max_batch: int, decode_steps: int = 1,
                 use_graph: bool = True, seed: int = 1, buckets: Optional[List[int]] = None,
                 fused_decode: Optional[bool] = None, n_stop_ids: int = 8, kv_dtype: Optional[torch.dtype] = None):
        self.model = model
        self.cfg: ModelConfig = model.cfg
        self.device = next(model.parameters()).device
        self.dtype = model.dtype
        self.num_pages = num_pages
        self.max_context = max_context
        self.max_pages = (max_context + PAGE_SIZE - 1) // PAGE_SIZE
        self.max_batch = max_batch
        self.decode_steps = decode_steps
        self.use_graph = use_graph and self.device.type == "cuda"
        self.seed = seed
        self.buckets = sorted(b for b in (buckets or DEFAULT_BUCKETS) if b <= max_batch)
        if not self.buckets or self.buckets[-1] < max_batch:
            self.buckets.append(max_batch)
        cfg = self.cfg
        self.hq = model.layers[0].attn.hq
        self.hkv = model.layers[0].attn.hkv
        # kv_dtype: torch.float8_e4m3fn stores the cache as e4m3 bytes (None: the model's dtype)
        self.kv = [ops.alloc_kv_cache(num_pages, self.hkv, cfg.head_dim, kv_dtype or self.dtype, self.device)
                   for _ in model.layers]
        if cfg.pos_embed == "rope":
            self.rope = model.rope(self.device)
        else:
            # learned absolute positions (GPT-2) are added to the embeddings; q and k go into attention and the
            # cache unrotated: the rotate-and-write kernels run on a table of angle zero
            self.rope = ops.RotaryTable(cfg.head_dim, 1, device=self.device)
            self.rope.cos.fill_(1.0)
            self.rope.sin.zero_()
        # LayerNorm with bias, projections with bias, the MLP without a gate (GPT-2): the unfused layer body
        self.plain_family = cfg.norm_type != "rms" or not cfg.mlp_gated or cfg.linear_bias or cfg.pos_embed != "rope"
        self.step_counter = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._gloo_group = None  # lazily: is the tp group a host backend (one-GPU rehearsal)
        # do the projections' outputs go through the tp group's all-reduce (and the sampled tokens through its
        # broadcast)? Also for a group of one when collectives are forced: RCCL calls inside the decode graph
        from ..parallel import comm as _comm

        self.tp_comm = model.tp_size > 1 or (_comm.FORCE_COLLECTIVES and torch.distributed.is_initialized())
        self.ipc_ar = None  # peer-memory all-reduce of the tp group (ops/ipc_allreduce.py), set by the engine
        self.graphs: Dict[int, Any] = {}
        self.static: Dict[int, Dict[str, torch.Tensor]] = {}
        self._graph_pool = None
        self.V = cfg.vocab_size
        self.n_prefill_tokens = 0
        self.n_decode_tokens = 0
        self.n_stop_ids = n_stop_ids  # stop ids per row checked on the device (the scheduler checks all of them)
        self.counts: Optional[ops.TokenCounts] = None  # frequency / presence penalties: allocated on first use
        self._launches = 0
        # vision-language requests: scheduler request id -> positions / image features (see register_multimodal)
        self.is_vl = cfg.vision is not None
        self.mm: Dict[int, Dict[str, Any]] = {}
        # Decode batches of FUSED_MIN_BATCH rows and more run the layer as seven launches (ops/decode_gemm.py);
        # model variants its epilogues do not cover (LayerNorm / ungated MLP families) take the library GEMMs.
        # The gated-MLP epilogue has a SiLU and a tanh-GELU gate (SwiGLU: Llama / Qwen / Mistral; GeGLU: Gemma).
        # (tensor-parallel shards included: the row-parallel projections then hand their summed slabs to the group's
        # all-reduce ahead of the residual + norm kernel; mixture-of-experts layers run their attention half this
        # way and the routed experts through the grouped kernels)
        fusable = (not self.plain_family and cfg.hidden_act in ("silu", "gelu_tanh", "gelu_pytorch_tanh")
                   and cfg.hidden_size % 64 == 0
                   and (cfg.n_experts > 0 or (cfg.intermediate_size // model.tp_size) % 64 == 0)
                   and cfg.head_dim % 16 == 0 and (self.hq * cfg.head_dim) % 64 == 0)
        if fused_decode is None:
            env = os.environ.get("AREAL_FUSED_DECODE", "1")  # "force": also on the CPU fallbacks of the kernels (tests)
            fused_decode = env == "force" or (self.device.type == "cuda" and env != "0")
        self.FUSED_MIN_BATCH = int(os.environ.get("AREAL_FUSED_MIN_BATCH", self.FUSED_MIN_BATCH))
        self.fused_decode = bool(fused_decode) and fusable
        self._gate_up_choice: Dict[int, str] = {}  # batch bucket -> fastest gate|up kernel (timed once)
        # qkv projection with bias + RoPE + KV-cache write in its epilogue (head sizes that are multiples of 64)
        self.fused_qkv_rope = (cfg.head_dim % 64 == 0 and not cfg.qk_norm
                               and os.environ.get("AREAL_FUSED_QKV_ROPE", "1") != "0")

    # ------------------------------------------------------------------- vision-language requests
    def _placeholder_mask(self, prompt_ids: List[int], grid_thw: torch.Tensor):
        from ..models.vision import mrope_position_ids

        cfg = self.cfg
        img = [t for t in (cfg.image_token_id, cfg.video_token_id) if t is not None]
        return mrope_position_ids(prompt_ids, grid_thw.tolist(), img, int(cfg.vision.get("spatial_merge_size", 2)), True)

    def salted_prompt(self, prompt_ids: List[int], pixel_values: torch.Tensor, grid_thw: torch.Tensor) -> List[int]:
        """Token ids handed to the scheduler for a prompt with images: every image placeholder becomes an
        id >= vocab_size derived from the pixel data and its index, so that the prefix cache can only match
        prompts that show the same images (samples of one group still share their prefill)."""
        cfg = self.cfg
        salt = int(torch.frombuffer(bytearray(pixel_values.float().sum(-1).numpy().tobytes()), dtype=torch.int32)
                   .long().sum()) & 0x3FFFFFFF
        span = (1 << 31) - self.V - 1
        is_img = self._placeholder_mask(prompt_ids, grid_thw)[2].tolist()
        out, j = [], 0
        for t, m in zip(prompt_ids, is_img):
            if m:
                out.append(self.V + (salt + 7919 * j) % span)
                j += 1
            else:
                out.append(int(t))
        return out

    def register_multimodal(self, req_id: int, prompt_ids: List[int], pixel_values: torch.Tensor,
                            grid_thw: torch.Tensor):
        pos3, delta, is_img = self._placeholder_mask(prompt_ids, grid_thw)
        img_index = torch.where(is_img, torch.cumsum(is_img.long(), 0) - 1, torch.full_like(is_img.long(), -1))
        self.mm[int(req_id)] = dict(pos3=pos3, delta=int(delta), img_index=img_index, pixel_values=pixel_values,
                                    grid_thw=grid_thw, embeds=None)

    def release_multimodal(self, req_id: int):
        self.mm.pop(int(req_id), None)

    def _multimodal_prefill_inputs(self, plan, ids: torch.Tensor):
        """-> (hidden [T, H] with image features in place, 3-D rotary positions [3, T]) for a prefill plan.
        Tokens are addressed by their absolute position in the request: prompt positions use the request's
        precomputed tables, positions past the prompt (re-prefill after preemption) are text."""
        dev = self.device
        abs_pos = torch.from_numpy(plan["positions"].astype(np.int64))
        cu = plan["cu_q"].astype(np.int64)
        T = abs_pos.numel()
        pos3 = abs_pos.view(1, T).repeat(3, 1)
        h = self.model.embed_tokens(torch.where(ids >= self.V, torch.full_like(ids, self.cfg.image_token_id), ids))
        for i, rid in enumerate(plan["req_ids"].tolist()):
            m = self.mm.get(int(rid))
            if m is None:
                continue
            lo, hi = int(cu[i]), int(cu[i + 1])
            ap = abs_pos[lo:hi]
            P = m["pos3"].shape[1]
            inside = ap < P
            apc = ap.clamp(max=P - 1)
            pos3[:, lo:hi] = torch.where(inside, m["pos3"][:, apc], ap + m["delta"])
            idx = torch.where(inside, m["img_index"][apc], torch.full_like(ap, -1))
            sel = (idx >= 0).nonzero().flatten()
            if sel.numel():
                if m["embeds"] is None:
                    m["embeds"] = self.model.visual(m["pixel_values"].to(dev), m["grid_thw"].to(dev)).to(h.dtype)
                h[(sel + lo).to(dev)] = m["embeds"][idx[sel].to(dev)]
        return h, pos3.to(dev)

    # ------------------------------------------------------------------------ model pieces
    def _layer(self, blk, h, residual, positions, slots, attn_fn):
        cfg = self.cfg
        if self.plain_family:
            return self._layer_plain(blk, h, residual, attn_fn)
        x, residual = ops.rmsnorm(h, blk.ln1, cfg.rms_norm_eps, residual, cfg.gemma_norm)
        qkv = blk.attn.project_qkv(x)
        attn = attn_fn(blk, qkv)
        h = F.linear(attn.reshape(attn.shape[0], -1), blk.attn.o)
        if self.tp_comm:
            self._tp_all_reduce(h)
        x, residual = ops.rmsnorm(h, blk.ln2, cfg.rms_norm_eps, residual, cfg.gemma_norm)
        if cfg.n_experts > 0:
            h = blk.mlp(x, None, self.model.ep_group)
        else:
            a = ops.train_gemm.gate_up_act(x, blk.mlp.gate_up, blk.mlp.act)  # prefill of many rows: one kernel
            if a is None:
                a = ops.swiglu(F.linear(x, blk.mlp.gate_up), blk.mlp.act)
            h = F.linear(a, blk.mlp.down)
        if self.tp_comm:
            self._tp_all_reduce(h)
        return h, residual

    def _layer_plain(self, blk, h, residual, attn_fn):
        """The layer of the families outside the fused kernels' reach (GPT-2: LayerNorm with bias, biases on the
        output projections, fc -> activation -> proj): the trainer's own norm helper and library GEMMs. Biases of the
        row-parallel projections are added once, after the reduction."""
        from ..models.transformer import add_norm

        cfg = self.cfg
        tp = self.tp_comm
        x, residual = add_norm(cfg, h, blk.ln1, getattr(blk, "ln1_bias", None), residual)
        attn = attn_fn(blk, blk.attn.project_qkv(x))
        h = F.linear(attn.reshape(attn.shape[0], -1), blk.attn.o)
        if tp:
            self._tp_all_reduce(h)
        if cfg.linear_bias:
            h = h + blk.attn.o_bias
        x, residual = add_norm(cfg, h, blk.ln2, getattr(blk, "ln2_bias", None), residual)
        if cfg.n_experts > 0:
            h = blk.mlp(x, None, self.model.ep_group)
        elif cfg.mlp_gated:
            h = F.linear(ops.swiglu(F.linear(x, blk.mlp.gate_up), blk.mlp.act), blk.mlp.down)
        else:
            a = F.linear(x, blk.mlp.fc, blk.mlp.fc_bias)
            a = F.gelu(a, approximate="tanh") if blk.mlp.act == "gelu_tanh" else F.silu(a)
            h = F.linear(a, blk.mlp.proj)
        if tp:
            self._tp_all_reduce(h)
        if not cfg.mlp_gated and cfg.n_experts == 0:
            h = h + blk.mlp.proj_bias
        return h, residual

    def _tp_all_reduce(self, h: torch.Tensor):
        """Sum the partial outputs of a row-parallel projection over the tp group, in place: decode-sized
        messages through the peer-memory kernel (`ipc_ar`, set by the engine when the ranks share a node), the
        rest - prefill activations, CPU runs - through the group's collective."""
        ar = self.ipc_ar
        if ar is not None and ar.fits(h):
            ar.all_reduce_(h)
        elif self._host_staged(h):
            t = h.cpu()
            torch.distributed.all_reduce(t, group=self.model.tp_group)
            h.copy_(t)
        else:
            torch.distributed.all_reduce(h, group=self.model.tp_group)

    def _host_staged(self, t: torch.Tensor) -> bool:
        """A host-backend (gloo) group asked to move device tensors - the one-GPU rehearsal of a server's ranks:
        stage through host memory here. The backend's own device path copies on private streams and worker
        threads, and two ranks sharing one GPU were seen to wait on each other inside it."""
        if not t.is_cuda:
            return False
        if self._gloo_group is None:
            self._gloo_group = torch.distributed.get_backend(self.model.tp_group) == "gloo"
        return self._gloo_group

    def _sync_tokens(self, tok: torch.Tensor, lp: torch.Tensor):
        """Tensor-parallel ranks sample from identical logits with identical Philox keys; broadcasting the
        leader's draw (a few hundred bytes per step) makes agreement unconditional, so a rounding
        difference between ranks can never fork the KV caches. With the peer-memory all-reduce the broadcast is
        left out: that kernel sums every rank's data in rank order on every rank, so the ranks hold bit-identical
        activations by construction, and a decode plan then contains no collective that could wait for a missing
        peer without a time limit."""
        if self.tp_comm and self.ipc_ar is None:
            import torch.distributed as dist

            src = dist.get_global_rank(self.model.tp_group, 0) if self.model.tp_group is not None else 0
            if self._host_staged(tok):
                th, lh = tok.cpu(), lp.cpu()
                dist.broadcast(th, src=src, group=self.model.tp_group)
                dist.broadcast(lh, src=src, group=self.model.tp_group)
                tok.copy_(th)
                lp.copy_(lh)
                return
            dist.broadcast(tok, src=src, group=self.model.tp_group)
            dist.broadcast(lp, src=src, group=self.model.tp_group)

    def _final_logits(self, h, residual) -> torch.Tensor:
        cfg = self.cfg
        if cfg.norm_type != "rms":
            from ..models.transformer import add_norm

            x, _ = add_norm(cfg, h, self.model.norm, getattr(self.model, "norm_bias", None), residual)
        else:
            x, _ = ops.rmsnorm(h, self.model.norm, cfg.rms_norm_eps, residual, cfg.gemma_norm)
        return F.linear(x, self.model.head_weight)

    # ----------------------------------------------------------------------------- prefill
    @torch.no_grad()
    def prefill(self, plan: Dict[str, Any]) -> Tuple[np.ndarray, np.ndarray]:
        dev = self.device
        ids = torch.from_numpy(plan["input_ids"]).to(dev, non_blocking=True).long()
        pos = torch.from_numpy(plan["positions"]).to(dev, non_blocking=True)
        slots = torch.from_numpy(plan["slot_mapping"]).to(dev, non_blocking=True)
        cu_q = torch.from_numpy(plan["cu_q"]).to(dev, non_blocking=True)
        cu_k = torch.from_numpy(plan["cu_k"]).to(dev, non_blocking=True)
        has_prefix = bool(plan["has_prefix"])
        bt = torch.from_numpy(plan["block_tables"]).to(dev, non_blocking=True) if has_prefix else None
        max_q = int(plan["max_q"])
        B = len(plan["req_ids"])
        T = ids.numel()
        hq, hkv, D = self.hq, self.hkv, self.cfg.head_dim
        self.n_prefill_tokens += T

        rope = self.rope
        if self.is_vl and any(int(r) in self.mm for r in plan["req_ids"].tolist()):
            h, pos3 = self._multimodal_prefill_inputs(plan, ids)
            rope = self.model.mrope(pos3)  # per-token table, addressed by the token's index in the plan
            pos = torch.arange(T, dtype=torch.int32, device=dev)
        else:
            h = self.model.embed_tokens(ids, pos)
        if self.cfg.pos_embed != "rope":
            pos = torch.zeros_like(pos)  # the only row of the angle-zero table
        residual = None
        for li, blk in enumerate(self.model.layers):
            kc, vc = self.kv[li]

            def attn_fn(blk_, qkv2d, kc=kc, vc=vc):
                qkv = qkv2d.view(T, hq + 2 * hkv, D)
                ops.rope_and_cache_write_(qkv, pos, slots, rope, kc, vc, hq, hkv)
                q = qkv[:, :hq]
                if has_prefix:  # some row has context in the cache: attention reads the pages directly
                    out = ops.flash_attn_paged_prefill(q, kc, vc, bt, cu_q, cu_k, max_q)
                else:
                    out, _ = ops.flash_attn_varlen(q, qkv[:, hq:hq + hkv], qkv[:, hq + hkv:], cu_q, cu_q, max_q)
                return out

            h, residual = self._layer(blk, h, residual, pos, slots, attn_fn)
        last = (cu_q[1:] - 1).long()
        logits = self._final_logits(h[last], residual[last])
        temp = torch.from_numpy(plan["temperature"]).to(dev)
        top_p = torch.from_numpy(plan["top_p"]).to(dev)
        top_k = torch.from_numpy(np.minimum(plan["top_k"], self.V).astype(np.int32)).to(dev)
        filt = bool((plan["top_k"] < self.V).any() or (plan["top_p"] < 1.0).any())
        pen = {}
        if "penalty_slot" in plan and (plan["penalty_slot"] >= 0).any():  # chunks that do not sample carry slot -1
            pen = dict(counts=self.token_counts(), count_slot=torch.from_numpy(plan["penalty_slot"]).to(dev),
                       freq_pen=torch.from_numpy(plan["frequency_penalty"]).to(dev),
                       pres_pen=torch.from_numpy(plan["presence_penalty"]).to(dev))
        tok, lp = ops.sample(logits, temp, top_k if filt else None, top_p if filt else None, self.seed,
                             self.step_counter, **pen)
        self._sync_tokens(tok, lp)
        tok_h, lp_h = tok.cpu().numpy(), lp.cpu().numpy()
        if self.ipc_ar is not None:
            self.ipc_ar.raise_if_error()
        return tok_h, lp_h

    # ------------------------------------------------------------------ penalty count table
    def token_counts(self) -> "ops.TokenCounts":
        if self.counts is None:
            self.counts = ops.TokenCounts(self.max_batch, self.V, self.device)
        return self.counts

    def clear_penalty_slots(self, slots: List[int]):
        """Zero the count rows of slots that were just handed to new requests."""
        if slots:
            self.token_counts().clear(slots)

    # ------------------------------------------------------------------------------ decode
    def _bucket(self, b: int) -> int:
        for x in self.buckets:
            if x >= b:
                return x
        raise RuntimeError(f"decode batch {b} exceeds the largest bucket {self.buckets[-1]}")

    def smaller_bucket(self, b: int) -> int:
        """Largest batch bucket below the one `b` rows run in (0 when there is none)."""
        cur = self._bucket(b)
        below = [x for x in self.buckets if x < cur]
        return below[-1] if below else 0

    def _alloc_static(self, Bp: int) -> Dict[str, torch.Tensor]:
        dev = self.device
        k = self.decode_steps
        nsplit = ops.decode_num_splits(Bp, self.hkv, 4096, fp8=ops.attention.is_fp8(self.kv[0][0]))
        st = dict(
            input_ids=torch.zeros(Bp, dtype=torch.int64, device=dev),
            positions=torch.zeros(Bp, dtype=torch.int32, device=dev),
            context_lens=torch.ones(Bp, dtype=torch.int32, device=dev),
            slots=torch.zeros(Bp, dtype=torch.int64, device=dev),
            block_tables=torch.zeros(Bp, self.max_pages, dtype=torch.int32, device=dev),
            temperature=torch.ones(Bp, dtype=torch.float32, device=dev),
            top_p=torch.ones(Bp, dtype=torch.float32, device=dev),
            top_k=torch.full((Bp,), self.V, dtype=torch.int32, device=dev),
            rope_delta=torch.zeros(Bp, dtype=torch.int32, device=dev),
            rope_pos=torch.zeros(Bp, dtype=torch.int32, device=dev),
            zero_pos=torch.zeros(Bp, dtype=torch.int32, device=dev),
            out_tokens=torch.zeros(k, Bp, dtype=torch.int64, device=dev),
            out_logprobs=torch.zeros(k, Bp, dtype=torch.float32, device=dev),
            active=torch.zeros(Bp, dtype=torch.int32, device=dev),
            remaining=torch.ones(Bp, dtype=torch.int32, device=dev),
            min_remaining=torch.zeros(Bp, dtype=torch.int32, device=dev),
            stop_ids=torch.full((Bp, self.n_stop_ids), -1, dtype=torch.int32, device=dev),
            kill=torch.zeros(Bp, dtype=torch.int32, device=dev),
            penalty_slot=torch.full((Bp,), -1, dtype=torch.int32, device=dev),
            freq_pen=torch.zeros(Bp, dtype=torch.float32, device=dev),
            pres_pen=torch.zeros(Bp, dtype=torch.float32, device=dev),
        )
        st["nsplit"] = nsplit
        st["ws"] = ops.DecodeWorkspace(Bp, self.hq, self.cfg.head_dim, nsplit, dev) if nsplit > 1 else None
        # Pinned host memory, two sets used alternately: uploads of plan n+1 are written while the copies of
        # plan n may still be queued, and plan n's results are read while plan n+1 fills the other set.
        if dev.type == "cuda":
            st["host"] = [dict(i64=torch.zeros(2, Bp, dtype=torch.int64).pin_memory(),
                               i32=torch.zeros(8, Bp, dtype=torch.int32).pin_memory(),
                               f32=torch.zeros(4, Bp, dtype=torch.float32).pin_memory(),
                               stop=torch.zeros(Bp, self.n_stop_ids, dtype=torch.int32).pin_memory(),
                               bt=torch.zeros(Bp, self.max_pages, dtype=torch.int32).pin_memory(),
                               out_tokens=torch.zeros(k, Bp, dtype=torch.int64).pin_memory(),
                               out_logprobs=torch.zeros(k, Bp, dtype=torch.float32).pin_memory(),
                               ar_err=torch.zeros(1, dtype=torch.int32).pin_memory(),
                               done=torch.cuda.Event()) for _ in range(2)]
        return st

    FUSED_MIN_BATCH = 1  # the hand-written layer wins at every batch size (1.8 against 3.5 ms per step at one row)

    def tune_decode_paths(self):
        """Time the per-bucket choices of the fused decode path now - at engine start-up, while the device is idle -
        instead of at the first decode step of each bucket, when a colocated trainer may be sharing the device."""
        if self.fused_decode and self.device.type == "cuda":
            for b in self.buckets:
                if b >= self.FUSED_MIN_BATCH:
                    self._gate_up_mode(b)

    def _gate_up_mode(self, Bp: int) -> str:
        """Which kernel runs the widest GEMM of the layer (gate|up + SwiGLU) at this batch bucket: "balanced" (one
        column range per CU, ``decode_swiglu.hip``), "tiled" (``decode_gemm.hip`` with the SwiGLU epilogue) or
        "library" (library GEMM followed by the activation kernel). Depends on the model's shapes (the hand-written
        kernels win where the GEMM waits for its loads: Qwen2-1.5B at every bucket; compute-bound shapes such as
        Qwen2.5-7B from ~128 rows on go to the library), so the candidates are timed once per bucket on the model's
        own weights, rotating over the layers so that no arm streams from a warm cache. The other three GEMMs stay
        fused either way."""
        if Bp in self._gate_up_choice:
            return self._gate_up_choice[Bp]
        if self.cfg.n_experts > 0:  # routed experts: the grouped kernels, nothing to choose
            self._gate_up_choice[Bp] = "experts"
            return "experts"
        from ..ops import decode_gemm as dg

        has_plan = dg.swiglu_plan(Bp, self.cfg.intermediate_size // max(1, self.model.tp_size)) is not None
        forced = os.environ.get("AREAL_FUSED_GATE_UP")
        # tensor-parallel ranks must run the SAME kernels (their roundings differ, and the ranks' activations have to
        # stay bit-identical): no per-rank timing there
        if (forced is not None or self.device.type != "cuda" or torch.cuda.is_current_stream_capturing()
                or self.model.tp_size > 1):
            mode = {"0": "library", "1": "tiled"}.get(forced or "", "balanced" if has_plan else "tiled")
            if mode == "balanced" and not has_plan:
                mode = "tiled"
            self._gate_up_choice[Bp] = mode
            return mode
        layers = self.model.layers
        x = torch.randn(Bp, self.cfg.hidden_size, device=self.device, dtype=self.dtype)
        act = "silu" if self.cfg.hidden_act == "silu" else "gelu_tanh"
        epi = dg.gated_epilogue(act)
        arms = {"tiled": lambda i: dg.dgemm(x, layers[i % len(layers)].mlp.gate_up, epilogue=epi),
                "library": lambda i: ops.swiglu(F.linear(x, layers[i % len(layers)].mlp.gate_up),
                                                layers[i % len(layers)].mlp.act)}
        if has_plan:
            arms["balanced"] = lambda i: dg.dgemm_swiglu_balanced(x, layers[i % len(layers)].mlp.gate_up, act=act)
        times = {}
        for name, fn in arms.items():
            for i in range(2):
                fn(i)
            torch.cuda.current_stream().synchronize()
            best = float("inf")
            n = max(8, len(layers))
            for _ in range(3):  # the smallest of three: other work on the device (a colocated trainer) only adds time
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(n):
                    fn(i)
                e1.record()
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1) / n)
            times[name] = best
        mode = min(times, key=times.get)
        self._gate_up_choice[Bp] = mode
        logger.info(f"decode batch bucket {Bp}: gate|up GEMM " + ", ".join(f"{k} {v * 1e3:.1f} us" for k, v in times.items())
                    + f" -> {mode}")
        return mode

    def _decode_layers_fused(self, st: Dict[str, torch.Tensor], h: torch.Tensor, rope_pos: torch.Tensor):
        """All decoder layers of one decode step on the fused path -> final-normed hidden states [Bp, H].

        Per layer: qkv GEMM with bias + RoPE + KV-cache write in its epilogue (smaller batches: split-K slabs ->
        slab sum + bias + RoPE + KV-cache write) -> paged attention ->
        o GEMM (slabs) -> slab sum + residual + RMSNorm -> gate|up GEMM with SwiGLU epilogue -> down GEMM
        (slabs) -> slab sum + residual + the NEXT layer's RMSNorm. Mixture-of-experts layers keep the attention
        half and run router and experts through the grouped kernels, then residual + the next norm."""
        from ..ops import decode_gemm as dg

        cfg = self.cfg
        hq, hkv, D = self.hq, self.hkv, cfg.head_dim
        Bp, H = h.shape
        eps, gemma = cfg.rms_norm_eps, cfg.gemma_norm
        layers = self.model.layers
        gate_up_mode = self._gate_up_mode(Bp)
        gate_act = "silu" if cfg.hidden_act == "silu" else "gelu_tanh"  # SwiGLU, or GeGLU (Gemma)
        tp = self.tp_comm

        def finish(slab, residual, weight):
            """Split-K slabs of a projection -> (normalised input of the next GEMM, new residual stream). A
            row-parallel projection's slabs hold this rank's partial sums: they are summed to bf16, all-reduced over
            the tensor-parallel group, and only then added to the residual and normalised."""
            if not tp:
                return dg.splitk_residual_rmsnorm(slab, slab.shape[0], Bp, H, residual=residual, weight=weight,
                                                  eps=eps, gemma=gemma, dtype=self.dtype)
            part, _ = dg.splitk_residual_rmsnorm(slab, slab.shape[0], Bp, H, dtype=self.dtype)
            self._tp_all_reduce(part)
            return ops.rmsnorm(part, weight, eps, residual, gemma)

        x, residual = ops.rmsnorm(h, layers[0].ln1, eps, None, gemma)
        for li, blk in enumerate(layers):
            kc, vc = self.kv[li]
            if self.fused_qkv_rope and Bp >= dg.QKV_ROPE_MIN_ROWS:
                q = dg.dgemm_qkv_rope(x, blk.attn.qkv, blk.attn.qkv_bias, rope_pos, st["slots"], self.rope, kc, vc,
                                      hq, hkv, D)
            else:
                slab = dg.dgemm(x, blk.attn.qkv, epilogue=dg.EPI_SLAB)
                q = dg.splitk_rope_cache_write(slab, slab.shape[0], blk.attn.qkv_bias, rope_pos, st["slots"],
                                               self.rope, kc, vc, hq, hkv, D, dtype=self.dtype,
                                               q_norm=blk.attn.q_norm if cfg.qk_norm else None,
                                               k_norm=blk.attn.k_norm if cfg.qk_norm else None, norm_eps=eps)
            attn = ops.paged_attention_decode(q, kc, vc, st["block_tables"], st["context_lens"], nsplit=st["nsplit"],
                                              workspace=st["ws"])
            slab = dg.dgemm(attn.view(Bp, hq * D), blk.attn.o, epilogue=dg.EPI_SLAB)
            x, residual = finish(slab, residual, blk.ln2)
            nw = layers[li + 1].ln1 if li + 1 < len(layers) else self.model.norm
            if gate_up_mode == "experts":
                # router + permutation + grouped GEMMs + weighted combine; the sum joins the residual stream in the
                # next norm's kernel
                mixed = blk.mlp(x, None, self.model.ep_group)
                if tp:
                    self._tp_all_reduce(mixed)
                x, residual = ops.rmsnorm(mixed, nw, eps, residual, gemma)
                continue
            if gate_up_mode == "balanced":
                act = dg.dgemm_swiglu_balanced(x, blk.mlp.gate_up, act=gate_act)
            elif gate_up_mode == "tiled":
                act = dg.dgemm(x, blk.mlp.gate_up, epilogue=dg.gated_epilogue(gate_act))
            else:
                act = ops.swiglu(F.linear(x, blk.mlp.gate_up), blk.mlp.act)
            slab = dg.dgemm(act, blk.mlp.down, epilogue=dg.EPI_SLAB)
            x, residual = finish(slab, residual, nw)
        return x

    LM_HEAD_OWN_MAX_ROWS = 96

    def _lm_head(self, x: torch.Tensor) -> torch.Tensor:
        """Final-normed hidden states [Bp, H] -> logits [Bp, V]. Up to ~100 rows the projection is a pure weight
        stream (467 MB for Qwen2's vocabulary) and the decode GEMM on 64-row x 128-column tiles reads it at
        5.4-6.1 TB/s (77 / 86 / 99 us at 8 / 32 / 64 rows against 163 / 106 / 114 us for the library); from 128 rows
        on the two are level and the library stays."""
        w = self.model.head_weight
        if (x.is_cuda and x.shape[0] <= self.LM_HEAD_OWN_MAX_ROWS and x.dtype == torch.bfloat16
                and w.dtype == torch.bfloat16 and w.is_contiguous() and w.shape[0] % 32 == 0 and w.shape[1] % 64 == 0
                and x.stride(1) == 1):
            from ..ops import decode_gemm as dg

            return dg.dgemm(x, w, epilogue=dg.EPI_BF16, cfg=4, stages=3, ksplit=1, loader_waves=4, nt=True)
        return F.linear(x, w)

    def _decode_body(self, st: Dict[str, torch.Tensor], filtered: bool, penalized: bool = False):
        """`decode_steps` decode iterations on the static buffers (captured into a graph)."""
        hq, hkv, D = self.hq, self.hkv, self.cfg.head_dim
        Bp = st["input_ids"].numel()
        qkv_stride = (hq + 2 * hkv) * D
        fused = self.fused_decode and Bp >= self.FUSED_MIN_BATCH
        pen = {}
        if penalized:
            pen = dict(counts=self.token_counts(), count_slot=st["penalty_slot"], freq_pen=st["freq_pen"],
                       pres_pen=st["pres_pen"])
        # rows the host has given up on retire; live rows take their KV slot from the (refreshed) page table
        ops.decode_begin_(st["kill"], st["active"], st["positions"], st["context_lens"], st["slots"],
                          st["block_tables"])
        for s in range(self.decode_steps):
            h = self.model.embed_tokens(st["input_ids"], st["positions"])
            residual = None
            rope_pos = st["positions"] if self.cfg.pos_embed == "rope" else st["zero_pos"]
            if self.is_vl:  # generated tokens of a request with images sit `delta` away from their index
                torch.add(st["positions"], st["rope_delta"], out=st["rope_pos"])
                rope_pos = st["rope_pos"]
            if fused:
                logits = self._lm_head(self._decode_layers_fused(st, h, rope_pos))
            else:
                for li, blk in enumerate(self.model.layers):
                    kc, vc = self.kv[li]

                    def attn_fn(blk_, qkv2d, kc=kc, vc=vc, rope_pos=rope_pos):
                        qkv = qkv2d.view(Bp, hq + 2 * hkv, D)
                        ops.rope_and_cache_write_(qkv, rope_pos, st["slots"], self.rope, kc, vc, hq, hkv)
                        return ops.paged_attention_decode(qkv[:, :hq], kc, vc, st["block_tables"], st["context_lens"],
                                                          nsplit=st["nsplit"], workspace=st["ws"],
                                                          q_row_stride=qkv_stride)

                    h, residual = self._layer(blk, h, residual, st["positions"], st["slots"], attn_fn)
                logits = self._final_logits(h, residual)
            ops.sample(logits, st["temperature"], st["top_k"] if filtered else None, st["top_p"] if filtered else None,
                       self.seed, self.step_counter, bump_step=True, out_tokens=st["out_tokens"][s],
                       out_logprobs=st["out_logprobs"][s], active=st["active"], **pen)
            self._sync_tokens(st["out_tokens"][s], st["out_logprobs"][s])
            ops.decode_advance_(st["out_tokens"][s], st["input_ids"], st["positions"], st["context_lens"],
                                st["slots"], st["block_tables"], st["active"], st["remaining"], st["min_remaining"],
                                st["stop_ids"])

    _STATE_KEYS = ("input_ids", "positions", "context_lens", "slots", "active", "remaining", "min_remaining")

    def _get_graph(self, Bp: int, filtered: bool, penalized: bool = False):
        key = (Bp, filtered, penalized)
        if key in self.graphs:
            return self.graphs[key], self.static[Bp]
        if Bp not in self.static:
            self.static[Bp] = self._alloc_static(Bp)
        st = self.static[Bp]
        if not self.use_graph:
            self.graphs[key] = None
            return None, st
        logger.info(f"capturing decode graph: batch bucket {Bp}, filtered={filtered}, penalized={penalized}")
        # warm up eagerly on a side stream (allocator + hipBLASLt workspaces), then capture. The warm-up runs on
        # whatever the static buffers hold: rows are inactive (no KV writes) and no penalty slot is touched.
        saved = {k: st[k].clone() for k in self._STATE_KEYS + ("penalty_slot",)}
        st["active"].zero_()
        st["penalty_slot"].fill_(-1)
        step_saved = self.step_counter.clone()
        # (one side stream per runner, and waits on that stream only: a device-wide synchronisation would also wait
        # for whatever other threads of the process - a colocated trainer - have queued)
        if getattr(self, "_warm_stream", None) is None:
            self._warm_stream = torch.cuda.Stream(device=self.device)
        s = self._warm_stream
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._decode_body(st, filtered, penalized)
        torch.cuda.current_stream().wait_stream(s)
        s.synchronize()
        torch.cuda.current_stream().synchronize()
        for k, v in saved.items():
            st[k].copy_(v)
        g = torch.cuda.CUDAGraph()
        if self._graph_pool is None:
            self._graph_pool = torch.cuda.graph_pool_handle()
        st["active"].zero_()
        st["penalty_slot"].fill_(-1)
        with torch.cuda.graph(g, pool=self._graph_pool, capture_error_mode="thread_local"):
            self._decode_body(st, filtered, penalized)
        for k, v in saved.items():
            st[k].copy_(v)
        self.step_counter.copy_(step_saved)
        self.graphs[key] = g
        logger.info(f"captured decode graph: batch bucket {Bp}, {self.decode_steps} step(s), "
                    f"kv splits {st['nsplit']}, filtered={filtered}, penalized={penalized}")
        return g, st

    def _with_defaults(self, plan: Dict[str, Any], B: int) -> Dict[str, Any]:
        """Hand-made plans (probes, tests) may leave out what only the scheduler knows: no stop ids, no
        penalties, a budget that outlasts the plan."""
        if "kill" in plan and "remaining" in plan and "penalty_slot" in plan:
            return plan
        plan = dict(plan)
        plan.setdefault("kill", np.zeros(B, np.int32))
        plan.setdefault("remaining", np.full(B, 1 << 30, np.int32))
        plan.setdefault("min_remaining", np.zeros(B, np.int32))
        plan.setdefault("stop_ids", np.full((B, self.n_stop_ids), -1, np.int32))
        plan.setdefault("penalty_slot", np.full(B, -1, np.int32))
        plan.setdefault("frequency_penalty", np.zeros(B, np.float32))
        plan.setdefault("presence_penalty", np.zeros(B, np.float32))
        return plan

    @torch.no_grad()
    def decode_launch(self, plan: Dict[str, Any]) -> Dict[str, Any]:
        """Issue a decode plan: uploads, graph replay, copy of the sampled tokens / log-probs into pinned memory,
        completion event. Returns a handle for `decode_wait`. Nothing here waits for the device.

        A continuation plan (`plan["continuation"]`) runs on the rows and the device state its predecessor left
        behind (input tokens, positions, slots, activity, budgets are advanced on the device): only the page
        tables and the kill mask are uploaded."""
        B = len(plan["req_ids"])
        assert int(plan["n_steps"]) == self.decode_steps
        plan = self._with_defaults(plan, B)
        cont = bool(plan.get("continuation", False))
        if cont:
            Bp, filtered, penalized = self._last_decode
            assert B <= Bp, "a continuation keeps the rows of the plan it follows"
        else:
            Bp = self._bucket(B)
            filtered = bool((plan["top_k"] < self.V).any() or (plan["top_p"] < 1.0).any())
            penalized = bool((plan["penalty_slot"] >= 0).any())
            self._last_decode = (Bp, filtered, penalized)
        g, st = self._get_graph(Bp, filtered, penalized)
        mp = int(plan["max_pages"])
        assert mp <= self.max_pages, "request context exceeds the engine's context length"
        self._launches += 1
        cuda = self.device.type == "cuda"
        host = st["host"][self._launches % 2] if cuda else None
        if cuda:
            hbt, hi32 = host["bt"], host["i32"]
            hbt.zero_()
            hbt[:B, :mp] = torch.from_numpy(plan["block_tables"])
            st["block_tables"].copy_(hbt, non_blocking=True)
            hi32[4].fill_(1)  # kill: padded rows stay dead
            hi32[4, :B] = torch.from_numpy(plan["kill"])
            st["kill"].copy_(hi32[4], non_blocking=True)
        else:
            st["block_tables"].zero_()
            st["block_tables"][:B, :mp] = torch.from_numpy(plan["block_tables"])
            st["kill"].fill_(1)
            st["kill"][:B] = torch.from_numpy(plan["kill"])
        if not cont:
            top_k = np.minimum(plan["top_k"], self.V).astype(np.int32)
            delta = None
            if self.is_vl:
                delta = np.asarray([self.mm.get(int(r), {}).get("delta", 0) for r in plan["req_ids"].tolist()],
                                   dtype=np.int32)
            if cuda:
                hi64, hf32, hstop = host["i64"], host["f32"], host["stop"]
                hi64.zero_()
                hi32[:4].zero_(); hi32[5:].zero_()
                hi32[1].fill_(1); hi32[2].fill_(self.V); hi32[5].fill_(1); hi32[7].fill_(-1)
                hf32[0].fill_(1.0); hf32[1].fill_(1.0); hf32[2].zero_(); hf32[3].zero_()
                hstop.fill_(-1)
                hi64[0, :B] = torch.from_numpy(plan["input_ids"].astype(np.int64))
                hi64[1, :B] = torch.from_numpy(plan["slot_mapping"])
                hi32[0, :B] = torch.from_numpy(plan["positions"])
                hi32[1, :B] = torch.from_numpy(plan["context_lens"])
                hi32[2, :B] = torch.from_numpy(top_k)
                if delta is not None:
                    hi32[3, :B] = torch.from_numpy(delta)
                    st["rope_delta"].copy_(hi32[3], non_blocking=True)
                hi32[5, :B] = torch.from_numpy(plan["remaining"])
                hi32[6, :B] = torch.from_numpy(plan["min_remaining"])
                hi32[7, :B] = torch.from_numpy(plan["penalty_slot"])
                hf32[0, :B] = torch.from_numpy(plan["temperature"])
                hf32[1, :B] = torch.from_numpy(plan["top_p"])
                hf32[2, :B] = torch.from_numpy(plan["frequency_penalty"])
                hf32[3, :B] = torch.from_numpy(plan["presence_penalty"])
                hstop[:B] = torch.from_numpy(plan["stop_ids"])
                for name, src in (("input_ids", hi64[0]), ("slots", hi64[1]), ("positions", hi32[0]),
                                  ("context_lens", hi32[1]), ("top_k", hi32[2]), ("remaining", hi32[5]),
                                  ("min_remaining", hi32[6]), ("penalty_slot", hi32[7]), ("temperature", hf32[0]),
                                  ("top_p", hf32[1]), ("freq_pen", hf32[2]), ("pres_pen", hf32[3]), ("stop_ids", hstop)):
                    st[name].copy_(src, non_blocking=True)
                st["active"].fill_(1)  # decode_begin retires the padded rows through `kill`
            else:
                for name, src, fill in (("input_ids", plan["input_ids"].astype(np.int64), 0),
                                        ("slots", plan["slot_mapping"], 0), ("positions", plan["positions"], 0),
                                        ("context_lens", plan["context_lens"], 1), ("top_k", top_k, self.V),
                                        ("remaining", plan["remaining"], 1), ("min_remaining", plan["min_remaining"], 0),
                                        ("penalty_slot", plan["penalty_slot"], -1),
                                        ("temperature", plan["temperature"], 1.0), ("top_p", plan["top_p"], 1.0),
                                        ("freq_pen", plan["frequency_penalty"], 0.0),
                                        ("pres_pen", plan["presence_penalty"], 0.0)):
                    st[name].fill_(fill)
                    st[name][:B] = torch.from_numpy(src)
                st["stop_ids"].fill_(-1)
                st["stop_ids"][:B] = torch.from_numpy(plan["stop_ids"])
                st["active"].fill_(1)
                if self.is_vl:
                    st["rope_delta"].zero_()
                    st["rope_delta"][:B] = torch.from_numpy(delta)
        if g is not None:
            g.replay()
        else:
            self._decode_body(st, filtered, penalized)
        self.n_decode_tokens += B * self.decode_steps
        if cuda:
            host["out_tokens"].copy_(st["out_tokens"], non_blocking=True)
            host["out_logprobs"].copy_(st["out_logprobs"], non_blocking=True)
            if self.ipc_ar is not None:  # did every peer-memory all-reduce of this plan see its peers?
                self.ipc_ar.read_error_async(host["ar_err"])
            host["done"].record()
            return dict(host=host)
        return dict(tokens=st["out_tokens"].numpy().copy(), logprobs=st["out_logprobs"].numpy().copy())

    @staticmethod
    def decode_wait(handle: Dict[str, Any]) -> Tuple[np.ndarray, np.ndarray]:
        """Sampled tokens / log-probs [n_steps, Bp] of a launched plan (blocks until its event has fired). The
        arrays alias pinned memory that the launch after next overwrites: consume them before launching again."""
        host = handle.get("host")
        if host is None:
            return handle["tokens"], handle["logprobs"]
        host["done"].synchronize()
        if int(host["ar_err"][0]) != 0:
            from ..ops.ipc_allreduce import AllReduceTimeout

            raise AllReduceTimeout("a tensor-parallel peer did not arrive in a decode plan's all-reduce: its tokens "
                                   "are void")
        return host["out_tokens"].numpy(), host["out_logprobs"].numpy()

    def decode(self, plan: Dict[str, Any]) -> Tuple[np.ndarray, np.ndarray]:
        return self.decode_wait(self.decode_launch(plan))

    def warmup(self, batch_sizes: Optional[List[int]] = None):
        """Capture decode graphs ahead of time for the given batch sizes."""
        for b in batch_sizes or []:
            self._get_graph(self._bucket(b), False, False)

    def reset_graphs(self):
        self.graphs.clear()

    def kv_cache_bytes(self) -> int:
        return sum(k.numel() * k.element_size() + v.numel() * v.element_size() for k, v in self.kv)
