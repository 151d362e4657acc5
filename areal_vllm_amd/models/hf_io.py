"""HuggingFace checkpoint interop for the native transformer (safetensors + config.json).

Load: every rank opens the safetensors shards with ``safe_open`` (mmap) and reads only the slices it
owns (its tensor-parallel head / FFN slice, its pipeline stage's layers) - no rank-0 broadcast of the
full state dict as in the reference (``areal/utils/fsdp.py:288-328``). q/k/v and gate/up are fused on
load and split again on save, so checkpoints stay in the stock HF layout
(``areal/experimental/model/hf_load.py``, ``hf_save.py`` and
``realhf/impl/model/conversion/hf_registry.py:73-382`` are the reference counterparts).

Save: tensor-parallel shards are gathered to tp rank 0 of every pipeline stage, files are split at
``max_shard_size`` bytes and indexed by ``model.safetensors.index.json``.
"""
import json
import os
import re
from typing import Callable, Dict, Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist

from .config import ModelConfig

MAX_SHARD_BYTES = int(os.environ.get("AREAL_SAVE_MAX_SHARD_SIZE_BYTE", 3 * 1024**3))


# ------------------------------------------------------------------------------- name mapping
def _layer_prefix(i: int) -> str:
    return f"model.layers.{i}."


def hf_names_for_layer(cfg: ModelConfig, i: int) -> Dict[str, List[str]]:
    """native sub-name -> list of HF tensor names that are concatenated (dim 0) to form it."""
    p = _layer_prefix(i)
    a = p + "self_attn."
    m = {
        "ln1": [p + "input_layernorm.weight"],
        "ln2": [p + "post_attention_layernorm.weight"],
        "attn.qkv": [a + "q_proj.weight", a + "k_proj.weight", a + "v_proj.weight"],
        "attn.o": [a + "o_proj.weight"],
    }
    if cfg.qkv_bias:
        m["attn.qkv_bias"] = [a + "q_proj.bias", a + "k_proj.bias", a + "v_proj.bias"]
    if cfg.qk_norm:
        m["attn.q_norm"] = [a + "q_norm.weight"]
        m["attn.k_norm"] = [a + "k_norm.weight"]
    if cfg.n_experts > 0:
        if cfg.family == "mixtral":
            m["mlp.router"] = [p + "block_sparse_moe.gate.weight"]
        else:
            m["mlp.router"] = [p + "mlp.gate.weight"]
    else:
        m["mlp.gate_up"] = [p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight"]
        m["mlp.down"] = [p + "mlp.down_proj.weight"]
    return m


def hf_expert_names(cfg: ModelConfig, i: int, e: int) -> Tuple[str, str, str]:
    p = _layer_prefix(i)
    if cfg.family == "mixtral":
        b = p + f"block_sparse_moe.experts.{e}."
        return b + "w1.weight", b + "w3.weight", b + "w2.weight"
    b = p + f"mlp.experts.{e}."
    return b + "gate_proj.weight", b + "up_proj.weight", b + "down_proj.weight"


class _ShardReader:
    """Lazy reader over one or many safetensors files of a HF checkpoint directory."""

    def __init__(self, path: str):
        from safetensors import safe_open

        self.path = path
        idx = os.path.join(path, "model.safetensors.index.json")
        self.files: Dict[str, str] = {}
        if os.path.exists(idx):
            with open(idx) as f:
                self.files = json.load(f)["weight_map"]
        else:
            cands = sorted(f for f in os.listdir(path) if f.endswith(".safetensors"))
            if not cands:
                raise FileNotFoundError(f"no safetensors files under {path}")
            for fn in cands:
                with safe_open(os.path.join(path, fn), framework="pt") as f:
                    for k in f.keys():
                        self.files[k] = fn
        self._open = {}
        self._safe_open = safe_open

    def has(self, name: str) -> bool:
        return name in self.files

    def _handle(self, name: str):
        fn = self.files[name]
        if fn not in self._open:
            self._open[fn] = self._safe_open(os.path.join(self.path, fn), framework="pt")
        return self._open[fn]

    def get(self, name: str) -> torch.Tensor:
        return self._handle(name).get_tensor(name)

    def get_rows(self, name: str, lo: int, hi: int) -> torch.Tensor:
        return self._handle(name).get_slice(name)[lo:hi]

    def get_cols(self, name: str, lo: int, hi: int) -> torch.Tensor:
        return self._handle(name).get_slice(name)[:, lo:hi]


def _tp_range(n: int, tp_size: int, tp_rank: int) -> Tuple[int, int]:
    assert n % tp_size == 0
    s = n // tp_size
    return tp_rank * s, (tp_rank + 1) * s


def _vocab_rows(model, rd: "_ShardReader", name: str) -> torch.Tensor:
    """A [vocab, H] checkpoint tensor, or this rank's rows of it when the model shards the vocabulary."""
    if getattr(model, "vocab_parallel", False):
        lo, hi = model.vocab_range
        return rd.get_rows(name, lo, hi)
    return rd.get(name)


def _gather_vocab(model, t: torch.Tensor, group) -> torch.Tensor:
    """Inverse of `_vocab_rows`: the full [vocab, H] tensor from the tp ranks' row shards (the last shard
    may be shorter, so shards travel padded to the longest)."""
    if not getattr(model, "vocab_parallel", False) or group is None or dist.get_world_size(group) == 1:
        return t
    from ..ops import vocab_shard_range

    n = dist.get_world_size(group)
    rows = [hi - lo for lo, hi in (vocab_shard_range(model.cfg.vocab_size, n, i) for i in range(n))]
    pad = max(rows) - t.shape[0]
    padded = torch.cat([t, t.new_zeros(pad, t.shape[1])], 0) if pad else t.contiguous()
    parts = [torch.empty_like(padded) for _ in range(n)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:r] for p, r in zip(parts, rows)], 0)


def load_hf_weights(model, path: str, strict: bool = True) -> None:
    """Fill `model` (a ``Transformer``, possibly a TP shard / pipeline stage) from a HF checkpoint."""
    cfg: ModelConfig = model.cfg
    rd = _ShardReader(path)
    tp, tr = model.tp_size, model.tp_rank
    D = cfg.head_dim
    params = dict(model.named_parameters())

    def put(name: str, t: torch.Tensor):
        if name not in params:
            if strict:
                raise KeyError(f"model has no parameter {name}")
            return
        p = params[name]
        assert tuple(p.shape) == tuple(t.shape), f"{name}: {tuple(p.shape)} vs checkpoint {tuple(t.shape)}"
        p.data.copy_(t.to(p.dtype))

    if cfg.family == "gpt2":
        return _load_gpt2(model, rd, put)
    if model.has_embed:
        put("embed", _vocab_rows(model, rd, "model.embed_tokens.weight"))
        for n in params:  # vision tower: parameter names are the checkpoint names
            if n.startswith("visual."):
                put(n, rd.get(n))
    if model.has_head:
        put("norm", rd.get("model.norm.weight"))
        if "head" in params:
            if cfg.is_critic:
                for cand in ("score.weight", "v_head.weight", "value_head.weight"):
                    if rd.has(cand):
                        put("head", rd.get(cand))
                        break
                else:
                    if strict:
                        raise KeyError("critic checkpoint has no score/v_head weight")
            elif rd.has("lm_head.weight"):
                put("head", _vocab_rows(model, rd, "lm_head.weight"))
            else:  # tied checkpoint loaded into an untied stage (last pipeline stage)
                put("head", _vocab_rows(model, rd, "model.embed_tokens.weight"))
    hq_lo, hq_hi = _tp_range(cfg.n_q_heads, tp, tr)
    if cfg.n_kv_heads >= tp:
        hkv_lo, hkv_hi = _tp_range(cfg.n_kv_heads, tp, tr)
    else:  # kv heads replicated across tp ranks
        rep = tp // cfg.n_kv_heads
        hkv_lo = tr // rep
        hkv_hi = hkv_lo + 1
    for li, i in enumerate(model.layer_ids):
        names = hf_names_for_layer(cfg, i)
        pre = f"layers.{li}."
        put(pre + "ln1", rd.get(names["ln1"][0]))
        put(pre + "ln2", rd.get(names["ln2"][0]))
        qn, kn, vn = names["attn.qkv"]
        put(pre + "attn.qkv", torch.cat([rd.get_rows(qn, hq_lo * D, hq_hi * D), rd.get_rows(kn, hkv_lo * D, hkv_hi * D),
                                         rd.get_rows(vn, hkv_lo * D, hkv_hi * D)], 0))
        if cfg.qkv_bias:
            qb, kb, vb = names["attn.qkv_bias"]
            put(pre + "attn.qkv_bias", torch.cat([rd.get(qb)[hq_lo * D:hq_hi * D], rd.get(kb)[hkv_lo * D:hkv_hi * D],
                                                  rd.get(vb)[hkv_lo * D:hkv_hi * D]], 0))
        put(pre + "attn.o", rd.get_cols(names["attn.o"][0], hq_lo * D, hq_hi * D))
        if cfg.qk_norm:
            put(pre + "attn.q_norm", rd.get(names["attn.q_norm"][0]))
            put(pre + "attn.k_norm", rd.get(names["attn.k_norm"][0]))
        if cfg.n_experts > 0:
            put(pre + "mlp.router", rd.get(names["mlp.router"][0]))
            blk = model.layers[li].mlp
            f_lo, f_hi = _tp_range(cfg.moe_intermediate_size, model.etp_size, model.etp_rank)
            gus, downs = [], []
            for e in range(blk.ep_rank * blk.n_local, (blk.ep_rank + 1) * blk.n_local):
                g, u, d = hf_expert_names(cfg, i, e)
                gus.append(torch.cat([rd.get_rows(g, f_lo, f_hi), rd.get_rows(u, f_lo, f_hi)], 0))
                downs.append(rd.get_cols(d, f_lo, f_hi))
            put(pre + "mlp.gate_up", torch.stack(gus))
            put(pre + "mlp.down", torch.stack(downs))
        else:
            f_lo, f_hi = _tp_range(cfg.intermediate_size, tp, tr)
            gn, un = names["mlp.gate_up"]
            put(pre + "mlp.gate_up", torch.cat([rd.get_rows(gn, f_lo, f_hi), rd.get_rows(un, f_lo, f_hi)], 0))
            put(pre + "mlp.down", rd.get_cols(names["mlp.down"][0], f_lo, f_hi))


# ------------------------------------------------------------------------------------- GPT-2
# GPT-2 checkpoints store every linear as Conv1D, i.e. weight [in, out] (the transpose of nn.Linear), q|k|v
# fused in `c_attn` along the output axis, and (for the original hub files) names without the
# `transformer.` prefix. Reference converter: ``realhf/api/from_hf/gpt2.py``.
def _load_gpt2(model, rd: "_ShardReader", put) -> None:
    cfg: ModelConfig = model.cfg
    tp, tr = model.tp_size, model.tp_rank
    D, H = cfg.head_dim, cfg.hidden_size
    pre_hf = "transformer." if rd.has("transformer.wte.weight") or rd.has("transformer.h.0.ln_1.weight") else ""
    if model.has_embed:
        put("embed", _vocab_rows(model, rd, pre_hf + "wte.weight"))
        put("pos_embed", rd.get(pre_hf + "wpe.weight"))
    if model.has_head:
        put("norm", rd.get(pre_hf + "ln_f.weight"))
        put("norm_bias", rd.get(pre_hf + "ln_f.bias"))
        if "head" in dict(model.named_parameters()):
            if cfg.is_critic:
                put("head", rd.get("score.weight"))
            else:
                put("head", _vocab_rows(model, rd, "lm_head.weight" if rd.has("lm_head.weight") else pre_hf + "wte.weight"))
    q_lo, q_hi = (x * D for x in _tp_range(cfg.n_q_heads, tp, tr))
    f_lo, f_hi = _tp_range(cfg.intermediate_size, tp, tr)
    for li, i in enumerate(model.layer_ids):
        b, pre = f"{pre_hf}h.{i}.", f"layers.{li}."
        for ours, theirs in (("ln1", "ln_1.weight"), ("ln1_bias", "ln_1.bias"), ("ln2", "ln_2.weight"),
                             ("ln2_bias", "ln_2.bias"), ("attn.o_bias", "attn.c_proj.bias"),
                             ("mlp.proj_bias", "mlp.c_proj.bias")):
            put(pre + ours, rd.get(b + theirs))
        w, bias = rd.get(b + "attn.c_attn.weight"), rd.get(b + "attn.c_attn.bias")
        put(pre + "attn.qkv", torch.cat([w[:, o + q_lo:o + q_hi] for o in (0, H, 2 * H)], 1).t().contiguous())
        put(pre + "attn.qkv_bias", torch.cat([bias[o + q_lo:o + q_hi] for o in (0, H, 2 * H)], 0))
        put(pre + "attn.o", rd.get_rows(b + "attn.c_proj.weight", q_lo, q_hi).t().contiguous())
        put(pre + "mlp.fc", rd.get_cols(b + "mlp.c_fc.weight", f_lo, f_hi).t().contiguous())
        put(pre + "mlp.fc_bias", rd.get(b + "mlp.c_fc.bias")[f_lo:f_hi])
        put(pre + "mlp.proj", rd.get_rows(b + "mlp.c_proj.weight", f_lo, f_hi).t().contiguous())


def _gpt2_state(model, g) -> Iterable[Tuple[str, torch.Tensor]]:
    cfg: ModelConfig = model.cfg
    params = dict(model.named_parameters())
    if model.has_embed:
        yield "transformer.wte.weight", _gather_vocab(model, params["embed"].data, g)
        yield "transformer.wpe.weight", params["pos_embed"].data
    for li, blk in enumerate(model.layers):
        b, pre = f"transformer.h.{model.layer_ids[li]}.", f"layers.{li}."
        for ours, theirs in (("ln1", "ln_1.weight"), ("ln1_bias", "ln_1.bias"), ("ln2", "ln_2.weight"),
                             ("ln2_bias", "ln_2.bias"), ("attn.o_bias", "attn.c_proj.bias"),
                             ("mlp.proj_bias", "mlp.c_proj.bias")):
            yield b + theirs, params[pre + ours].data
        n = blk.attn.hq * cfg.head_dim
        qkv, qb = params[pre + "attn.qkv"].data, params[pre + "attn.qkv_bias"].data
        yield b + "attn.c_attn.weight", torch.cat([_gather_tp(qkv[o:o + n], 0, g) for o in (0, n, 2 * n)], 0).t().contiguous()
        yield b + "attn.c_attn.bias", torch.cat([_gather_tp(qb[o:o + n], 0, g) for o in (0, n, 2 * n)], 0)
        yield b + "attn.c_proj.weight", _gather_tp(params[pre + "attn.o"].data, 1, g).t().contiguous()
        yield b + "mlp.c_fc.weight", _gather_tp(params[pre + "mlp.fc"].data, 0, g).t().contiguous()
        yield b + "mlp.c_fc.bias", _gather_tp(params[pre + "mlp.fc_bias"].data, 0, g)
        yield b + "mlp.c_proj.weight", _gather_tp(params[pre + "mlp.proj"].data, 1, g).t().contiguous()
    if model.has_head:
        yield "transformer.ln_f.weight", params["norm"].data
        yield "transformer.ln_f.bias", params["norm_bias"].data
        tied_copy = cfg.tie_word_embeddings and not cfg.is_critic and not model.has_embed
        if "head" in params and not tied_copy:
            if cfg.is_critic:
                yield "score.weight", params["head"].data
            else:
                yield "lm_head.weight", _gather_vocab(model, params["head"].data, g)


# -------------------------------------------------------------------------------------- save
def _gather_tp(t: torch.Tensor, dim: int, group) -> torch.Tensor:
    if group is None or dist.get_world_size(group) == 1:
        return t
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, t.contiguous(), group=group)
    return torch.cat(parts, dim)


def native_to_hf_state(model, gather: bool = True, experts_only: bool = False) -> Iterable[Tuple[str, torch.Tensor]]:
    """Yield (hf_name, full tensor) for everything this pipeline stage owns (tp shards gathered).
    `experts_only` restricts the stream to this rank's routed experts (expert-parallel ranks > 0)."""
    if experts_only:
        for name, t in native_to_hf_state(model, gather, False):
            if ".experts." in name:
                yield name, t
        return
    cfg: ModelConfig = model.cfg
    g = model.tp_group if gather else None
    if cfg.family == "gpt2":
        yield from _gpt2_state(model, g)
        return
    tp = model.tp_size
    D = cfg.head_dim
    params = dict(model.named_parameters())
    if model.has_embed:
        yield "model.embed_tokens.weight", _gather_vocab(model, params["embed"].data, g)
        # the tower is replicated over tensor-parallel ranks: every rank yields it, so parameter specs and
        # collective weight pushes agree across the tp group (checkpoint writers are tp rank 0 only)
        for n, p in params.items():
            if n.startswith("visual."):
                yield n, p.data
    for li, blk in enumerate(model.layers):
        i = model.layer_ids[li]
        names = hf_names_for_layer(cfg, i)
        pre = f"layers.{li}."
        yield names["ln1"][0], params[pre + "ln1"].data
        yield names["ln2"][0], params[pre + "ln2"].data
        hq, hkv = blk.attn.hq, blk.attn.hkv
        qkv = params[pre + "attn.qkv"].data
        q, k, v = qkv[:hq * D], qkv[hq * D:(hq + hkv) * D], qkv[(hq + hkv) * D:]
        kv_rep = max(1, tp // cfg.n_kv_heads)

        def kv_full(t):
            full = _gather_tp(t, 0, g)
            if kv_rep > 1 and g is not None:  # drop the replicas
                rows = t.shape[0]
                full = torch.cat([full[r * rows:(r + 1) * rows] for r in range(0, tp, kv_rep)], 0)
            return full

        yield names["attn.qkv"][0], _gather_tp(q, 0, g)
        yield names["attn.qkv"][1], kv_full(k)
        yield names["attn.qkv"][2], kv_full(v)
        if cfg.qkv_bias:
            b = params[pre + "attn.qkv_bias"].data
            yield names["attn.qkv_bias"][0], _gather_tp(b[:hq * D], 0, g)
            yield names["attn.qkv_bias"][1], kv_full(b[hq * D:(hq + hkv) * D])
            yield names["attn.qkv_bias"][2], kv_full(b[(hq + hkv) * D:])
        yield names["attn.o"][0], _gather_tp(params[pre + "attn.o"].data, 1, g)
        if cfg.qk_norm:
            yield names["attn.q_norm"][0], params[pre + "attn.q_norm"].data
            yield names["attn.k_norm"][0], params[pre + "attn.k_norm"].data
        if cfg.n_experts > 0:
            yield names["mlp.router"][0], params[pre + "mlp.router"].data
            gu, dn = params[pre + "mlp.gate_up"].data, params[pre + "mlp.down"].data
            f = gu.shape[1] // 2
            eg = model.etp_group if gather else None  # the ranks an expert is split over (none: whole experts)
            for le in range(gu.shape[0]):
                e = blk.mlp.ep_rank * blk.mlp.n_local + le
                gn, un, dnn = hf_expert_names(cfg, i, e)
                yield gn, _gather_tp(gu[le, :f], 0, eg)
                yield un, _gather_tp(gu[le, f:], 0, eg)
                yield dnn, _gather_tp(dn[le], 1, eg)
        else:
            gu = params[pre + "mlp.gate_up"].data
            f = gu.shape[0] // 2
            yield names["mlp.gate_up"][0], _gather_tp(gu[:f], 0, g)
            yield names["mlp.gate_up"][1], _gather_tp(gu[f:], 0, g)
            yield names["mlp.down"][0], _gather_tp(params[pre + "mlp.down"].data, 1, g)
    if model.has_head:
        yield "model.norm.weight", params["norm"].data
        tied_copy = cfg.tie_word_embeddings and not cfg.is_critic and not model.has_embed
        if "head" in params and not tied_copy:  # a tied LM head on the last pipeline stage is not a checkpoint tensor
            if cfg.is_critic:
                yield "score.weight", params["head"].data
            else:
                yield "lm_head.weight", _gather_vocab(model, params["head"].data, g)


def save_hf_checkpoint(model, path: str, tokenizer=None, stage_tag: str = "", write: bool = True,
                       max_shard_bytes: int = MAX_SHARD_BYTES, experts_only: bool = False) -> Dict[str, str]:
    """Write this stage's tensors as (possibly several) safetensors files and return its weight map.
    Every tp rank must call it (collective gather); only ranks with `write=True` touch the disk."""
    from safetensors.torch import save_file

    os.makedirs(path, exist_ok=True)
    weight_map: Dict[str, str] = {}
    cur: Dict[str, torch.Tensor] = {}
    cur_bytes = 0
    files: List[Dict[str, torch.Tensor]] = []
    for name, t in native_to_hf_state(model, experts_only=experts_only):
        t = t.detach().to("cpu").contiguous().clone() if write else None
        if not write:
            continue
        nbytes = t.numel() * t.element_size()
        if cur and cur_bytes + nbytes > max_shard_bytes:
            files.append(cur)
            cur, cur_bytes = {}, 0
        cur[name] = t
        cur_bytes += nbytes
    if cur:
        files.append(cur)
    if write:
        n = len(files)
        for i, f in enumerate(files):
            fn = f"model{stage_tag}-{i + 1:05d}-of-{n:05d}.safetensors" if (n > 1 or stage_tag) else "model.safetensors"
            save_file(f, os.path.join(path, fn), metadata={"format": "pt"})
            for k in f:
                weight_map[k] = fn
    return weight_map


def finalize_hf_checkpoint(path: str, cfg: ModelConfig, weight_maps: List[Dict[str, str]], tokenizer=None):
    """Write config.json, the index and the tokenizer (called once, by the global saver rank)."""
    merged: Dict[str, str] = {}
    for m in weight_maps:
        merged.update(m)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg.to_hf_dict(), f, indent=2)
    if len(set(merged.values())) > 1:
        with open(os.path.join(path, "model.safetensors.index.json"), "w") as f:
            json.dump({"metadata": {}, "weight_map": merged}, f, indent=2)
    if tokenizer is not None:
        tokenizer.save_pretrained(path)
