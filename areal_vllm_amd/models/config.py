"""Architecture description of the native transformer and its mapping from HF ``config.json``.

Covers the dense families the reference converts in ``realhf/api/from_hf/{llama,qwen2,qwen3,mistral,
gemma,gpt2}.py`` plus MoE (``mixtral``, ``qwen3_moe``) through the ``n_experts`` fields. GPT-2 differs
from the rest in four switches: LayerNorm with bias (``norm_type``), learned absolute positions
(``pos_embed``), a plain two-matrix MLP (``mlp_gated``) and biases on every linear (``linear_bias``). The reference's
``ReaLModelConfig`` (``realhf/api/core/model_api.py:340``) is the analogue.
"""
import dataclasses
import json
import os
from typing import Any, Dict, List, Optional


@dataclasses.dataclass
class ModelConfig:
    family: str = "qwen2"
    vocab_size: int = 151936
    hidden_size: int = 1536
    n_layers: int = 28
    n_q_heads: int = 12
    n_kv_heads: int = 2
    head_dim: int = 128
    intermediate_size: int = 8960
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1000000.0
    rope_scaling: float = 1.0
    max_position_embeddings: int = 32768
    tie_word_embeddings: bool = True
    qkv_bias: bool = True
    qk_norm: bool = False  # per-head RMSNorm on q and k (Qwen3)
    gemma_norm: bool = False  # (1 + w) RMSNorm weights and sqrt(H) embedding scale (Gemma)
    hidden_act: str = "silu"  # gate activation of the MLP: "silu" (SwiGLU) or "gelu_tanh" (GeGLU, Gemma)
    norm_type: str = "rms"  # "rms" or "layer" (LayerNorm with bias, GPT-2)
    pos_embed: str = "rope"  # "rope" or "learned" (absolute position table added to the embeddings, GPT-2)
    mlp_gated: bool = True  # False: fc -> activation -> proj without a gate (GPT-2)
    linear_bias: bool = False  # biases on the attention output and MLP projections (GPT-2)
    is_critic: bool = False  # value head [1, H] instead of the LM head
    # mixture of experts (0 = dense)
    n_experts: int = 0
    n_experts_per_tok: int = 0
    moe_intermediate_size: int = 0
    norm_topk_prob: bool = True
    router_aux_loss_coef: float = 0.0
    # router options of the reference's native MoE (``realhf/api/core/model_api.py:302-335``,
    # ``realhf/impl/model/modules/moe/router.py``): "aux_loss" (top-k of the softmax + load-balancing loss), "sinkhorn"
    # (balanced assignment by Sinkhorn normalisation while training) or "none"; z-loss on the router logits; a capacity
    # of capacity_factor * tokens * k / experts per expert, pairs beyond it dropped by lowest probability ("probs") or
    # latest position ("position"); multiplicative input jitter in [1 - eps, 1 + eps] on the router's input
    moe_routing_type: str = "aux_loss"
    moe_z_loss_coef: float = 0.0
    moe_capacity_factor: Optional[float] = None
    moe_token_drop_policy: str = "probs"
    moe_pad_to_capacity: bool = False  # accepted: dispatch here is shape-dynamic, padding would change no result
    moe_input_jitter_eps: Optional[float] = None
    # dropout (GPT-2 style checkpoints carry non-zero values; RL recipes switch it off with `disable_dropout`)
    embd_pdrop: float = 0.0
    resid_pdrop: float = 0.0
    attn_pdrop: float = 0.0
    bos_token_id: Optional[int] = None
    eos_token_id: Optional[int] = None
    pad_token_id: Optional[int] = None
    # vision-language models (Qwen2-VL layout): settings of the vision tower (models/vision.py) and the
    # split of the rotary pairs over (temporal, row, column) positions; None = text-only
    vision: Optional[Dict[str, Any]] = None
    mrope_section: Optional[List[int]] = None
    image_token_id: Optional[int] = None
    video_token_id: Optional[int] = None
    vision_start_token_id: Optional[int] = None
    vision_end_token_id: Optional[int] = None

    @property
    def q_dim(self) -> int:
        return self.n_q_heads * self.head_dim

    @property
    def kv_dim(self) -> int:
        return self.n_kv_heads * self.head_dim

    @property
    def qkv_dim(self) -> int:
        return self.q_dim + 2 * self.kv_dim

    def n_params(self, with_embedding: bool = True) -> int:
        H, F, L = self.hidden_size, self.intermediate_size, self.n_layers
        if not self.mlp_gated:  # GPT-2 layout: biases everywhere, two norms with bias, fc / proj
            per_layer = H * self.qkv_dim + self.qkv_dim + self.q_dim * H + H + 4 * H + 2 * H * F + F + H
            n = L * per_layer + 2 * H
            if with_embedding:
                n += (self.vocab_size + self.max_position_embeddings) * H
                if not self.tie_word_embeddings and not self.is_critic:
                    n += self.vocab_size * H
            return n + (H if self.is_critic else 0)
        per_layer = H * self.qkv_dim + (self.qkv_dim if self.qkv_bias else 0) + self.q_dim * H + 2 * H
        if self.n_experts > 0:
            per_layer += H * self.n_experts + self.n_experts * 3 * H * self.moe_intermediate_size
        else:
            per_layer += 3 * H * F
        if self.qk_norm:
            per_layer += 2 * self.head_dim
        n = L * per_layer + H
        if with_embedding:
            n += self.vocab_size * H
            if not self.tie_word_embeddings and not self.is_critic:
                n += self.vocab_size * H
        if self.is_critic:
            n += H
        return n

    def n_active_params(self, with_embedding: bool = True) -> int:
        """Parameters one token passes through: for mixture-of-experts models the routed experts count
        `n_experts_per_tok` of `n_experts` times (FLOP accounting); equal to `n_params` for dense models."""
        n = self.n_params(with_embedding)
        if self.n_experts > 0:
            n -= self.n_layers * (self.n_experts - self.n_experts_per_tok) * 3 * self.hidden_size * self.moe_intermediate_size
        return n

    # ------------------------------------------------------------------ HF interop
    @classmethod
    def from_hf_dict(cls, d: Dict[str, Any], is_critic: bool = False) -> "ModelConfig":
        if d.get("model_type") in ("qwen2_vl", "qwen2_5_vl") and "text_config" in d:
            # nested layout: language model settings under text_config, tower under vision_config
            text = dict(d["text_config"])
            text["model_type"] = "qwen2"
            text.setdefault("tie_word_embeddings", d.get("tie_word_embeddings", False))
            cfg = cls.from_hf_dict(text, is_critic=is_critic)
            cfg.family = d["model_type"]
            cfg.vision = dict(d["vision_config"])
            rp = text.get("rope_parameters") or text.get("rope_scaling") or {}
            cfg.mrope_section = list(rp.get("mrope_section") or []) or None
            for k in ("image_token_id", "video_token_id", "vision_start_token_id", "vision_end_token_id"):
                setattr(cfg, k, d.get(k))
            cfg.qkv_bias = bool(text.get("attention_bias", True))
            return cfg
        mt = d.get("model_type", "qwen2")
        if mt == "gpt2":
            H, nq = d["n_embd"], d["n_head"]
            return cls(family="gpt2", vocab_size=d["vocab_size"], hidden_size=H, n_layers=d["n_layer"], n_q_heads=nq,
                       n_kv_heads=nq, head_dim=H // nq, intermediate_size=d.get("n_inner") or 4 * H,
                       rms_norm_eps=d.get("layer_norm_epsilon", 1e-5), max_position_embeddings=d.get("n_positions", 1024),
                       tie_word_embeddings=bool(d.get("tie_word_embeddings", True)), qkv_bias=True,
                       hidden_act="gelu_tanh", norm_type="layer", pos_embed="learned", mlp_gated=False,
                       linear_bias=True, is_critic=is_critic, embd_pdrop=float(d.get("embd_pdrop", 0.0) or 0.0),
                       resid_pdrop=float(d.get("resid_pdrop", 0.0) or 0.0), attn_pdrop=float(d.get("attn_pdrop", 0.0) or 0.0),
                       bos_token_id=d.get("bos_token_id"),
                       eos_token_id=d.get("eos_token_id"), pad_token_id=d.get("pad_token_id"))
        H = d["hidden_size"]
        nq = d["num_attention_heads"]
        nkv = d.get("num_key_value_heads", nq)
        hd = d.get("head_dim") or H // nq
        fam_bias = {"qwen2": True, "qwen2_moe": True}
        # transformers >= 5 nests the rotary settings in `rope_parameters`
        rp = d.get("rope_parameters") or {}
        rs = d.get("rope_scaling") or rp
        scaling = float(rs.get("factor", 1.0)) if rs.get("type", rs.get("rope_type")) == "linear" else 1.0
        act = d.get("hidden_activation") or d.get("hidden_act") or "silu"
        cfg = cls(
            family=mt,
            vocab_size=d["vocab_size"],
            hidden_size=H,
            n_layers=d["num_hidden_layers"],
            n_q_heads=nq,
            n_kv_heads=nkv,
            head_dim=hd,
            intermediate_size=d.get("intermediate_size", 4 * H),
            rms_norm_eps=d.get("rms_norm_eps", 1e-6),
            rope_theta=float(d.get("rope_theta") or rp.get("rope_theta") or 10000.0),
            rope_scaling=scaling,
            max_position_embeddings=d.get("max_position_embeddings", 32768),
            tie_word_embeddings=bool(d.get("tie_word_embeddings", False)),
            qkv_bias=bool(d.get("attention_bias", fam_bias.get(mt, False))),
            qk_norm=mt in ("qwen3", "qwen3_moe"),
            gemma_norm=mt in ("gemma", "gemma2"),
            hidden_act="gelu_tanh" if act in ("gelu_pytorch_tanh", "gelu_tanh", "gelu") else "silu",
            is_critic=is_critic,
            n_experts=d.get("num_local_experts", d.get("num_experts", 0)) or 0,
            n_experts_per_tok=d.get("num_experts_per_tok", 0) or 0,
            moe_intermediate_size=d.get("moe_intermediate_size", d.get("intermediate_size", 0))
            if (d.get("num_local_experts") or d.get("num_experts")) else 0,
            norm_topk_prob=bool(d.get("norm_topk_prob", True)),
            router_aux_loss_coef=float(d.get("router_aux_loss_coef", 0.0) or 0.0),
            moe_z_loss_coef=float(d.get("router_z_loss_coef", 0.0) or 0.0),
            moe_input_jitter_eps=d.get("router_jitter_noise") or None,
            attn_pdrop=float(d.get("attention_dropout", 0.0) or 0.0),
            bos_token_id=d.get("bos_token_id"),
            eos_token_id=d.get("eos_token_id") if not isinstance(d.get("eos_token_id"), list) else d["eos_token_id"][0],
            pad_token_id=d.get("pad_token_id"),
        )
        return cfg

    @classmethod
    def from_pretrained(cls, path: str, is_critic: bool = False) -> "ModelConfig":
        with open(os.path.join(path, "config.json")) as f:
            return cls.from_hf_dict(json.load(f), is_critic=is_critic)

    def to_hf_dict(self) -> Dict[str, Any]:
        if self.vision is not None:
            text = dataclasses.replace(self, vision=None, family="qwen2").to_hf_dict()
            text.pop("architectures")
            text["model_type"] = "qwen2_vl_text"
            text["rope_parameters"] = {"rope_type": "default", "type": "mrope", "rope_theta": self.rope_theta,
                                       "mrope_section": list(self.mrope_section or [])}
            text["rope_scaling"] = {"type": "mrope", "mrope_section": list(self.mrope_section or [])}
            vis = dict(self.vision)
            vis.setdefault("model_type", "qwen2_vl_vision")
            return {"architectures": ["Qwen2VLForConditionalGeneration"], "model_type": self.family,
                    "text_config": text, "vision_config": vis, "tie_word_embeddings": self.tie_word_embeddings,
                    "image_token_id": self.image_token_id, "video_token_id": self.video_token_id,
                    "vision_start_token_id": self.vision_start_token_id,
                    "vision_end_token_id": self.vision_end_token_id, "torch_dtype": "bfloat16"}
        if self.family == "gpt2":
            return {"architectures": ["GPT2LMHeadModel"], "model_type": "gpt2", "vocab_size": self.vocab_size,
                    "n_embd": self.hidden_size, "n_layer": self.n_layers, "n_head": self.n_q_heads,
                    "n_inner": self.intermediate_size, "n_positions": self.max_position_embeddings,
                    "n_ctx": self.max_position_embeddings, "layer_norm_epsilon": self.rms_norm_eps,
                    "activation_function": "gelu_new", "tie_word_embeddings": self.tie_word_embeddings,
                    "torch_dtype": "bfloat16", "bos_token_id": self.bos_token_id, "eos_token_id": self.eos_token_id,
                    "pad_token_id": self.pad_token_id}
        arch = {
            "qwen2": "Qwen2ForCausalLM", "qwen3": "Qwen3ForCausalLM", "llama": "LlamaForCausalLM",
            "mistral": "MistralForCausalLM", "gemma": "GemmaForCausalLM", "mixtral": "MixtralForCausalLM",
            "qwen3_moe": "Qwen3MoeForCausalLM", "qwen2_moe": "Qwen2MoeForCausalLM",
        }.get(self.family, "Qwen2ForCausalLM")
        d = {
            "architectures": [arch],
            "model_type": self.family,
            "vocab_size": self.vocab_size,
            "hidden_size": self.hidden_size,
            "num_hidden_layers": self.n_layers,
            "num_attention_heads": self.n_q_heads,
            "num_key_value_heads": self.n_kv_heads,
            "head_dim": self.head_dim,
            "intermediate_size": self.intermediate_size,
            "rms_norm_eps": self.rms_norm_eps,
            "rope_theta": self.rope_theta,
            "rope_parameters": {"rope_type": "default", "rope_theta": self.rope_theta},
            "max_position_embeddings": self.max_position_embeddings,
            "tie_word_embeddings": self.tie_word_embeddings,
            "attention_bias": self.qkv_bias,
            "hidden_act": "gelu_pytorch_tanh" if self.hidden_act == "gelu_tanh" else "silu",
            "torch_dtype": "bfloat16",
            "bos_token_id": self.bos_token_id,
            "eos_token_id": self.eos_token_id,
            "pad_token_id": self.pad_token_id,
        }
        if self.hidden_act == "gelu_tanh":
            d["hidden_activation"] = "gelu_pytorch_tanh"
        if self.n_experts:
            d["num_local_experts" if self.family == "mixtral" else "num_experts"] = self.n_experts
            d["num_experts_per_tok"] = self.n_experts_per_tok
            d["moe_intermediate_size"] = self.moe_intermediate_size
            d["norm_topk_prob"] = self.norm_topk_prob
            d["router_aux_loss_coef"] = self.router_aux_loss_coef
        return d


# Named presets (public model-card dimensions) for random-init benchmarking without checkpoints.
PRESETS: Dict[str, Dict[str, Any]] = {
    "qwen2-1.5b": dict(family="qwen2", vocab_size=151936, hidden_size=1536, n_layers=28, n_q_heads=12, n_kv_heads=2,
                       head_dim=128, intermediate_size=8960, tie_word_embeddings=True, qkv_bias=True,
                       rope_theta=1e6, eos_token_id=151645, pad_token_id=151643),
    "qwen2.5-7b": dict(family="qwen2", vocab_size=152064, hidden_size=3584, n_layers=28, n_q_heads=28, n_kv_heads=4,
                       head_dim=128, intermediate_size=18944, tie_word_embeddings=False, qkv_bias=True,
                       rope_theta=1e6, eos_token_id=151645, pad_token_id=151643),
    "qwen2.5-32b": dict(family="qwen2", vocab_size=152064, hidden_size=5120, n_layers=64, n_q_heads=40, n_kv_heads=8,
                        head_dim=128, intermediate_size=27648, tie_word_embeddings=False, qkv_bias=True,
                        rope_theta=1e6, eos_token_id=151645, pad_token_id=151643),
    "llama3-8b": dict(family="llama", vocab_size=128256, hidden_size=4096, n_layers=32, n_q_heads=32, n_kv_heads=8,
                      head_dim=128, intermediate_size=14336, tie_word_embeddings=False, qkv_bias=False,
                      rope_theta=5e5, rms_norm_eps=1e-5, eos_token_id=128001),
    "tiny": dict(family="qwen2", vocab_size=512, hidden_size=256, n_layers=2, n_q_heads=2, n_kv_heads=1,
                 head_dim=128, intermediate_size=512, tie_word_embeddings=True, qkv_bias=True, rope_theta=1e4,
                 max_position_embeddings=4096, eos_token_id=1, pad_token_id=0),
    "tiny-deep": dict(family="qwen2", vocab_size=512, hidden_size=256, n_layers=5, n_q_heads=2, n_kv_heads=1,
                      head_dim=128, intermediate_size=512, tie_word_embeddings=True, qkv_bias=True, rope_theta=1e4,
                      max_position_embeddings=4096, eos_token_id=1, pad_token_id=0),
    "small": dict(family="qwen2", vocab_size=4096, hidden_size=512, n_layers=4, n_q_heads=4, n_kv_heads=2,
                  head_dim=128, intermediate_size=1536, tie_word_embeddings=True, qkv_bias=True, rope_theta=1e4,
                  max_position_embeddings=4096, eos_token_id=1, pad_token_id=0),
    "tiny-qknorm": dict(family="qwen3", vocab_size=512, hidden_size=256, n_layers=2, n_q_heads=4, n_kv_heads=2,
                        head_dim=64, intermediate_size=512, tie_word_embeddings=False, qkv_bias=False, qk_norm=True,
                        rope_theta=1e4, max_position_embeddings=4096, eos_token_id=1, pad_token_id=0),
    "tiny-gemma": dict(family="gemma", vocab_size=512, hidden_size=256, n_layers=2, n_q_heads=2, n_kv_heads=1,
                       head_dim=256, intermediate_size=512, tie_word_embeddings=True, qkv_bias=False,
                       hidden_act="gelu_pytorch_tanh", gemma_norm=True, rope_theta=1e4, max_position_embeddings=4096,
                       eos_token_id=1, pad_token_id=0),
    "tiny-gpt2": dict(family="gpt2", vocab_size=512, hidden_size=128, n_layers=2, n_q_heads=2, n_kv_heads=2,
                      head_dim=64, intermediate_size=512, tie_word_embeddings=True, qkv_bias=True,
                      hidden_act="gelu_tanh", norm_type="layer", pos_embed="learned", mlp_gated=False,
                      linear_bias=True, rms_norm_eps=1e-5, max_position_embeddings=1024, eos_token_id=1,
                      pad_token_id=0),
    "tiny-vl": dict(family="qwen2_vl", vocab_size=512, hidden_size=256, n_layers=2, n_q_heads=2, n_kv_heads=1,
                    head_dim=128, intermediate_size=512, tie_word_embeddings=False, qkv_bias=True, rope_theta=1e4,
                    max_position_embeddings=4096, eos_token_id=1, pad_token_id=0, mrope_section=[16, 24, 24],
                    image_token_id=500, video_token_id=501, vision_start_token_id=502, vision_end_token_id=503,
                    vision=dict(depth=2, embed_dim=64, hidden_size=256, num_heads=4, mlp_ratio=2, patch_size=14,
                                spatial_merge_size=2, temporal_patch_size=2, in_channels=3)),
    # Qwen3-30B-A3B (128 experts of width 768, 8 per token) and a 12-layer cut of it that trains on one GPU with
    # fp32 master weights and Adam state resident (expert GEMM benchmarking)
    "qwen3-30b-a3b": dict(family="qwen3_moe", vocab_size=151936, hidden_size=2048, n_layers=48, n_q_heads=32,
                          n_kv_heads=4, head_dim=128, intermediate_size=6144, tie_word_embeddings=False,
                          qkv_bias=False, qk_norm=True, rope_theta=1e6, max_position_embeddings=40960,
                          rms_norm_eps=1e-6, n_experts=128, n_experts_per_tok=8, moe_intermediate_size=768,
                          norm_topk_prob=True, eos_token_id=151645, pad_token_id=151643),
    "qwen3-moe-a3b-12l": dict(family="qwen3_moe", vocab_size=151936, hidden_size=2048, n_layers=12, n_q_heads=32,
                              n_kv_heads=4, head_dim=128, intermediate_size=6144, tie_word_embeddings=False,
                              qkv_bias=False, qk_norm=True, rope_theta=1e6, max_position_embeddings=40960,
                              rms_norm_eps=1e-6, n_experts=128, n_experts_per_tok=8, moe_intermediate_size=768,
                              norm_topk_prob=True, eos_token_id=151645, pad_token_id=151643),
    "tiny-moe": dict(family="qwen3_moe", vocab_size=512, hidden_size=256, n_layers=2, n_q_heads=2, n_kv_heads=1,
                     head_dim=128, intermediate_size=512, tie_word_embeddings=False, qkv_bias=False, qk_norm=True,
                     rope_theta=1e4, max_position_embeddings=4096, n_experts=4, n_experts_per_tok=2,
                     moe_intermediate_size=256, eos_token_id=1, pad_token_id=0),
}


def preset(name: str, **overrides) -> ModelConfig:
    d = dict(PRESETS[name.lower()])
    d.update(overrides)
    return ModelConfig(**d)
