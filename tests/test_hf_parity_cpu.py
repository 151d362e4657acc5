"""The native transformer against Hugging Face ``transformers`` reference implementations on CPU (fp32):
a tiny random model of every supported family is written by HF, loaded through ``hf_io`` into the native
model and must give the same logits; the checkpoint written back by ``hf_io`` must load into HF and
give the same logits again. Counterpart of ``realhf/tests/model/test_cpu_inference.py``."""
import os

import pytest
import torch

transformers = pytest.importorskip("transformers")

COMMON = dict(vocab_size=320, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
              intermediate_size=96, max_position_embeddings=256, rms_norm_eps=1e-6)

FAMILIES = {
    "qwen2": ("Qwen2Config", dict(tie_word_embeddings=True)),
    "qwen3": ("Qwen3Config", dict(head_dim=32, tie_word_embeddings=False)),
    "llama": ("LlamaConfig", dict(tie_word_embeddings=False, rms_norm_eps=1e-5)),
    "mistral": ("MistralConfig", dict(tie_word_embeddings=False, sliding_window=None)),
    "gemma": ("GemmaConfig", dict(head_dim=16, tie_word_embeddings=True)),
    "mixtral": ("MixtralConfig", dict(num_local_experts=4, num_experts_per_tok=2, tie_word_embeddings=False,
                                      sliding_window=None)),
    "qwen3_moe": ("Qwen3MoeConfig", dict(head_dim=32, num_experts=4, num_experts_per_tok=2, moe_intermediate_size=48,
                                         norm_topk_prob=True, decoder_sparse_step=1, mlp_only_layers=[],
                                         tie_word_embeddings=False)),
}


def _hf_model(family):
    cls_name, extra = FAMILIES[family]
    if not hasattr(transformers, cls_name):
        pytest.skip(f"transformers has no {cls_name}")
    cfg = getattr(transformers, cls_name)(**{**COMMON, **extra})
    cfg._attn_implementation = "eager"
    torch.manual_seed(0)
    model = transformers.AutoModelForCausalLM.from_config(cfg).float().eval()
    with torch.no_grad():  # random norm weights / biases so that every parameter matters
        for n, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    return model


def _native_logits(path, ids_list):
    from areal_vllm_amd.models import hf_io
    from areal_vllm_amd.models.config import ModelConfig
    from areal_vllm_amd.models.transformer import Transformer

    cfg = ModelConfig.from_pretrained(path)
    m = Transformer(cfg, dtype=torch.float32, device="cpu")
    m.flatten()
    hf_io.load_hf_weights(m, path)
    m.eval()
    lens = [len(x) for x in ids_list]
    ids = torch.tensor([t for x in ids_list for t in x])
    pos = torch.cat([torch.arange(n, dtype=torch.int32) for n in lens])
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    with torch.no_grad():
        logits = m.logits(m(ids, pos, cu, max(lens)))
    return m, list(logits.split(lens))


@pytest.mark.parametrize("family", list(FAMILIES))
def test_native_model_matches_hf(family, tmp_path):
    from areal_vllm_amd.models import hf_io

    hf = _hf_model(family)
    src = str(tmp_path / "hf")
    hf.save_pretrained(src, safe_serialization=True)
    g = torch.Generator().manual_seed(1)
    seqs = [torch.randint(3, 300, (n,), generator=g).tolist() for n in (17, 5, 33)]
    with torch.no_grad():
        want = [hf(torch.tensor([s])).logits[0] for s in seqs]
    model, got = _native_logits(src, seqs)
    for a, b in zip(got, want):
        torch.testing.assert_close(a, b, atol=2e-4, rtol=1e-3)
    # native -> HF: the written checkpoint loads into transformers and reproduces the logits
    dst = str(tmp_path / "native")
    hf_io.finalize_hf_checkpoint(dst, model.cfg, [hf_io.save_hf_checkpoint(model, dst)])
    back = transformers.AutoModelForCausalLM.from_pretrained(dst, torch_dtype=torch.float32).eval()
    with torch.no_grad():
        again = back(torch.tensor([seqs[0]])).logits[0]
    torch.testing.assert_close(again, want[0], atol=2e-4, rtol=1e-3)
