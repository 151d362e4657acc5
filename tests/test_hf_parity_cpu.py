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
    # LayerNorm with bias, learned positions, ungated GELU MLP, Conv1D (transposed) weights, fused c_attn
    "gpt2": ("GPT2Config", dict(vocab_size=320, n_embd=64, n_layer=2, n_head=4, n_inner=96, n_positions=256,
                                attn_pdrop=0.0, embd_pdrop=0.0, resid_pdrop=0.0)),
}


def _hf_model(family):
    cls_name, extra = FAMILIES[family]
    if not hasattr(transformers, cls_name):
        pytest.skip(f"transformers has no {cls_name}")
    cfg = getattr(transformers, cls_name)(**(extra if family == "gpt2" else {**COMMON, **extra}))
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


def test_qwen2_vl_matches_hf(tmp_path):
    """Vision tower + placeholder scatter + multimodal rotary positions against HF Qwen2-VL, two packed
    sequences with one image each (different grids), and the round trip of the checkpoint."""
    from areal_vllm_amd.models import hf_io
    from areal_vllm_amd.models.config import ModelConfig
    from areal_vllm_amd.models.transformer import Transformer
    from areal_vllm_amd.models.vision import mrope_position_ids

    if not hasattr(transformers, "Qwen2VLForConditionalGeneration"):
        pytest.skip("transformers has no Qwen2-VL")
    IMG, VS, VE = 390, 392, 393
    cfg = transformers.Qwen2VLConfig(
        text_config=dict(vocab_size=400, hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
                         num_key_value_heads=2, intermediate_size=96, max_position_embeddings=512,
                         rope_scaling={"type": "mrope", "mrope_section": [2, 3, 3]}, tie_word_embeddings=False,
                         bos_token_id=None, eos_token_id=1, pad_token_id=0),
        vision_config=dict(depth=2, embed_dim=32, hidden_size=64, num_heads=4, mlp_ratio=2, patch_size=14,
                           spatial_merge_size=2, temporal_patch_size=2, in_channels=3),
        image_token_id=IMG, video_token_id=391, vision_start_token_id=VS, vision_end_token_id=VE)
    cfg._attn_implementation = "eager"
    torch.manual_seed(0)
    hf = transformers.Qwen2VLForConditionalGeneration(cfg).float().eval()
    with torch.no_grad():
        for n, p in hf.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    src = str(tmp_path / "hf")
    hf.save_pretrained(src, safe_serialization=True)

    g = torch.Generator().manual_seed(2)
    grids = [(1, 4, 6), (1, 6, 4)]
    seqs, pixels = [], []
    for (t, h, w), n_text in zip(grids, (5, 9)):
        n_img = t * h * w // 4
        txt = torch.randint(3, 380, (n_text + 4,), generator=g).tolist()
        seqs.append(txt[:n_text] + [VS] + [IMG] * n_img + [VE] + txt[n_text:])
        pixels.append(torch.randn(t * h * w, 3 * 2 * 14 * 14, generator=g))
    want = []
    with torch.no_grad():
        for s, px, gr in zip(seqs, pixels, grids):
            out = hf(input_ids=torch.tensor([s]), pixel_values=px, image_grid_thw=torch.tensor([gr]),
                     mm_token_type_ids=torch.tensor([[int(t == IMG) for t in s]], dtype=torch.int32))
            want.append(out.logits[0])

    mc = ModelConfig.from_pretrained(src)
    assert mc.vision is not None and mc.mrope_section == [2, 3, 3] and mc.image_token_id == IMG
    m = Transformer(mc, dtype=torch.float32, device="cpu")
    m.flatten()
    hf_io.load_hf_weights(m, src)
    m.eval()
    lens = [len(s) for s in seqs]
    ids = torch.tensor([t for s in seqs for t in s])
    pos = torch.cat([mrope_position_ids(s, [gr], [IMG, 391], 2)[0] for s, gr in zip(seqs, grids)], 1)
    cu = torch.tensor([0, lens[0], lens[0] + lens[1]], dtype=torch.int32)
    with torch.no_grad():
        h = m.embed_multimodal(ids, torch.cat(pixels), torch.tensor(grids))
        got = m.logits(m(None, pos, cu, max(lens), hidden_in=h)).split(lens)
    for a, b in zip(got, want):
        torch.testing.assert_close(a, b, atol=3e-4, rtol=1e-3)
    # positions: text counts up, the image block is numbered by (frame, row, column), text resumes after it
    p0, delta = mrope_position_ids(seqs[0], [grids[0]], [IMG], 2)
    assert p0[:, :6].tolist() == [list(range(6))] * 3 and p0[1, 6:12].tolist() == [6, 6, 6, 7, 7, 7]
    assert p0[2, 6:12].tolist() == [6, 7, 8, 6, 7, 8] and int(p0[0, 12]) == 9 and delta == int(p0.max()) + 1 - lens[0]

    dst = str(tmp_path / "native")
    hf_io.finalize_hf_checkpoint(dst, m.cfg, [hf_io.save_hf_checkpoint(m, dst)])
    back = transformers.Qwen2VLForConditionalGeneration.from_pretrained(dst, torch_dtype=torch.float32).eval()
    with torch.no_grad():
        again = back(input_ids=torch.tensor([seqs[0]]), pixel_values=pixels[0], image_grid_thw=torch.tensor([grids[0]]),
                     mm_token_type_ids=torch.tensor([[int(t == IMG) for t in seqs[0]]], dtype=torch.int32)).logits[0]
    torch.testing.assert_close(again, want[0], atol=3e-4, rtol=1e-3)
