"""Offline evaluation pipeline on CPU: a tiny random checkpoint with a word-level tokenizer is served by the
native engine, answers are generated (greedy and sampled), scored by the math verifier and aggregated over
two benchmark files into one table (counterpart of running the reference's evaluation/eval_and_aggregate.py)."""
import json
import os

import pytest
import torch


def _tiny_checkpoint(path):
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast

    from areal_vllm_amd.models import hf_io
    from areal_vllm_amd.models.config import preset
    from areal_vllm_amd.models.transformer import Transformer

    words = ["<pad>", "<eos>", "<unk>"] + [str(i) for i in range(100)] + \
        "what is plus minus the answer boxed { } \\ + - = ? .".split()
    tok = Tokenizer(models.WordLevel({w: i for i, w in enumerate(words)}, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, pad_token="<pad>", eos_token="<eos>", unk_token="<unk>")
    cfg = preset("tiny", vocab_size=128, eos_token_id=1, pad_token_id=0)
    m = Transformer(cfg, dtype=torch.float32, device="cpu")
    m.flatten()
    m.init_weights(seed=3)
    hf_io.finalize_hf_checkpoint(path, cfg, [hf_io.save_hf_checkpoint(m, path)], tokenizer=fast)


def test_math_eval_and_aggregate_on_cpu(tmp_path, monkeypatch):
    monkeypatch.setenv("AREAL_PLATFORM", "cpu")
    from areal_vllm_amd.evaluation import eval_and_aggregate
    from areal_vllm_amd.evaluation.math_eval import evaluate_checkpoint, pass_at_k

    ckpt = str(tmp_path / "ckpt")
    _tiny_checkpoint(ckpt)
    data = tmp_path / "data"
    for name, n in (("toy_a", 3), ("toy_b", 2)):
        os.makedirs(data / name)
        with open(data / name / "test.jsonl", "w") as f:
            for i in range(n):
                f.write(json.dumps(dict(query_id=f"{name}-{i}", prompt=f"what is {i} plus {i + 1} ?",
                                        solutions=[f"\\boxed{{{2 * i + 1}}}"])) + "\n")
    out = str(tmp_path / "one.json")
    m = evaluate_checkpoint(ckpt, str(data / "toy_a" / "test.jsonl"), n_sampling=2, max_new_tokens=6, temperature=1.0,
                            output=out, device="cpu", max_running=8)
    assert {"mean@2", "pass@1", "pass@2", "maj@2", "avg_gen_tokens", "truncated_ratio"} <= set(m)
    assert 0.0 <= m["mean@2"] <= 1.0 and 0 < m["avg_gen_tokens"] <= 6
    saved = json.load(open(out))
    assert len(saved["samples"]) == 3 and all(len(s["answers"]) == 2 and len(s["correct"]) == 2 for s in saved["samples"])
    assert pass_at_k(4, 0, 2) == 0.0 and pass_at_k(4, 4, 2) == 1.0 and pass_at_k(4, 1, 2) == pytest.approx(0.5)

    res = eval_and_aggregate.main(["--model-path", ckpt, "--data-dir", str(data), "--data-names", "toy_a,toy_b",
                                   "--n-sampling", "2", "--max-gen-tokens", "6", "--prompt-type", "plain",
                                   "--device", "cpu", "--output-path", str(tmp_path / "out")])
    assert set(res) == {"toy_a", "toy_b"} and {"greedy_acc", "greedy_length", "sample_pass@1", "sample_length"} <= set(res["toy_a"])
    cached = tmp_path / "out" / "math_eval_6" / "aggregate_plain_0.6_0.95.json"
    assert cached.exists()
    # a second call reads the cached table instead of generating again
    again = eval_and_aggregate.main(["--model-path", "/nonexistent", "--data-dir", str(data), "--data-names", "toy_a,toy_b",
                                     "--n-sampling", "2", "--max-gen-tokens", "6", "--prompt-type", "plain",
                                     "--output-path", str(tmp_path / "out")])
    assert again == res
