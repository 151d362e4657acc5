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


def test_benchmark_formats_normalise_to_evaluator_rows(tmp_path):
    """Raw records in the field layouts of the public test sets (synthetic rows of those shapes) come out as
    {query_id, prompt, solutions} rows that the reward parser scores; multiple-choice sets score by letter."""
    from areal_vllm_amd.evaluation import benchmarks as B
    from areal_vllm_amd.reward.math_parser import math_equal, extract_answer, last_boxed_content

    raw = {
        "gsm8k": dict(question="Tom has 3 apples and buys 4. How many?", answer="3 + 4 = 7\n#### 7"),
        "math_500": dict(problem="Compute $\\frac12+\\frac12$.", solution="We add: $\\boxed{1}$", unique_id="t/1"),
        "svamp": dict(Body="There are 5 birds", Question="2 fly away. How many stay?", Equation="5-2", Answer=3.0),
        "asdiv": dict(body="A box holds 12 eggs.", question="How many in 3 boxes?", formula="12*3", answer="36 (eggs)"),
        "mawps": dict(input="7 plus 8?", target=15.0),
        "tabmwp": dict(table_title="Fruit", table="apple | 3\npear | 1/2", question="pears?", choices=None,
                       answer="1/2", ans_type="decimal_number", solution="read it"),
        "carp_en": dict(content="Area of a 2 cm square?", steps="2*2", answer="4 cm^2"),
        "mmlu_stem": dict(question="2+2?", choices=[3, 4, 5, 6], answer=1),
        "sat_math": dict(question="x+1=3, x?", options="A) 1 B) 2 C) 3 D) 4", Answer="B"),
        "aqua": dict(question="Half of 43?", options=["A)21", "B)21.5", "C)22", "D)20", "E)23"], correct="B"),
        "gaokao_math_qa": dict(question="1+1=?", options={"A": "1", "B": "2"}, label="B"),
        "gaokao2023en": dict(question="Solve x^2=4, x>0.", answer="$2$"),
        "gaokao2024_mix": dict(question="Pick.", choice_answer="C", answer=""),
        "olympiadbench": dict(question="Find n.", final_answer=["$12$"]),
        "aime24": dict(problem="Find the remainder.", answer="204"),
        "train_amc_aime": dict(prompt="What is 6*7?", final_answer="42"),
    }
    rows = {k: B.normalise_row(v, k, 5) for k, v in raw.items()}
    truth = {k: last_boxed_content(r["solutions"][0]) for k, r in rows.items()}
    assert truth["gsm8k"] == "7" and truth["math_500"] == "1" and truth["svamp"] == "3" and truth["asdiv"] == "36"
    assert truth["mawps"] == "15" and truth["tabmwp"] == "\\frac{1}{2}" and truth["olympiadbench"] == "12"
    assert truth["carp_en"] == "4cm^2"  # units are kept for this set
    assert truth["mmlu_stem"] == "B" and truth["aqua"] == "B" and truth["gaokao2024_mix"] == "C"
    assert rows["math_500"]["query_id"] == "t/1" and rows["gsm8k"]["query_id"] == "5"
    assert rows["svamp"]["prompt"] == "There are 5 birds. 2 fly away. How many stay?"
    assert rows["mmlu_stem"]["prompt"].endswith("Answer Choices: (A) 3 (B) 4 (C) 5 (D) 6")
    assert rows["sat_math"]["prompt"].endswith("Answer Choices: (A) 1 (B) 2 (C) 3 (D) 4")
    assert rows["aqua"]["prompt"].endswith("Answer Choices: (A) 21 (B) 21.5 (C) 22 (D) 20 (E) 23")
    assert "选项: (A) 1 (B) 2" in rows["gaokao_math_qa"]["prompt"]
    assert rows["tabmwp"]["prompt"].startswith('Read the following table regarding "Fruit" and answer')
    assert B.normalise_row(dict(question="Is 7 prime?", answer="yes"), "aime24")["prompt"].endswith("(Yes or No)")
    with pytest.raises(KeyError):
        B.normalise_row({}, "no_such_set")
    # the normalised truth is what the scorer accepts
    assert math_equal(extract_answer("so the answer is \\boxed{0.5}"), truth["tabmwp"])
    assert math_equal(extract_answer("\\boxed{7.0}"), truth["gsm8k"])
    assert B.score_choice("I think (A) is wrong, so the answer is \\boxed{B}", truth["aqua"])
    assert B.score_choice("The answer is (B).", "B") and not B.score_choice("Answer: C", "B")
    # files: duplicates dropped, start / limit applied
    d = tmp_path / "gsm8k"
    d.mkdir()
    recs = [dict(raw["gsm8k"], idx=i % 3) for i in range(5)]
    (d / "test.jsonl").write_text("\n".join(json.dumps(r) for r in recs))
    assert [r["query_id"] for r in B.load_benchmark(str(tmp_path), "gsm8k")] == ["0", "1", "2"]
    assert [r["query_id"] for r in B.load_benchmark(str(tmp_path), "gsm8k", start=1, limit=1)] == ["1"]
    p = B.few_shot_prompt("2+2?", [("1+1?", "1+1=2. The answer is 2")], instruction="Solve.")
    assert p == "Solve.\n\nQuestion: 1+1?\nAnswer: 1+1=2. The answer is 2\n\nQuestion: 2+2?\nAnswer:"


def test_majority_and_reward_model_voting(tmp_path):
    from areal_vllm_amd.evaluation import voting

    # string grouping keeps 2/4 apart from 1/2; symbolic grouping merges them
    groups, top = voting.group_predictions(["\\frac{1}{2}", "3", "\\frac{1}{2}", "\\frac{2}{4}"], normalise=True)
    assert top == "\\frac{1}{2}" and sorted(map(len, groups.values())) == [1, 1, 2]
    groups, top = voting.group_predictions(["0.5", "3", "\\frac{1}{2}", "3", "\\frac{2}{4}"], symbolic=True)
    assert top == "0.5" and sorted(map(len, groups.values())) == [2, 3]
    rows = [dict(pred=["4", "5", "4", "7"], score=[1, 0, 1, 0], pred_score=[[0.1], [0.9], [0.2], [0.3]]),
            dict(pred=["1", "2", "2", "3"], score=[1, 0, 0, 0], pred_score=[0.8, 0.1, 0.2, 0.3])]
    assert voting.maj_at_k(rows, 4) == 50.0  # first problem's vote is right, second's is not
    assert voting.rm_at_k(rows, 4) == 50.0  # the reward model picks a wrong sample first, the right one second
    assert voting.maj_at_k(rows, 1) == 100.0
    p = tmp_path / "s.jsonl"
    p.write_text("\n".join(json.dumps(r) for r in rows))
    assert voting.main(["--samples", str(p), "-k", "4"]) == {"maj@4": 50.0, "rm@4": 50.0}


def test_accuracy_from_rollout_dumps(tmp_path):
    from areal_vllm_amd.evaluation import voting

    (tmp_path / "a.jsonl").write_text("\n".join(json.dumps(x) for x in [
        dict(query_id="q1@idx:0", reward=5.0), dict(query_id="q1@idx:1", reward=-5.0), dict(prompt="p", reward=1.0)]))
    (tmp_path / "b.jsonl").write_text(json.dumps(dict(prompt="p", reward=2.0)) + "\n")
    agg = voting.accuracy_from_rollout_dumps(str(tmp_path))
    assert agg["per_query"]["q1"] == (2, 0.5) and len(agg["per_query"]) == 2
    assert abs(agg["mean_accuracy"] - 0.75) < 1e-9


def test_tool_integrated_transcripts():
    from areal_vllm_amd.evaluation import voting

    t = ("Let me compute.\n```python\nx = 2\nprint(x)\n```\n```output\n2\n```\nNow a slip.\n```python\ny = 1/0\n```\n"
         "```output\nTraceback (most recent call last):\nZeroDivisionError: division by zero\n```\n"
         "Again.\n```python\nprint(x + 1)\n```\n")
    pieces = voting.split_trajectory(t)
    assert [p["role"] for p in pieces] == ["text", "program", "output", "text", "program", "output", "text", "program"]
    assert voting.split_trajectory(voting.join_trajectory(pieces)) == pieces
    assert voting.program_of(t, last_only=True) == "print(x + 1)"
    prog = voting.program_of(t)
    assert "1/0" not in prog and "x = 2" in prog
    from areal_vllm_amd.evaluation.python_executor import PythonExecutor

    out = PythonExecutor().run(prog)
    text = out[0] if isinstance(out, tuple) else out
    assert str(text).strip() == "3"  # earlier successful cell re-run silently, failed cell skipped
    assert voting.last_output_of(t).startswith("Traceback") and not voting.execution_succeeded(voting.last_output_of(t))


def test_prompt_styles_render_the_reference_templates():
    """Expected strings: the reference's literal templates (evaluation/utils.py:78-245) filled in by hand."""
    from areal_vllm_amd.evaluation.prompts import STYLES, build_prompt, style

    q = "What is 1+1?"
    boxed = "Please reason step by step, and put your final answer within \\boxed{}."
    assert style("qwen25-math-cot").render(q) == (
        f"<|im_start|>system\n{boxed}<|im_end|>\n<|im_start|>user\n{q}<|im_end|>\n<|im_start|>assistant\n")
    assert style("qwen-boxed").render(q) == (
        "<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n"
        f"<|im_start|>user\n{q}\n{boxed}<|im_end|>\n<|im_start|>assistant\n")
    assert style("deepseek-math").render(q) == f"User: {q}\n{boxed}\n\nAssistant:"
    assert style("r1-distilled-qwen").render(q) == f"<｜User｜>{q}\n{boxed}<｜Assistant｜><think>\n"
    assert style("deepscaler-fix") is style("r1-distilled-qwen")
    assert style("qwen3").render(q) == (
        f"<|im_start|>user\n{q}\n{boxed}/no_think<|im_end|>\n<|im_start|>assistant\n<think></think>")
    assert style("mistral").render(q) == f"[INST] {q}[/INST]"
    assert style("kpmath").render(q) == (
        'User: Please reason step by step and put your final answer at the end with "The answer is: ".\n\n'
        f"{q}\n\nAssistant:")
    assert style("mmiqc").render(q).startswith(
        'Please solve the following problem and put your answer at the end with "The answer is: ".\n\n')
    shots = [("2+2?", "4"), ("3+3?", "6")]
    assert style("cot").render(q, shots) == (
        "Question: 2+2?\nAnswer: 4\n\n\nQuestion: 3+3?\nAnswer: 6\n\n\nQuestion: What is 1+1?\nAnswer:")
    # chat models take the examples inside ONE user turn
    one_turn = style("qwen25-math-cot").render(q, shots)
    assert one_turn.count("<|im_start|>user") == 1 and "2+2?\n4\n\n3+3?\n6\n\n" + q in one_turn
    assert style("numina").render(q, shots[:1]) == "### Problem: 2+2?\n### Solution: 4\n\n### Problem: What is 1+1?\n### Solution:"
    assert style("platypus_fs").render(q).startswith("Below is an instruction that describes a task.")
    assert style("tora").render(q).startswith("Integrate step-by-step reasoning and Python code")
    assert len(STYLES) >= 35
    with pytest.raises(KeyError):
        style("no-such-style")
    ex = {"gsm8k": shots, "gaokao": [("q%d" % i, "A") for i in range(6)]}
    row = {"question": q, "gt_ans": "2"}
    assert build_prompt(row, "svamp", "cot", num_shots=1, examples=ex).startswith("Question: 2+2?\nAnswer: 4")
    assert build_prompt(row, "svamp", "cot", num_shots=0, examples=ex) == "Question: What is 1+1?\nAnswer:"
    # adapt_few_shot: five examples for the Chinese exam sets, but only for multiple-choice rows
    assert build_prompt(row, "gaokao2024_I", "cot", 0, ex, adapt_few_shot=True) == "Question: What is 1+1?\nAnswer:"
    mc = build_prompt({"question": q, "gt_ans": "B"}, "gaokao2024_I", "cot", 0, ex, adapt_few_shot=True)
    assert mc.count("Question:") == 6
