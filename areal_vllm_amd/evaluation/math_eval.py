"""Generate answers for a benchmark file with the native engine and score them.

    python -m areal_vllm_amd.evaluation.math_eval --model /ckpt/epoch1... --data aime24.jsonl \
        --n-sampling 8 --max-new-tokens 4096 --temperature 0.6 --output results.json

Rows of the data file: {"query_id"|"id", "prompt", "solutions": [...]} for math, or
{"query_id", "prompt", "task": "code", "input_output": ...} for code. Metrics: mean accuracy over all
samples (``mean@n``), unbiased ``pass@k`` and majority-vote accuracy for math. What the reference's
``evaluation/eval_and_aggregate.py`` / ``math_eval.py`` / ``code_eval.py`` report.
"""
import argparse
import json
import math
import os
from collections import Counter
from typing import Dict, List, Optional

import numpy as np


def pass_at_k(n: int, c: int, k: int) -> float:
    """Unbiased estimator 1 - C(n-c, k) / C(n, k) of solving a problem within k of n samples, c correct."""
    if n - c < k:
        return 1.0
    return 1.0 - float(np.prod(1.0 - k / np.arange(n - c + 1, n + 1)))


def _score(rows: List[Dict], answers: List[List[str]]) -> List[List[int]]:
    from ..functioncall.local_code_verify import code_verify
    from ..impl.environment.math_code_single_step_env import extract_code
    from ..reward.math_parser import parse_lines_in_parallel

    id2info = {str(r.get("query_id", r.get("id"))): dict(r, query_id=str(r.get("query_id", r.get("id"))))
               for r in rows}
    out = []
    for r, ans in zip(rows, answers):
        qid = str(r.get("query_id", r.get("id")))
        if r.get("task", "math") == "code":
            out.append([int(x) for x in code_verify(id2info, [extract_code(a) or "" for a in ans], [qid] * len(ans))])
        elif r.get("choices"):  # multiple choice: compare the option letter
            from .benchmarks import score_choice
            from ..reward.math_parser import last_boxed_content

            truth = last_boxed_content(r["solutions"][0]) or ""
            out.append([int(score_choice(a, truth)) for a in ans])
        else:
            out.append([int(x) for x in parse_lines_in_parallel(id2info, ans, [qid] * len(ans), max_workers=4)])
    return out


def _majority(answers: List[str], correct: List[int]) -> int:
    from ..reward.math_parser import extract_answer

    keys = [extract_answer(a) or "" for a in answers]
    top = Counter(keys).most_common(1)[0][0]
    return int(correct[keys.index(top)])


def evaluate_checkpoint(model_path: str, data_path: str, n_sampling: int = 1, max_new_tokens: int = 2048,
                        temperature: float = 0.6, top_p: float = 1.0, greedy: bool = False, seed: int = 1,
                        prompt_template: str = "{prompt}", output: Optional[str] = None, device: Optional[str] = None,
                        max_running: int = 256, kv_cache_gb: Optional[float] = None,
                        data_name: Optional[str] = None) -> Dict[str, float]:
    """`data_name`: the benchmark whose raw file format `data_path` is in (see ``benchmarks.FORMATS``); None =
    rows already in evaluator shape."""
    import torch

    from ..api.cli_args import InferenceServerConfig
    from ..inference.engine import NativeInferenceEngine
    from ..utils.hf_utils import load_hf_tokenizer

    if data_name:
        from .benchmarks import load_benchmark

        rows = load_benchmark(data_path, data_name)
    else:
        with open(data_path) as f:
            rows = [json.loads(l) for l in f if l.strip()]
    tok = load_hf_tokenizer(model_path)
    dev = torch.device(device) if device else None
    cfg = InferenceServerConfig(model_path=model_path, random_seed=seed, max_running_requests=max_running,
                                context_length=32768, disable_cuda_graph=(dev is not None and dev.type == "cpu"))
    eng = NativeInferenceEngine(cfg, device=dev, kv_cache_gb=kv_cache_gb).initialize(load_weights=True)
    try:
        prompts = [tok(prompt_template.format(prompt=r["prompt"]))["input_ids"] for r in rows]
        flat = [p for p in prompts for _ in range(n_sampling)]
        stops = [t for t in (tok.eos_token_id, tok.pad_token_id) if t is not None]
        res = eng.generate_blocking(flat, max_new_tokens=max_new_tokens, temperature=temperature, top_p=top_p,
                                    greedy=greedy, stop_token_ids=stops)
    finally:
        eng.destroy()
    answers = [[tok.decode(r.output_tokens, skip_special_tokens=True) for r in res[i * n_sampling:(i + 1) * n_sampling]]
               for i in range(len(rows))]
    correct = _score(rows, answers)
    n = n_sampling
    metrics = {f"mean@{n}": float(np.mean([np.mean(c) for c in correct])),
               "avg_gen_tokens": float(np.mean([len(r.output_tokens) for r in res])),
               "truncated_ratio": float(np.mean([r.finish_reason != "stop" for r in res]))}
    for k in sorted({1, 2, 4, 8, 16, 32} & set(range(1, n + 1))):
        metrics[f"pass@{k}"] = float(np.mean([pass_at_k(n, int(sum(c)), k) for c in correct]))
    if n > 1 and all(r.get("task", "math") == "math" for r in rows):
        metrics[f"maj@{n}"] = float(np.mean([_majority(a, c) for a, c in zip(answers, correct)]))
    if output:
        os.makedirs(os.path.dirname(os.path.abspath(output)), exist_ok=True)
        with open(output, "w") as f:
            json.dump(dict(model=model_path, data=data_path, metrics=metrics,
                           samples=[dict(id=str(r.get("query_id", r.get("id"))), answers=a, correct=c)
                                    for r, a, c in zip(rows, answers, correct)]), f)
    return metrics


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--model", required=True)
    p.add_argument("--data", required=True)
    p.add_argument("--n-sampling", type=int, default=1)
    p.add_argument("--max-new-tokens", type=int, default=2048)
    p.add_argument("--temperature", type=float, default=0.6)
    p.add_argument("--top-p", type=float, default=1.0)
    p.add_argument("--greedy", action="store_true")
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--prompt-template", default="{prompt}")
    p.add_argument("--output", default=None)
    p.add_argument("--device", default=None)
    p.add_argument("--data-name", default=None, help="raw benchmark format of --data (gsm8k, math_500, aqua, ...)")
    a = p.parse_args(argv)
    m = evaluate_checkpoint(a.model, a.data, a.n_sampling, a.max_new_tokens, a.temperature, a.top_p, a.greedy, a.seed,
                            a.prompt_template, a.output, a.device, data_name=a.data_name)
    print(json.dumps(m))


if __name__ == "__main__":
    main()
