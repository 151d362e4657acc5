"""Evaluate one checkpoint on several benchmarks and print / store one table.

    python -m areal_vllm_amd.evaluation.eval_and_aggregate --model-path /ckpt --data-dir ./data \
        --data-names aime24,amc23,math_500 --n-sampling 32 --max-gen-tokens 32768 --prompt-type r1-distilled-qwen

Per benchmark (``<data-dir>/<name>/test.jsonl``): a greedy pass (accuracy, generation length) and a sampled
pass (``sample_pass@1``, ``pass@8/16/32``, ``maj@32`` as far as ``--n-sampling`` allows, generation length);
for Codeforces problem sets (``--cf-*`` files given and the benchmark's name starting with ``codeforces``) the
Elo estimate of ``cf_elo.py``. Results are cached in
``<output>/math_eval_<max_gen_tokens>/aggregate_<prompt_type>_<t>_<top_p>.json`` and only recomputed with
``--overwrite``. The reference's ``evaluation/eval_and_aggregate.py`` drives shell scripts that start one
vLLM process per seed; here both passes run in this process on the native engine.
"""
import argparse
import json
import os
from typing import Dict, List, Optional

from tabulate import tabulate

from . import cf_elo
from .math_eval import evaluate_checkpoint

# Prompt templates by model family ({prompt} = the benchmark's question).
PROMPT_TYPES: Dict[str, str] = {
    "plain": "{prompt}",
    "r1-distilled-qwen": "<｜User｜>{prompt}\nPlease reason step by step, and put your final answer within \\boxed{{}}."
                         "<｜Assistant｜><think>\n",
    "qwen-boxed": "<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n<|im_start|>user\n{prompt}\nPlease reason "
                  "step by step, and put your final answer within \\boxed{{}}.<|im_end|>\n<|im_start|>assistant\n",
    "qwen3-think": "<|im_start|>user\n{prompt}\nPlease reason step by step, and put your final answer within "
                   "\\boxed{{}}.<|im_end|>\n<|im_start|>assistant\n<think>\n",
    "code-r1": "<｜User｜>{prompt}\nWrite a python program that reads from standard input and writes to standard output; "
               "put it in one ```python block.<｜Assistant｜><think>\n",
}


def evaluate_benchmark(model_path: str, data_file: str, out_dir: str, n_sampling: int, max_gen_tokens: int,
                       temperature: float, top_p: float, template: str, device: Optional[str] = None,
                       cf_files: Optional[Dict[str, str]] = None, data_name: Optional[str] = None) -> Dict[str, float]:
    os.makedirs(out_dir, exist_ok=True)
    greedy = evaluate_checkpoint(model_path, data_file, n_sampling=1, max_new_tokens=max_gen_tokens, greedy=True,
                                 prompt_template=template, output=os.path.join(out_dir, "greedy.json"), device=device,
                                 data_name=data_name)
    out = dict(greedy_acc=100.0 * greedy["mean@1"], greedy_length=greedy["avg_gen_tokens"])
    if n_sampling > 1:
        sampled_path = os.path.join(out_dir, f"sampled_n{n_sampling}.json")
        s = evaluate_checkpoint(model_path, data_file, n_sampling=n_sampling, max_new_tokens=max_gen_tokens,
                                temperature=temperature, top_p=top_p, prompt_template=template, output=sampled_path,
                                device=device, data_name=data_name)
        out["sample_length"] = s["avg_gen_tokens"]
        out["sample_pass@1"] = 100.0 * s[f"mean@{n_sampling}"]
        for k in (8, 16, 32):
            if f"pass@{k}" in s:
                out[f"pass@{k}"] = 100.0 * s[f"pass@{k}"]
        if f"maj@{n_sampling}" in s:
            out[f"maj@{n_sampling}"] = 100.0 * s[f"maj@{n_sampling}"]
        if cf_files:
            with open(sampled_path) as f:
                samples = json.load(f)["samples"]
            elo = cf_elo.estimate_elo_from_files([dict(idx=i, score=x["correct"]) for i, x in enumerate(samples)],
                                                 cf_files["metadata"], cf_files["ratings"], cf_files["contests"],
                                                 pass_n=min(8, n_sampling))
            if elo is not None:
                out["cf_rating"] = elo["estimated_rating"]
                out["cf_percentile"] = elo["estimated_percentile"]
    return out


def format_table(results: Dict[str, Dict[str, float]]) -> str:
    cols: List[str] = []
    for r in results.values():
        cols += [k for k in r if k not in cols]
    rows = [[name] + [("-" if r.get(c) is None else f"{r[c]:.0f}" if c == "cf_rating" else f"{r[c]:.1f}") for c in cols]
            for name, r in results.items()]
    return tabulate(rows, headers=["dataset"] + cols)


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--model-path", required=True)
    p.add_argument("--data-dir", default="./data")
    p.add_argument("--data-names", required=True, type=lambda s: [x for x in s.split(",") if x])
    p.add_argument("--output-path", default=None)
    p.add_argument("--n-sampling", type=int, default=32)
    p.add_argument("--max-gen-tokens", type=int, default=32768)
    p.add_argument("--temperature", type=float, default=0.6)
    p.add_argument("--top-p", type=float, default=0.95)
    p.add_argument("--prompt-type", default="r1-distilled-qwen", choices=sorted(PROMPT_TYPES))
    p.add_argument("--overwrite", action="store_true")
    p.add_argument("--raw-format", action="store_true",
                   help="the test.jsonl files are the benchmarks' published files (fields per data set, see "
                        "evaluation/benchmarks.py), not rows with prompt / solutions")
    p.add_argument("--device", default=None)
    p.add_argument("--cf-cache-file", default=None)
    p.add_argument("--cf-metadata-path", default=None)
    p.add_argument("--cf-ratings-path", default=None)
    a = p.parse_args(argv)
    base = os.path.join(a.output_path or a.model_path, f"math_eval_{a.max_gen_tokens}")
    os.makedirs(base, exist_ok=True)
    result_path = os.path.join(base, f"aggregate_{a.prompt_type}_{a.temperature:.1f}_{a.top_p:.2f}.json")
    if os.path.exists(result_path) and not a.overwrite:
        with open(result_path) as f:
            results = json.load(f)
    else:
        cf = None
        if a.cf_cache_file and a.cf_metadata_path and a.cf_ratings_path:
            cf = dict(contests=a.cf_cache_file, metadata=a.cf_metadata_path, ratings=a.cf_ratings_path)
        results = {}
        for name in a.data_names:
            results[name] = evaluate_benchmark(
                a.model_path, os.path.join(a.data_dir, name, "test.jsonl"), os.path.join(base, name), a.n_sampling,
                a.max_gen_tokens, a.temperature, a.top_p, PROMPT_TYPES[a.prompt_type], a.device,
                cf if name.startswith("codeforces") else None,
                data_name=name if a.raw_format else None)
        with open(result_path, "w") as f:
            json.dump(results, f, indent=2)
    print(format_table(results))
    return results


if __name__ == "__main__":
    main()
