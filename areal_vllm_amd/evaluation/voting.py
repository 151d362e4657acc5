"""Aggregations over several sampled answers per problem, and over dumped rollouts.

* ``maj_at_k``: majority-vote accuracy. Answers are grouped either by their normalised string or, with
  ``symbolic=True``, by mathematical equivalence (an answer joins the first group whose representative it
  equals under ``reward.math_parser.math_equal`` with its time-limited symbolic step); the vote is scored by
  the correctness of the winning group's first member.
* ``rm_at_k``: best-of-k under a reward model: per problem the sample with the highest reward-model score is
  the one that counts.
* ``accuracy_from_rollout_dumps``: per-prompt and mean accuracy from the ``*.jsonl`` files a training run
  dumps (one line per generated sample with ``reward`` and ``query_id`` or ``prompt``).
* ``split_trajectory`` / ``join_trajectory`` / ``program_of``: tool-integrated reasoning transcripts, i.e.
  text interleaved with ```` ```python ```` programs and ```` ```output ```` blocks.

Counterparts in the reference: ``evaluation/rm_maj_eval.py`` (group_pred, eval_maj_k_metrics,
eval_rm_k_metrics), ``evaluation/aggregate_acc_from_generated.py``, ``evaluation/trajectory.py``.
Record layout of the sample files (one JSON object per line): ``pred`` = list of extracted answers,
``score`` = list of 0/1 correctness, ``pred_score`` = list of reward-model scores (a scalar or a one-element
list per sample).
"""
import glob
import hashlib
import json
import os
import re
from typing import Any, Dict, Iterable, List, Optional, Sequence, Tuple


def _rows(path_or_rows) -> List[Dict[str, Any]]:
    if isinstance(path_or_rows, (str, os.PathLike)):
        with open(path_or_rows) as f:
            return [json.loads(line) for line in f if line.strip()]
    return list(path_or_rows)


def _equal_with_limit(a: str, b: str) -> bool:
    from ..reward.math_parser import math_equal

    try:
        return bool(math_equal(a, b, timeout=True))  # the symbolic step runs in a time-limited process
    except Exception:
        return False


def group_predictions(preds: Sequence[str], normalise: bool = True,
                      symbolic: bool = False) -> Tuple[Dict[str, List[int]], str]:
    """Partition sample indices by answer. Returns (groups keyed by representative answer, the ORIGINAL text
    of the first member of the largest group). Ties go to the group that appeared first."""
    from ..reward.math_parser import strip_string

    keys = [strip_string(p) if normalise else p for p in preds]
    groups: Dict[str, List[int]] = {}
    for i, k in enumerate(keys):
        home = k if k in groups else None
        if home is None and symbolic:
            home = next((g for g in groups if _equal_with_limit(k, g)), None)
        groups.setdefault(k if home is None else home, []).append(i)
    winner = max(groups.values(), key=len)  # max keeps the first of equally long lists
    return groups, preds[winner[0]]


def maj_at_k(samples, k: int = 8, normalise: bool = False, symbolic: bool = False) -> float:
    """Majority-vote accuracy in percent over the first k samples of every problem."""
    rows = _rows(samples)
    right = 0
    for r in rows:
        assert len(r["score"]) >= k and len(r["pred"]) >= k, f"a problem has fewer than {k} samples"
        groups, _ = group_predictions(r["pred"][:k], normalise=normalise, symbolic=symbolic)
        right += int(bool(r["score"][max(groups.values(), key=len)[0]]))
    return 100.0 * right / max(len(rows), 1)


def rm_at_k(samples, k: int = 8) -> float:
    """Best-of-k accuracy in percent: the sample a reward model scores highest is the answer."""
    rows = _rows(samples)
    right = 0
    for r in rows:
        assert len(r["pred_score"]) >= k and len(r["score"]) >= k, f"a problem has fewer than {k} samples"
        rm = [float(s[0]) if isinstance(s, (list, tuple)) else float(s) for s in r["pred_score"][:k]]
        right += int(bool(r["score"][rm.index(max(rm))]))
    return 100.0 * right / max(len(rows), 1)


def accuracy_from_rollout_dumps(log_dir: str) -> Dict[str, Any]:
    """Every ``*.jsonl`` under `log_dir`: samples of one prompt are pooled under its ``query_id`` (the part
    before ``@``) or, without one, under a digest of the prompt text; a sample is right when its reward is
    positive. Returns {"per_query": {id: (n, accuracy)}, "mean_accuracy": mean over prompts}."""
    pooled: Dict[str, List[bool]] = {}
    for path in sorted(glob.glob(os.path.join(log_dir, "*.jsonl"))):
        for x in _rows(path):
            qid = str(x["query_id"]).split("@")[0] if "query_id" in x else hashlib.md5(
                x["prompt"].encode("utf-8")).hexdigest()
            pooled.setdefault(qid, []).append(x["reward"] > 0)
    per = {q: (len(v), sum(v) / len(v)) for q, v in sorted(pooled.items())}
    return {"per_query": per, "mean_accuracy": (sum(a for _, a in per.values()) / len(per)) if per else 0.0}


# ------------------------------------------------------------------ tool-integrated transcripts
_FENCE = re.compile(r"```(python|output)[ \t]*\n(.*?)```", re.S)


def split_trajectory(text: str) -> List[Dict[str, str]]:
    """A transcript as a list of {"role": "text" | "program" | "output", "content": str} pieces in order.
    Text before an unterminated fence stays text; an empty piece is dropped."""
    out: List[Dict[str, str]] = []
    at = 0
    for m in _FENCE.finditer(text):
        if text[at:m.start()].strip():
            out.append({"role": "text", "content": text[at:m.start()].strip()})
        out.append({"role": "program" if m.group(1) == "python" else "output", "content": m.group(2).strip()})
        at = m.end()
    if text[at:].strip():
        out.append({"role": "text", "content": text[at:].strip()})
    return out


def join_trajectory(pieces: Iterable[Dict[str, str]]) -> str:
    tag = {"program": "python", "output": "output"}
    return "\n".join(p["content"] if p["role"] == "text" else f"```{tag[p['role']]}\n{p['content']}\n```"
                     for p in pieces) + "\n"


def execution_succeeded(output: Optional[str]) -> bool:
    """False for an output block that carries a Python error report."""
    if output is None:
        return False
    tail = output.strip().splitlines()[-1:] or [""]
    return not ("Traceback (most recent call last)" in output or re.match(r"^\w*(Error|Exception)\b", tail[0]))


def program_of(text: str, last_only: bool = False) -> str:
    """The program a transcript asks to run next. With `last_only` just its final program block; otherwise
    the final block preceded by every earlier block whose recorded output shows no error (state built by
    earlier successful cells is needed again, failed cells are left out), with the ``print`` calls of the
    earlier blocks neutralised so that only the last block's output is observed."""
    pieces = split_trajectory(text)
    progs = [i for i, p in enumerate(pieces) if p["role"] == "program"]
    if not progs:
        return ""
    if last_only:
        return pieces[progs[-1]]["content"]
    kept: List[str] = []
    for i in progs[:-1]:
        nxt = pieces[i + 1] if i + 1 < len(pieces) else None
        if nxt is not None and nxt["role"] == "output" and execution_succeeded(nxt["content"]):
            kept.append(re.sub(r"(?m)^(\s*)print\(", r"\1(lambda *a, **k: None)(", pieces[i]["content"]))
    return "\n".join(kept + [pieces[progs[-1]]["content"]])


def last_output_of(text: str) -> str:
    outs = [p["content"] for p in split_trajectory(text) if p["role"] == "output"]
    return outs[-1] if outs else ""


def main(argv=None):
    import argparse

    ap = argparse.ArgumentParser(description="maj@k / rm@k of a sample file, or accuracy of a rollout dump directory")
    ap.add_argument("--samples", default=None, help="jsonl with pred / score (/ pred_score) lists per problem")
    ap.add_argument("--log-path", default=None, help="directory of dumped rollouts (*.jsonl)")
    ap.add_argument("-k", type=int, default=8)
    ap.add_argument("--symbolic", action="store_true")
    a = ap.parse_args(argv)
    res: Dict[str, Any] = {}
    if a.samples:
        rows = _rows(a.samples)
        res[f"maj@{a.k}"] = maj_at_k(rows, a.k, symbolic=a.symbolic)
        if all("pred_score" in r for r in rows):
            res[f"rm@{a.k}"] = rm_at_k(rows, a.k)
    if a.log_path:
        agg = accuracy_from_rollout_dumps(a.log_path)
        res["n_queries"], res["mean_accuracy"] = len(agg["per_query"]), agg["mean_accuracy"]
    print(json.dumps(res))
    return res


if __name__ == "__main__":
    main()
