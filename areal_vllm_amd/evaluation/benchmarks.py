"""Benchmark file formats -> the one row shape the evaluator scores.

Public math / reasoning benchmarks ship their test sets with different field names, answer conventions and, for
multiple-choice sets, different ways of listing the options. `normalise_row(raw, data_name)` turns one raw record
into ``{"query_id", "prompt", "solutions": [boxed ground truth], "task": "math", "choices": bool}``, the shape
``math_eval.evaluate_checkpoint`` and the reward parser work on; `load_benchmark` does it for a ``test.jsonl``.
Covers the data sets the reference evaluates (``evaluation/parser.py:574-769`` parse_ground_truth /
parse_question, ``evaluation/data_loader.py``); everything is table-driven: one `Format` per family of data sets.
"""
import dataclasses
import json
import os
import re
from typing import Any, Callable, Dict, List, Optional, Sequence

from ..reward.math_parser import extract_answer, strip_string


def _first(raw: Dict[str, Any], keys: Sequence[str], default: str = "") -> Any:
    for k in keys:
        if k in raw and raw[k] is not None:
            return raw[k]
    return default


def _lettered(options: Sequence[Any], labels: str = "ABCDEFGH") -> str:
    return " ".join(f"({l}) {str(o).strip()}" for l, o in zip(labels, options))


@dataclasses.dataclass
class Format:
    """How one family of data sets spells its question and its ground truth."""
    question: Callable[[Dict[str, Any]], str]
    answer: Callable[[Dict[str, Any]], Any]
    strip: bool = True         # normalise the ground truth string (units, \\text, spacing, fractions)
    choices: bool = False      # the answer is an option letter
    skip_unit: bool = False


def _plain_question(raw):
    return str(_first(raw, ("question", "problem", "Question", "input", "prompt", "content")))


def _tabmwp_question(raw):
    title = f'regarding "{raw["table_title"]}" ' if raw.get("table_title") else ""
    q = f'Read the following table {title}and answer a question:\n{raw["table"]}\n{raw["question"]}'
    if raw.get("choices"):
        q += f' Please select from the following options: {raw["choices"]}'
    return q


def _tabmwp_answer(raw):
    a = str(raw["answer"])
    if raw.get("ans_type") in ("integer_number", "decimal_number"):
        if "/" in a:
            n, d = a.split("/")[:2]
            return str(int(n) / int(d))
        if "%" in a:
            return str(float(a.split("%")[0]) / 100)
        return str(float(a.replace(",", "")))
    return a


def _sat_question(raw):
    # options come as "A) 3 B) 4 C) 5 D) 6": bracket every letter
    opts = re.sub(r"(^|\s)([A-E])\)\s", lambda m: f"{m.group(1)}({m.group(2)}) ", str(raw["options"]).strip())
    return f'{str(raw["question"]).strip()}\nAnswer Choices: {opts}'


def _aqua_question(raw):
    # options come as ["A)21", "B)21.5", ...]
    opts = " ".join("(" + o.strip().replace(")", ") ", 1).strip() for o in raw["options"])
    return f'{str(raw["question"]).strip()}\nAnswer Choices: {opts}'


def _svamp_question(raw):
    body = str(raw["Body"]).strip()
    return f'{body if body.endswith(".") else body + "."} {str(raw["Question"]).strip()}'


def _gaokao_mix_answer(raw):
    return raw["choice_answer"] if raw.get("choice_answer") else raw["answer"]


_BOXED = Format(_plain_question, lambda r: extract_answer(str(r["solution"]), "math"), strip=False)
_ANSWER_FIELD = Format(_plain_question, lambda r: r["answer"])

FORMATS: Dict[str, Format] = {
    "math": _BOXED, "minerva_math": _BOXED, "math_500": _BOXED,
    "gsm8k": Format(_plain_question, lambda r: str(r["answer"]).split("####")[-1]),
    "svamp": Format(_svamp_question, lambda r: r["Answer"]),
    "asdiv": Format(lambda r: f'{str(r["body"]).strip()} {str(r["question"]).strip()}',
                    lambda r: re.sub(r"\(.*?\)", "", str(r["answer"]))),
    "mawps": Format(_plain_question, lambda r: r["target"]),
    "tabmwp": Format(_tabmwp_question, _tabmwp_answer),
    "carp_en": Format(lambda r: str(r["content"]), lambda r: r["answer"], skip_unit=True),
    "mmlu_stem": Format(lambda r: f'{str(r["question"]).strip()}\nAnswer Choices: {_lettered(r["choices"])}',
                        lambda r: "ABCD"[int(r["answer"])], strip=False, choices=True),
    "sat_math": Format(_sat_question, lambda r: r["Answer"], strip=False, choices=True),
    "aqua": Format(_aqua_question, lambda r: r["correct"], strip=False, choices=True),
    "gaokao_math_qa": Format(lambda r: f'{str(r["question"]).strip()}\n选项: '
                                       + " ".join(f"({k}) {v}" for k, v in r["options"].items()),
                             lambda r: r["label"], strip=False, choices=True),
    "gaokao2023en": Format(_plain_question, lambda r: str(r["answer"]).replace("$", "").strip()),
    "college_math": Format(_plain_question, lambda r: str(r["answer"]).replace("$", "").strip()),
    "gaokao_math_cloze": Format(_plain_question, lambda r: str(r["answer"]).replace("$", "").strip()),
    "gaokao2024_mix": Format(_plain_question, _gaokao_mix_answer),
    "cn_middle_school": Format(_plain_question, _gaokao_mix_answer),
    "olympiadbench": Format(_plain_question, lambda r: str(r["final_answer"][0]).strip("$")),
    "gpqa_diamond": Format(_plain_question, lambda r: r["answer"], strip=False, choices=True),
}
for _name in ("aime24", "aime25", "amc23", "cmath", "gaokao2024_I", "gaokao2024_II", "imo2024"):
    FORMATS[_name] = _ANSWER_FIELD


def format_for(data_name: str) -> Format:
    if data_name in FORMATS:
        return FORMATS[data_name]
    if data_name.startswith("train"):  # training splits re-used as held-out sets: {"prompt", "final_answer"}
        return Format(lambda r: str(r["prompt"]), lambda r: r["final_answer"])
    raise KeyError(f"unknown benchmark format {data_name!r}; known: {sorted(FORMATS)}")


def ground_truth(raw: Dict[str, Any], data_name: str) -> str:
    """The reference answer of one raw record as a normalised string."""
    fmt = format_for(data_name)
    if "gt" in raw and "gt_cot" in raw:  # files that were pre-processed once carry both
        gt = extract_answer(str(raw["gt_cot"]), data_name) if data_name == "math" else raw["gt"]
    else:
        gt = fmt.answer(raw)
    gt = str(gt).strip()
    if fmt.strip:
        return strip_string(gt, skip_unit=fmt.skip_unit)
    return gt.replace("\\neq", "\\ne").replace("\\leq", "\\le").replace("\\geq", "\\ge")


def question_text(raw: Dict[str, Any], data_name: str) -> str:
    q = format_for(data_name).question(raw).strip()
    gt = ground_truth(raw, data_name).lower()
    if gt in ("true", "false"):
        q += " (True or False)"
    elif gt in ("yes", "no"):
        q += " (Yes or No)"
    return q


def normalise_row(raw: Dict[str, Any], data_name: str, index: int = 0) -> Dict[str, Any]:
    if "prompt" in raw and "solutions" in raw and data_name not in FORMATS:
        return dict(raw, query_id=str(_first(raw, ("query_id", "id", "idx"), index)))  # already in evaluator shape
    fmt = format_for(data_name)
    gt = ground_truth(raw, data_name)
    return dict(query_id=str(_first(raw, ("query_id", "id", "idx", "unique_id"), index)), task="math",
                prompt=question_text(raw, data_name), solutions=[f"\\boxed{{{gt}}}"], choices=fmt.choices,
                data_name=data_name)


def load_benchmark(data_dir: str, data_name: str, split: str = "test", limit: Optional[int] = None,
                   start: int = 0) -> List[Dict[str, Any]]:
    """``<data_dir>/<data_name>/<split>.jsonl`` (or a path to one file) as evaluator rows, deduplicated by id."""
    path = data_dir if os.path.isfile(data_dir) else os.path.join(data_dir, data_name, f"{split}.jsonl")
    rows, seen = [], set()
    with open(path) as f:
        for i, line in enumerate(l for l in f if l.strip()):
            r = normalise_row(json.loads(line), data_name, i)
            if r["query_id"] in seen:
                continue
            seen.add(r["query_id"])
            rows.append(r)
    return rows[start:None if limit is None else start + limit]


# ------------------------------------------------------------------------------- few-shot prompting
def few_shot_prompt(question: str, shots: Sequence[Sequence[str]], instruction: str = "",
                    q_prefix: str = "Question: ", a_prefix: str = "Answer: ", sep: str = "\n\n") -> str:
    """Classic k-shot prompt for base models: instruction, k (question, worked answer) pairs, the question."""
    parts = [instruction.strip()] if instruction.strip() else []
    parts += [f"{q_prefix}{q.strip()}\n{a_prefix}{a.strip()}" for q, a in shots]
    parts.append(f"{q_prefix}{question.strip()}\n{a_prefix}".rstrip())
    return sep.join(parts)


def score_choice(prediction: str, truth: str) -> bool:
    """Multiple-choice scoring: the LAST option letter the model commits to (boxed or in running text)."""
    from ..reward.math_parser import choice_answer_clean, last_boxed_content

    boxed = last_boxed_content(prediction)
    letter = choice_answer_clean(boxed if boxed is not None else prediction)
    return letter.strip().upper() == str(truth).strip().strip("()").upper()
