"""Math answer extraction and equivalence checking for verifiable rewards.

Public surface of the reference's verifier (``areal/reward/math_parser.py:360-846``):
``extract_answer`` (last ``\\boxed{}`` / "the answer is" / last number), ``strip_string`` normalisation,
``math_equal`` (numeric with relative tolerance and percent forms, element-wise tuples / intervals /
matrices, equations, symbolic equality through sympy), ``process_results(answer, solution) -> (0|1,
(pred, ref))`` and ``parse_lines_in_parallel`` over a process pool with per-job timeouts.

The LaTeX handling is a self-contained translator to sympy's expression parser (latex2sympy is not a
dependency): fractions, roots, powers, \\pi, \\cdot/\\times, \\left/\\right, degrees, percent, mixed
text wrappers. Anything the translator cannot parse falls back to normalised string comparison.
"""
import json
import multiprocessing as mp
import re
from concurrent.futures import ProcessPoolExecutor
from concurrent.futures import TimeoutError as FutTimeout
from math import isclose
from typing import Dict, List, Optional, Tuple

from ..utils import logging

logger = logging.getLogger("math-parser")

_UNITS = ("degrees", "degree", "cm", "centimeters", "meters", "meter", "mph", "inches", "inch", "feet", "foot",
          "units", "unit", "square", "sq", "dollars", "dollar", "cents", "cent", "hours", "hour", "minutes",
          "minute", "seconds", "second", "days", "day", "km", "kg", "grams", "gram", "miles", "mile", "years", "year",
          "points", "point", "ways", "people", "students", "percent")
_WORD_NUM = {"zero": 0, "one": 1, "two": 2, "three": 3, "four": 4, "five": 5, "six": 6, "seven": 7, "eight": 8,
             "nine": 9, "ten": 10, "eleven": 11, "twelve": 12, "thirteen": 13, "fourteen": 14, "fifteen": 15,
             "sixteen": 16, "seventeen": 17, "eighteen": 18, "nineteen": 19, "twenty": 20}


# --------------------------------------------------------------------------------- extraction
def last_boxed_content(text: str) -> Optional[str]:
    """Content of the last \\boxed{...} / \\fbox{...} with balanced braces, or of `\\boxed x`."""
    idx = max(text.rfind("\\boxed"), text.rfind("\\fbox"))
    if idx < 0:
        return None
    i = text.find("{", idx)
    head = text[idx:i] if i >= 0 else ""
    if i < 0 or head.strip() not in ("\\boxed", "\\fbox"):
        m = re.match(r"\\boxed\s+([^\s$]+)", text[idx:])
        return m.group(1) if m else None
    depth, j = 0, i
    while j < len(text):
        if text[j] == "{":
            depth += 1
        elif text[j] == "}":
            depth -= 1
            if depth == 0:
                return text[i + 1:j]
        j += 1
    return None


_NUM_RE = re.compile(r"-?\d[\d,]*\.?\d*(?:[eE][-+]?\d+)?|-?\.\d+")


def extract_answer(pred_str: str, data_name: str = "math", use_last_number: bool = True) -> str:
    pred_str = pred_str.replace("\u043a\u0438", "")
    pred = None
    boxed = last_boxed_content(pred_str)
    if boxed is not None:
        pred = boxed
    elif "final answer is $" in pred_str and "$. I hope" in pred_str:
        pred = pred_str.split("final answer is $", 1)[1].split("$. I hope", 1)[0]
    elif "he answer is" in pred_str:
        pred = pred_str.split("he answer is")[-1].strip()
    elif "final answer is" in pred_str:
        pred = pred_str.split("final answer is")[-1].strip()
    elif "####" in pred_str:
        pred = pred_str.split("####")[-1].strip()
    elif use_last_number:
        nums = _NUM_RE.findall(pred_str.replace(",", ""))
        pred = nums[-1] if nums else ""
    else:
        pred = ""
    pred = re.sub(r"\n\s*", "", pred).strip()
    pred = pred.lstrip(":").strip()
    pred = pred.rstrip(".").rstrip("/").strip()
    return strip_string(pred)


# ------------------------------------------------------------------------------ normalisation
def _fix_fracs(s: str) -> str:
    # \frac12 -> \frac{1}{2}, \frac1{72} -> \frac{1}{72}
    def rep(m):
        a, b = m.group(1), m.group(2)
        return f"\\frac{{{a}}}{{{b}}}"

    s = re.sub(r"\\frac(\d)(\d)", rep, s)
    s = re.sub(r"\\frac(\d)\{([^{}]*)\}", rep, s)
    s = re.sub(r"\\frac\{([^{}]*)\}(\d)", rep, s)
    return s


def _fix_sqrt(s: str) -> str:
    return re.sub(r"\\sqrt(\w)", r"\\sqrt{\1}", s)


def strip_string(string: str, skip_unit: bool = False) -> str:
    s = str(string).strip().replace("\n", "")
    s = s.rstrip(".")
    s = s.replace("\\!", "").replace("\\,", "").replace("\\;", "").replace("\\:", "")
    s = re.sub(r"(?<!\\)\\ ", " ", s)
    s = s.replace("tfrac", "frac").replace("dfrac", "frac")
    s = s.replace("\\left", "").replace("\\right", "")
    s = s.replace("\\neq", "\\ne").replace("\\leq", "\\le").replace("\\geq", "\\ge")
    s = re.sub(r"\\begin\{array\}\{.*?\}", r"\\begin{pmatrix}", s).replace("\\end{array}", "\\end{pmatrix}")
    s = s.replace("bmatrix", "pmatrix")
    s = re.sub(r"\\text\{\s*([^{}]*)\}", r"\1", s)
    s = re.sub(r"\\mbox\{\s*([^{}]*)\}", r"\1", s)
    s = re.sub(r"\\(?:mathrm|mathbf|textbf|mathit)\{([^{}]*)\}", r"\1", s)
    if not skip_unit:
        for _ in range(2):
            for u in _UNITS:
                s2 = re.sub(r"(^|[\d\s})])\s*" + u + r"(es|s)?(\^\{?2\}?|\^\{?3\}?)?\s*$", r"\1", s)
                if s2 != s and s2.strip():
                    s = s2.strip()
    s = s.replace("^{\\circ}", "").replace("^\\circ", "").replace("°", "")
    s = s.replace("\\$", "").replace("$", "")
    s = s.replace("\\(", "").replace("\\)", "")
    low = s.lower().strip()
    if low in _WORD_NUM:
        s = str(_WORD_NUM[low])
    s = s.replace("\\%", "%").replace("\\emptyset", "{}")
    s = s.replace("infinity", "\\infty")
    if "\\infty" not in s:
        s = s.replace("inf", "\\infty")
    s = s.replace("and", "").replace("\\mathbf", "")
    s = re.sub(r"(\d),(\d\d\d)(?!\d)", r"\1\2", s)
    s = re.sub(r"(\d),(\d\d\d)(?!\d)", r"\1\2", s)
    # x = 5 -> 5 for single short variable prefixes
    if s.count("=") == 1 and len(s.split("=")[0].strip()) <= 2 and re.fullmatch(r"[A-Za-z]\w?\s*", s.split("=")[0]):
        s = s.split("=")[1]
    s = _fix_sqrt(s)
    s = _fix_fracs(s)
    s = re.sub(r"\s+", "", s)
    if s.startswith("."):
        s = "0" + s
    s = re.sub(r"(\d+)\.0+$", r"\1", s)
    s = re.sub(r"(\d+\.\d*?[1-9])0+$", r"\1", s)
    if re.fullmatch(r"-?\d+/\d+", s):
        a, b = s.split("/")
        s = f"\\frac{{{a}}}{{{b}}}" if not a.startswith("-") else f"-\\frac{{{a[1:]}}}{{{b}}}"
    if s == "0.5":
        s = "\\frac{1}{2}"
    return s


# ----------------------------------------------------------------------------- numeric layer
def parse_digits(num) -> Optional[float]:
    s = str(num).replace(",", "").strip()
    try:
        return float(s)
    except ValueError:
        pass
    if s.endswith("%"):
        try:
            return float(s[:-1].rstrip("\\")) / 100.0
        except ValueError:
            return None
    return None


def is_digit(num) -> bool:
    return parse_digits(num) is not None


def numeric_equal(prediction: float, reference: float) -> bool:
    return isclose(reference, prediction, rel_tol=1e-4, abs_tol=1e-9)


def choice_answer_clean(pred: str) -> str:
    pred = pred.strip("\n").rstrip(".").rstrip("/").strip(" ").lstrip(":")
    found = re.findall(r"\b(A|B|C|D|E)\b", pred.upper())
    return (found[-1] if found else pred.strip().strip(".")).rstrip(".").rstrip("/")


# ---------------------------------------------------------------------------- symbolic layer
def _braced(s: str, i: int) -> Tuple[str, int]:
    """`s[i]` is '{': return (content, index after the matching '}')."""
    depth = 0
    for j in range(i, len(s)):
        if s[j] == "{":
            depth += 1
        elif s[j] == "}":
            depth -= 1
            if depth == 0:
                return s[i + 1:j], j + 1
    raise ValueError("unbalanced braces")


def latex_to_expr_text(s: str) -> str:
    """Translate a (normalised) LaTeX expression into text for sympy's `parse_expr`."""
    out = []
    i = 0
    while i < len(s):
        if s.startswith("\\frac", i):
            i += 5
            a, i = _braced(s, i)
            b, i = _braced(s, i)
            out.append(f"(({latex_to_expr_text(a)})/({latex_to_expr_text(b)}))")
        elif s.startswith("\\sqrt", i):
            i += 5
            n = None
            if i < len(s) and s[i] == "[":
                j = s.index("]", i)
                n, i = s[i + 1:j], j + 1
            a, i = _braced(s, i)
            out.append(f"(({latex_to_expr_text(a)})**(1/({latex_to_expr_text(n)})))" if n else
                       f"sqrt({latex_to_expr_text(a)})")
        elif s[i] == "^":
            i += 1
            if i < len(s) and s[i] == "{":
                a, i = _braced(s, i)
            else:
                a, i = s[i], i + 1
            out.append(f"**({latex_to_expr_text(a)})")
        elif s[i] == "{":
            a, i = _braced(s, i)
            out.append(f"({latex_to_expr_text(a)})")
        elif s[i] == "\\":
            m = re.match(r"\\([A-Za-z]+)", s[i:])
            if not m:
                i += 1
                continue
            name = m.group(1)
            i += len(m.group(0))
            table = {"pi": " pi ", "cdot": "*", "times": "*", "div": "/", "infty": " oo ", "le": "<=", "ge": ">=",
                     "ne": "!=", "ln": " log", "log": " log", "sin": " sin", "cos": " cos", "tan": " tan",
                     "pm": "+", "mp": "-"}
            out.append(table.get(name, f" {name} "))
        elif s[i] == "%":
            out.append("/100")
            i += 1
        else:
            out.append(s[i])
            i += 1
    return "".join(out)


def _to_sympy(text: str):
    from sympy.parsing.sympy_parser import (
        convert_xor,
        implicit_multiplication_application,
        parse_expr,
        standard_transformations,
    )

    tf = standard_transformations + (implicit_multiplication_application, convert_xor)
    if "\\" in text or "^" in text or "_" in text:
        # anything with LaTeX markup goes through the LaTeX grammar first (sums, integrals, matrices, \log_b,
        # \sin^2 x ... have no reading as plain expression text); calculus objects are evaluated so that
        # "\int_0^1 x dx" and "1/2" compare equal
        from ..evaluation.latex2sympy import try_latex2sympy

        e = try_latex2sympy(text)
        if e is not None and not isinstance(e, tuple):
            import sympy

            try:
                calculus = isinstance(e, sympy.Basic) and e.has(sympy.Integral, sympy.Sum, sympy.Product,
                                                                sympy.Limit, sympy.Derivative)
                return e.doit() if calculus else e
            except Exception:
                return e
    try:
        return parse_expr(latex_to_expr_text(text), transformations=tf, evaluate=True)
    except Exception:
        return parse_expr(text.replace("\\\\", "\\"), transformations=tf, evaluate=True)


def symbolic_equal(a: str, b: str) -> bool:
    import sympy

    try:
        ea, eb = _to_sympy(a), _to_sympy(b)
    except Exception:
        return False
    try:
        if str(ea) == str(eb) or ea == eb:
            return True
    except Exception:
        pass
    try:
        if isinstance(ea, sympy.Basic) and isinstance(eb, sympy.Basic):
            if ea.equals(eb) or sympy.simplify(ea - eb) == 0:
                return True
    except Exception:
        pass
    try:
        if isinstance(ea, sympy.Eq) and isinstance(eb, sympy.Eq):
            if abs(sympy.simplify((ea.lhs - ea.rhs) - (eb.lhs - eb.rhs))) == 0:
                return True
    except Exception:
        pass
    try:
        if numeric_equal(float(sympy.N(ea)), float(sympy.N(eb))):
            return True
    except Exception:
        pass
    return False


def _symbolic_worker(a, b, q):
    q.put(symbolic_equal(a, b))


def call_with_timeout(func, *args, timeout: float = 3, **kwargs):
    q = mp.Queue()
    p = mp.Process(target=func, args=args + (q,), kwargs=kwargs)
    p.start()
    p.join(timeout)
    if p.is_alive():
        p.terminate()
        p.join()
        return False
    return q.get() if not q.empty() else False


def _split_top_level(s: str, sep: str = ",") -> List[str]:
    parts, depth, cur = [], 0, []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == sep and depth == 0:
            parts.append("".join(cur))
            cur = []
        else:
            cur.append(ch)
    parts.append("".join(cur))
    return [p.strip() for p in parts]


def _matrix_rows(s: str) -> Optional[List[List[str]]]:
    m = re.fullmatch(r"\\begin\{pmatrix\}(.*)\\end\{pmatrix\}", s)
    if not m:
        return None
    return [[c.strip() for c in row.split("&")] for row in m.group(1).split("\\\\") if row.strip()]


def math_equal(prediction, reference, include_percentage: bool = True, is_close: bool = True,
               timeout: bool = False, depth: int = 0) -> bool:
    """Exact, numeric, structural or symbolic equality of two answer strings."""
    if prediction is None or reference is None:
        return False
    p, r = str(prediction).strip(), str(reference).strip()
    if p.lower() == r.lower():
        return True
    if r in ("A", "B", "C", "D", "E") and choice_answer_clean(p) == r:
        return True
    # numeric
    if is_digit(p) and is_digit(r):
        pv, rv = parse_digits(p), parse_digits(r)
        cands = [rv / 100.0, rv, rv * 100.0] if include_percentage else [rv]
        return any(numeric_equal(pv, c) if is_close else pv == c for c in cands)
    if not p or not r:
        return False
    if depth > 4:
        return False
    # matrices
    mp_, mr_ = _matrix_rows(p), _matrix_rows(r)
    if mp_ is not None and mr_ is not None:
        return (len(mp_) == len(mr_) and all(len(a) == len(b) for a, b in zip(mp_, mr_))
                and all(math_equal(x, y, include_percentage, is_close, timeout, depth + 1)
                        for a, b in zip(mp_, mr_) for x, y in zip(a, b)))
    # brackets: drop a matching outer pair on one side only if the other has none
    pb, rb = p, r
    if re.fullmatch(r"[\[(].*[\])]", p) and not re.fullmatch(r"[\[(].*[\])]", r) and "," not in p:
        pb = p[1:-1]
    if re.fullmatch(r"[\[(].*[\])]", r) and not re.fullmatch(r"[\[(].*[\])]", p) and "," not in r:
        rb = r[1:-1]
    if pb.lower() == rb.lower():
        return True
    # tuples / intervals / lists: same delimiters, element-wise
    if (re.fullmatch(r"[\[(].+[\])]", p) and re.fullmatch(r"[\[(].+[\])]", r)):
        pp, rr = _split_top_level(p[1:-1]), _split_top_level(r[1:-1])
        if len(pp) == len(rr) and len(pp) > 1:
            if p[0] == r[0] and p[-1] == r[-1] and all(
                    math_equal(x, y, include_percentage, is_close, timeout, depth + 1) for x, y in zip(pp, rr)):
                return True
            return False
    if "," in p and "," in r and not re.search(r"[\[(]", p + r):
        pp, rr = _split_top_level(p), _split_top_level(r)
        if len(pp) == len(rr) and all(
                any(math_equal(x, y, include_percentage, is_close, timeout, depth + 1) for y in rr) for x in pp):
            return True
    # equations
    if p.count("=") == 1 and r.count("=") == 1:
        pl, pr = p.split("=")
        rl, rr_ = r.split("=")
        if symbolic_equal(f"({pl})-({pr})", f"({rl})-({rr_})") or symbolic_equal(f"({pl})-({pr})", f"-(({rl})-({rr_}))"):
            return True
    elif p.count("=") == 1 and len(p.split("=")[0].strip()) <= 2 and "=" not in r:
        if math_equal(p.split("=")[1], r, include_percentage, is_close, timeout, depth + 1):
            return True
    elif r.count("=") == 1 and len(r.split("=")[0].strip()) <= 2 and "=" not in p:
        if math_equal(p, r.split("=")[1], include_percentage, is_close, timeout, depth + 1):
            return True
    if timeout:
        return bool(call_with_timeout(_symbolic_worker, pb, rb))
    return symbolic_equal(pb, rb)


def math_equal_process(param):
    return math_equal(param[-2], param[-1])


# ------------------------------------------------------------------------------- entry points
def process_results(answer: str, solution: str):
    try:
        pred = extract_answer(answer, "math", use_last_number=False)
        ref = extract_answer(solution, "math", use_last_number=True)
        if pred is None or pred.strip() in ("None", "none", ""):
            return 0, (pred, ref)
        if ref is None or ref.strip() in ("None", "none", ""):
            return 0, (pred, ref)
        return int(math_equal(pred, ref, timeout=False)), (pred, ref)
    except Exception:
        return 0, ("None", "None")


def loadJson(path: str):
    with open(path) as f:
        if path.endswith(".jsonl"):
            return [json.loads(line) for line in f if line.strip()]
        return json.load(f)


def parse_line(id2info: Dict, prompt_str: str, generated: str, query_id: str) -> int:
    info = id2info[query_id.split("@idx:")[0]]
    label = 0
    for sol in info["solutions"]:
        label = label or process_results(generated, sol)[0]
    return label


def parse_lines_in_parallel(id2info: Dict, generateds: List[str], query_ids: List[str], max_workers: int = 8,
                            check_xml_format: bool = False, timeout: float = 15.0) -> List[int]:
    assert len(generateds) == len(query_ids), (len(generateds), len(query_ids))
    labels = []
    with ProcessPoolExecutor(max_workers=max_workers, mp_context=mp.get_context("spawn")) as ex:
        jobs = []
        for qid, gen in zip(query_ids, generateds):
            info = id2info[qid.split("@idx:")[0]]
            jobs.append([ex.submit(process_results, gen, sol) for sol in info["solutions"]])
        for js in jobs:
            label = 0
            for j in js:
                try:
                    label = label or j.result(timeout=timeout)[0]
                except FutTimeout:
                    logger.warning("Timeout occurred while justifying the math answer.")
                except Exception as e:
                    logger.warning(f"verifier error: {e.__class__.__name__}, {e}")
            labels.append(int(label))
    return labels
