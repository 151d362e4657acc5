"""Run model-written Python snippets (tool-integrated reasoning prompts: the model writes a ```python block,
the evaluator executes it and feeds the output back or grades it).

Every snippet runs in its own interpreter process (isolated mode, no site packages, empty environment,
CPU-time and address-space limits, temporary working directory), with a wall-clock timeout; a batch runs
on a small thread pool that only waits on those processes. Output convention of the reference's
``evaluation/python_executor.py``: the value of ``answer``-style last expressions is not guessed - the
snippet's stdout is the result; with ``get_answer_from='solution()'`` the snippet is expected to define
``solution()`` and its return value is printed. Returns ``(output, report)`` with report ``"Done"`` or a
one-line error.
"""
import os
import resource
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor
from typing import List, Optional, Sequence, Tuple

MAX_OUTPUT_CHARS = 8192


def _limits(cpu_seconds: int, mem_bytes: int):
    def apply():
        resource.setrlimit(resource.RLIMIT_CPU, (cpu_seconds, cpu_seconds))
        resource.setrlimit(resource.RLIMIT_AS, (mem_bytes, mem_bytes))
        resource.setrlimit(resource.RLIMIT_NPROC, (64, 64))
        os.setsid()
    return apply


class PythonExecutor:
    def __init__(self, timeout: float = 10.0, get_answer_from: str = "stdout", memory_mb: int = 2048,
                 max_workers: int = 8):
        assert get_answer_from in ("stdout", "solution()")
        self.timeout = timeout
        self.get_answer_from = get_answer_from
        self.memory_mb = memory_mb
        self.max_workers = max_workers

    def run(self, code: str, stdin: Optional[str] = None) -> Tuple[str, str]:
        if self.get_answer_from == "solution()":
            code = code + "\nprint(solution())\n"
        with tempfile.TemporaryDirectory(prefix="areal_pyexec_") as tmp:
            path = os.path.join(tmp, "snippet.py")
            with open(path, "w") as f:
                f.write(code)
            try:
                p = subprocess.run([sys.executable, "-I", "-S", path], input=stdin or "", capture_output=True, text=True,
                                   timeout=self.timeout, cwd=tmp, env={"PATH": "/usr/bin:/bin"},
                                   preexec_fn=_limits(int(self.timeout) + 1, self.memory_mb << 20))
            except subprocess.TimeoutExpired:
                return "", "Timeout Error"
            except OSError as e:
                return "", f"{type(e).__name__}: {e}"
        out = p.stdout[:MAX_OUTPUT_CHARS].strip()
        if p.returncode != 0:
            err = p.stderr.strip().splitlines()
            return out, (err[-1] if err else f"exit code {p.returncode}")[:512]
        return out, "Done"

    def batch_run(self, codes: Sequence[str], stdins: Optional[Sequence[Optional[str]]] = None) -> List[Tuple[str, str]]:
        stdins = list(stdins) if stdins is not None else [None] * len(codes)
        with ThreadPoolExecutor(max_workers=self.max_workers) as pool:
            return list(pool.map(lambda cs: self.run(*cs), zip(codes, stdins)))


def extract_program(text: str, last_only: bool = True) -> str:
    """The ```python blocks of a model response (all of them joined, or only the last one)."""
    import re

    blocks = re.findall(r"```(?:python|py)?\n(.*?)```", text, flags=re.S)
    if not blocks:
        return ""
    return blocks[-1] if last_only else "\n".join(blocks)
