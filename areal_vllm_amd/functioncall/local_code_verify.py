"""Local code verification: run every program against its stdin/stdout test cases in a separate Python
subprocess with CPU-time, memory and wall-clock limits (``functioncall/code/local_verify.py`` in the
reference). Function-style problems (``fn_name``) are wrapped in a driver that calls the function with
the JSON arguments."""
import json
import os
import resource
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List

SINGLE_CASE_EXEC_TIMEOUT = 6
MEM_LIMIT_BYTES = 2 << 30


def _limits():
    resource.setrlimit(resource.RLIMIT_AS, (MEM_LIMIT_BYTES, MEM_LIMIT_BYTES))
    resource.setrlimit(resource.RLIMIT_CPU, (SINGLE_CASE_EXEC_TIMEOUT + 1, SINGLE_CASE_EXEC_TIMEOUT + 2))
    os.setsid()


def _same(a: str, b: str) -> bool:
    return [l.rstrip() for l in a.strip().splitlines()] == [l.rstrip() for l in b.strip().splitlines()]


def run_one(code: str, io: Dict, timeout: float = SINGLE_CASE_EXEC_TIMEOUT) -> bool:
    if not code:
        return False
    fn = io.get("fn_name")
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "prog.py")
        src = code
        if fn:
            src += (f"\n\nif __name__ == '__main__':\n    import sys, json\n    _a = json.loads(sys.stdin.read())\n"
                    f"    _f = {fn} if '{fn}' in globals() else getattr(Solution(), '{fn}')\n"
                    f"    print(json.dumps(_f(*_a)))\n")
        with open(path, "w") as f:
            f.write(src)
        for inp, exp in zip(io.get("inputs", []), io.get("outputs", [])):
            stdin = json.dumps(inp) if fn else (inp if isinstance(inp, str) else "\n".join(map(str, inp)))
            try:
                r = subprocess.run([sys.executable, "-I", path], input=stdin, capture_output=True, text=True,
                                   timeout=timeout, preexec_fn=_limits, cwd=d)
            except subprocess.TimeoutExpired:
                return False
            if r.returncode != 0:
                return False
            if fn:
                try:
                    got = json.loads(r.stdout.strip().splitlines()[-1])
                except Exception:
                    return False
                want = exp[0] if isinstance(exp, list) and len(exp) == 1 and not isinstance(got, list) else exp
                if got != want and got != exp:
                    return False
            elif not _same(r.stdout, exp if isinstance(exp, str) else "\n".join(map(str, exp))):
                return False
    return True


def code_verify(id2info: Dict, generateds: List[str], query_ids: List[str], max_workers: int = 4, **kw) -> List[int]:
    assert len(generateds) == len(query_ids)

    def job(args):
        code, qid = args
        io = id2info[qid.split("@idx:")[0]]["input_output"]
        io = json.loads(io) if isinstance(io, str) else io
        return int(run_one(code, io))

    with ThreadPoolExecutor(max_workers=max(1, max_workers)) as ex:
        return list(ex.map(job, zip(generateds, query_ids)))
