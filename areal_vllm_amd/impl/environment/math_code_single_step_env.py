"""Single-step verification environment for math and code prompts.

``step((qid, answers, ...))`` returns one 0/1 reward per answer. Math answers go through the local math
verifier (or the remote function-call service when ``FUNCTIONCALL_SERVICE_DOMAIN`` is set); code answers
have their last fenced code block extracted and are run against the prompt's test cases locally in a
subprocess sandbox or remotely. Behaviour of ``realhf/impl/environment/math_code_single_step_env.py:41-78``.
"""
import asyncio
import json
import os
import re
from typing import Dict, List, Optional, Tuple

from ...reward.math_parser import parse_lines_in_parallel
from ...system.agent_api import EnvironmentService, register_environment
from ...utils import logging

logger = logging.getLogger("math-code-env")


def extract_code(text: str, min_length: int = 20) -> Optional[str]:
    blocks = re.findall(r"(?i)```(?:python|py|cpp)?\s*\n?(.*?)\n?```", text, re.DOTALL)
    blocks = [b.strip() for b in blocks if len(b.strip()) >= min_length]
    return blocks[-1] if blocks else None


def load_metadata(path: str) -> Tuple[Dict[str, Dict], Dict[str, int]]:
    """jsonl with fields query_id, task (math|code), solutions (math) / input_output (code)."""
    id2info, task_cnt = {}, {}
    with open(path) as f:
        for line in f:
            if not line.strip():
                continue
            d = json.loads(line)
            d.setdefault("task", "math")
            id2info[str(d["query_id"])] = d
            task_cnt[d["task"]] = task_cnt.get(d["task"], 0) + 1
    return id2info, task_cnt


def math_verify_call(id2info, answers, qids, max_workers=1):
    if os.getenv("FUNCTIONCALL_SERVICE_DOMAIN", ""):
        from ...functioncall.math_verify import math_verify

        return math_verify(id2info, answers, qids)
    return parse_lines_in_parallel(id2info, answers, qids, max_workers=max_workers)


def code_verify_call(id2info, answers, qids, max_workers=1):
    if os.getenv("FUNCTIONCALL_SERVICE_DOMAIN", ""):
        from ...functioncall.code_verify import code_verify

        return code_verify(id2info, answers, qids)
    from ...functioncall.local_code_verify import code_verify as local

    return local(id2info, answers, qids, max_workers=max_workers)


class MathCodeSingleStepEnv(EnvironmentService):
    def __init__(self, dataset_path: Optional[str] = None, id2info: Optional[Dict] = None):
        self.id2info = id2info if id2info is not None else load_metadata(dataset_path)[0]

    async def step(self, action):
        qid, answers = action[0], action[1]
        qid = str(qid).split("@")[0]
        task = self.id2info[qid].get("task", "math")
        n = len(answers)
        if task == "math":
            r = await asyncio.to_thread(math_verify_call, self.id2info, list(answers), [qid] * n, 1)
        elif task == "code":
            codes = [extract_code(a) or "" for a in answers]
            r = await asyncio.to_thread(code_verify_call, self.id2info, codes, [qid] * n, 1)
        else:
            raise NotImplementedError(task)
        return None, r, True, False, {}


register_environment("math-code-single-step", MathCodeSingleStepEnv)
