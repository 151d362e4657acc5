"""Remote code verification: each generated program runs against the test cases of its problem in the
function-call sandbox; all cases must pass (``functioncall/code/verify.py`` in the reference)."""
import json
from typing import Dict, List

from .call import Language, batch_function_call, construct_uid, get_runtime_name, logger

SINGLE_CASE_EXEC_TIMEOUT = 6


def _testcases(info: Dict) -> List[Dict]:
    io = info["input_output"]
    io = json.loads(io) if isinstance(io, str) else io
    return [dict(input=i, expectedOutput=o) for i, o in zip(io.get("inputs", []), io.get("outputs", []))], io


def code_verify(id2info: Dict, generateds: List[str], query_ids: List[str], timeout: float = 1000,
                timeout_for_testcase: int = SINGLE_CASE_EXEC_TIMEOUT, test_case_batch_size: int = 20) -> List[int]:
    assert len(generateds) == len(query_ids)
    payloads, owner = [], []
    for idx, (qid, code) in enumerate(zip(query_ids, generateds)):
        cases, io = _testcases(id2info[qid.split("@idx:")[0]])
        for i in range(0, max(1, len(cases)), test_case_batch_size):
            payloads.append(dict(
                uid=construct_uid(qid, i, i + test_case_batch_size), language=str(Language.PYTHON).upper(),
                runtime=get_runtime_name(None, str(Language.PYTHON)), code=code or "",
                entryFunction=io.get("fn_name", ""), testcases=cases[i:i + test_case_batch_size],
                timeout=timeout_for_testcase, isFastFail=True))
            owner.append(idx)
    results = batch_function_call(payloads, "code", timeout)
    labels = [1 if g else 0 for g in generateds]
    for idx, res in zip(owner, results):
        if not (isinstance(res, dict) and res.get("success", False)):
            labels[idx] = 0
    return labels
