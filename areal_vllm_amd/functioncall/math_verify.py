"""Remote math verification: (answer, solution) pairs are shipped in batches to the function-call
service; a query is correct when any of its reference solutions matches
(``functioncall/math/verify.py:13-118``)."""
from typing import Dict, List

from .call import Language, batch_function_call, construct_uid, get_runtime_name, logger


def math_verify(id2info: Dict, generateds: List[str], query_ids: List[str], batch_size: int = 10,
                timeout: float = 1000, max_workers=None) -> List[int]:
    assert len(generateds) == len(query_ids), (len(generateds), len(query_ids))
    params = []
    for idx, (qid, gen) in enumerate(zip(query_ids, generateds)):
        for sol in id2info[qid.split("@idx:")[0]]["solutions"]:
            params.append((gen, sol, idx))
    batches = []
    for i in range(0, len(params), batch_size):
        chunk = params[i:i + batch_size]
        batches.append(dict(
            uid=construct_uid("math", i, i + len(chunk)), language=str(Language.MATH).upper(),
            runtime=get_runtime_name(None, str(Language.MATH)), code='print("hello math!")',
            testcases=[{}] * len(chunk), timeout=5, isFastFail=True,
            extraInfo=dict(answers=[c[0] for c in chunk], solutions=[c[1] for c in chunk],
                           query_ids=[query_ids[c[2]] for c in chunk])))
    results = batch_function_call(batches, "math", timeout)
    labels = [0] * len(query_ids)
    k = 0
    for b, res in zip(batches, results):
        n = len(b["extraInfo"]["answers"])
        ok = (isinstance(res, dict) and isinstance(res.get("results"), list) and len(res["results"]) == n
              and all(isinstance(x, dict) for x in res["results"]))
        if not ok:
            logger.warning(f"Invalid functioncall math results: {res}")
            k += n
            continue
        for item in res["results"]:
            idx = params[k][2]
            if item.get("success", False):
                labels[idx] = 1
            k += 1
    return labels
