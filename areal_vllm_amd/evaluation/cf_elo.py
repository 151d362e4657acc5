"""Codeforces Elo estimate of a code model from its per-problem attempt outcomes.

For every contest with cached official standings the model is scored like a virtual participant (first
successful attempt of a problem earns the problem's points minus 50 per failed attempt before it, or one
solved problem plus 10 penalty minutes per failed attempt under ICPC rules), ranked among the official
participants, and assigned the rating whose expected rank - one plus the sum over participants of their
win probability against that rating, ``1 / (1 + 10 ** ((r - r_i) / 400))`` - equals that rank. The reported
rating is the mean over contests, its percentile is taken in a histogram of human ratings. Same procedure
and file formats as the reference's ``evaluation/cf_elo_caculator.py`` (standings / rating-change dumps of
the Codeforces API keyed by contest id, a ``{rating: count}`` histogram, a list of problem ids aligned with
the evaluated samples).
"""
import bisect
import json
import re
from collections import defaultdict
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

MIN_PARTICIPANTS = 200  # contests with fewer rated participants give unstable estimates and are skipped


def percentile(rating: float, sorted_ratings: Sequence[float]) -> float:
    """Share (in %) of the rating population strictly below `rating`."""
    if not sorted_ratings:
        return 0.0
    return round(100.0 * bisect.bisect_left(sorted_ratings, float(rating)) / len(sorted_ratings), 1)


def load_rating_histogram(path: str) -> List[float]:
    with open(path) as f:
        hist = json.load(f)
    out: List[float] = []
    for rating, count in hist.items():
        out.extend([float(rating)] * int(count))
    return sorted(out)


def group_by_contest(problem_ids: Sequence[str], outcomes: Sequence[Sequence[bool]]) -> Dict[int, Dict[str, List[bool]]]:
    """``["1922A", "1922B", ...]`` + per-problem attempt outcomes -> {contest id: {problem id: outcomes}}."""
    if len(problem_ids) != len(outcomes):
        raise ValueError(f"{len(problem_ids)} problem ids but {len(outcomes)} results")
    grouped: Dict[int, Dict[str, List[bool]]] = defaultdict(dict)
    for pid, res in zip(problem_ids, outcomes):
        m = re.match(r"(\d+)([A-Z].*)", pid)
        if not m:
            raise ValueError(f"not a Codeforces problem id: {pid!r}")
        grouped[int(m.group(1))][pid] = [bool(x) for x in res]
    return dict(grouped)


def virtual_score(problems: Sequence[Dict], status: Dict[str, List[bool]], pass_n: Optional[int] = None) -> Tuple[float, int]:
    """(score, penalty) of the attempts in `status` on the contest's `problems` (entries of the standings'
    ``problems`` list: ``contestId``, ``index`` and, in point-scored contests, ``points``)."""
    score, penalty = 0.0, 0
    for p in problems:
        pid = f"{p['contestId']}{p['index']}"
        tries = status.get(pid)
        if not tries:
            continue
        tries = tries[: pass_n or len(tries)]
        if True in tries:
            first = tries.index(True)
            if "points" in p:
                score += max(0.0, float(p["points"]) - 50.0 * first)
            else:
                score += 1.0
                penalty += 10 * first
    return score, penalty


def rank_in_standings(rows: Sequence[Dict], score: float, penalty: int) -> int:
    """0-based position the virtual participant takes in the official (sorted) standings: before the first
    row with fewer points, or equal points and more penalty."""
    for i, row in enumerate(rows):
        if row["points"] < score or (row["points"] == score and row["penalty"] > penalty):
            return i
    return len(rows)


def rating_for_rank(old_ratings: np.ndarray, rank: int) -> int:
    """Largest integer rating whose expected rank among `old_ratings` is still >= `rank`."""
    lo, hi = 0, int(old_ratings.max()) + 100
    while hi - lo > 1:
        mid = (lo + hi) // 2
        seed = 1.0 + float(np.sum(1.0 / (1.0 + 10.0 ** ((mid - old_ratings) / 400.0))))
        if seed < rank:
            hi = mid
        else:
            lo = mid
    return lo


def contest_rating(contest: Dict, status: Dict[str, List[bool]], pass_n: Optional[int] = None) -> Optional[int]:
    """Rating estimate for one contest, None when its cached data is unusable. `contest` holds the API
    dumps ``{"standings": {...}, "rating_changes": {...}}``."""
    try:
        st, rc = contest["standings"], contest["rating_changes"]
        if st.get("status") != "OK" or rc.get("status") != "OK":
            return None
        changes = {c["handle"]: c for c in rc["result"]}
        rows = [r for r in st["result"]["rows"] if r["party"]["members"][0]["handle"] in changes]
        if len(rows) <= MIN_PARTICIPANTS:
            return None
        old = np.array([changes[r["party"]["members"][0]["handle"]]["oldRating"] for r in rows], dtype=np.float64)
        score, penalty = virtual_score(st["result"]["problems"], status, pass_n)
        return rating_for_rank(old, rank_in_standings(rows, score, penalty))
    except (KeyError, IndexError, TypeError, ValueError):
        return None


def estimate_elo(problem_ids: Sequence[str], outcomes: Sequence[Sequence[bool]], contests: Dict[str, Dict],
                 sorted_ratings: Sequence[float], pass_n: Optional[int] = 1) -> Optional[Dict]:
    """Mean rating over all contests that could be scored + its percentile + per-contest details."""
    per_contest, skipped = [], []
    for cid, status in sorted(group_by_contest(problem_ids, outcomes).items()):
        data = contests.get(str(cid))
        rating = contest_rating(data, status, pass_n) if data is not None else None
        if rating is None:
            skipped.append(cid)
        else:
            per_contest.append(dict(contest_id=cid, rating=rating, percentile=percentile(rating, sorted_ratings)))
    if not per_contest:
        return None
    mean = float(np.mean([c["rating"] for c in per_contest]))
    return dict(estimated_rating=mean, estimated_percentile=percentile(mean, sorted_ratings),
                contests_processed=len(per_contest), contests_skipped=len(skipped), skipped_contest_ids=skipped,
                individual_contest_results=per_contest)


def estimate_elo_from_files(samples: Sequence[Dict], metadata_path: str, ratings_path: str, contests_path: str,
                            pass_n: Optional[int] = 1) -> Optional[Dict]:
    """`samples`: evaluated rows with ``idx`` (position in the metadata list) and ``score`` (per-attempt
    outcomes); a row without scores counts as one failed attempt."""
    with open(metadata_path) as f:
        problem_ids = json.load(f)
    with open(contests_path) as f:
        contests = json.load(f)
    rows = sorted(samples, key=lambda s: s["idx"])
    outcomes = [[bool(x) for x in s["score"]] if "score" in s else [False] for s in rows]
    return estimate_elo(problem_ids, outcomes, contests, load_rating_histogram(ratings_path), pass_n)
