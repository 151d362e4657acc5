"""Sequence packing helpers: first-fit-decreasing micro-batch allocation and balanced partitioning.

The C++ runtime (``csrc/runtime/runtime.cpp``) provides the fast paths; pure-Python versions with the
same results are kept for environments where the extension is not built. Reference semantics:
``areal/utils/datapack.py:149-227`` (``ffd_allocate``) and its numba ``partition_balanced``.
"""
import bisect
from typing import List, Sequence, Tuple

import numpy as np

try:
    from areal_vllm_amd._C import _runtime as _rt
except ImportError:  # extension not built
    _rt = None


def _ffd_allocate_py(values: Sequence[int], capacity: int, min_groups: int) -> List[List[int]]:
    order = sorted(range(len(values)), key=lambda i: -values[i])
    groups: List[List[int]] = []
    by_sum: List[Tuple[int, int]] = []  # (sum, group id) sorted
    for idx in order:
        v = values[idx]
        if len(groups) < min_groups or not by_sum or by_sum[0][0] + v > capacity:
            bisect.insort(by_sum, (v, len(groups)))
            groups.append([idx])
            continue
        end = bisect.bisect_right(by_sum, (capacity - v, len(values)))
        best = min(range(end), key=lambda j: (len(groups[by_sum[j][1]]), j))
        s, g = by_sum.pop(best)
        groups[g].append(idx)
        bisect.insort(by_sum, (s + v, g))
    return groups


def ffd_allocate(values: Sequence[int], capacity: int, min_groups: int = 1) -> List[List[int]]:
    """Partition indices of `values` into >= min_groups groups whose sums stay <= capacity.

    Greedy: visit values in decreasing order; open a new group while fewer than `min_groups` exist or
    nothing fits; otherwise put the value into the fitting group that currently holds the fewest items.
    """
    values = [int(v) for v in values]
    if any(v > capacity for v in values):
        raise RuntimeError(f"a value exceeds the capacity {capacity}: max={max(values)}")
    if len(values) < min_groups:
        raise RuntimeError(f"number of values {len(values)} is smaller than min_groups {min_groups}")
    if _rt is not None:
        return [list(g) for g in _rt.ffd_allocate(values, int(capacity), int(min_groups))]
    return _ffd_allocate_py(values, capacity, min_groups)


def partition_balanced(lens: Sequence[int], k: int, min_size: int = 1) -> List[int]:
    """Split `lens` (order preserved) into k contiguous parts minimising the largest part sum.
    Returns the k+1 boundaries."""
    lens = [int(x) for x in lens]
    if _rt is not None:
        return list(_rt.partition_balanced(lens, int(k), int(min_size)))
    n = len(lens)
    min_size = max(1, min_size)
    if k <= 0 or n < k * min_size:
        raise RuntimeError("partition_balanced: infeasible")
    pre = np.concatenate([[0], np.cumsum(lens)])
    INF = float("inf")
    dp = [[INF] * (n + 1) for _ in range(k + 1)]
    arg = [[-1] * (n + 1) for _ in range(k + 1)]
    dp[0][0] = 0
    for j in range(1, k + 1):
        for i in range(j * min_size, n - (k - j) * min_size + 1):
            for s in range((j - 1) * min_size, i - min_size + 1):
                if dp[j - 1][s] == INF:
                    continue
                c = max(dp[j - 1][s], pre[i] - pre[s])
                if c < dp[j][i]:
                    dp[j][i], arg[j][i] = c, s
    bounds = [0] * (k + 1)
    i = n
    for j in range(k, -1, -1):
        bounds[j] = i
        if j > 0:
            i = arg[j][i]
    return bounds


def reorder_to_balanced_batches(seqlens: Sequence[int], n_groups: int) -> Tuple[np.ndarray, int]:
    """Greedy longest-first assignment of sequences to `n_groups` bins with near-equal token counts.
    Returns (indices concatenated bin by bin, largest pairwise difference of bin sizes)."""
    order = np.argsort(-np.asarray(seqlens))
    bins: List[List[int]] = [[] for _ in range(n_groups)]
    sums = np.zeros(n_groups, dtype=np.int64)
    for i in order:
        j = int(np.argmin(sums))
        bins[j].append(int(i))
        sums[j] += seqlens[i]
    out: List[int] = []
    for j in np.argsort(-sums):
        out.extend(bins[j])
    return np.asarray(out), int(sums.max() - sums.min())


def flat2d(xs):
    return [y for x in xs for y in x]


def balanced_partition(lens: Sequence[int], k: int, equal_size: bool = False) -> List[List[int]]:
    """Assign item indices to `k` parts with near-equal token sums (longest first, to the lightest part).
    With `equal_size` the item counts of the parts differ by at most one as well (training batches: every
    data-parallel replica must see the same number of samples). Indices inside a part keep input order."""
    n = len(lens)
    assert n >= k, f"cannot split {n} items into {k} non-empty parts"
    cap = (n + k - 1) // k if equal_size else n
    order = sorted(range(n), key=lambda i: (-int(lens[i]), i))
    parts: List[List[int]] = [[] for _ in range(k)]
    sums = [0] * k
    for pos, i in enumerate(order):
        remaining = n - pos
        empty = [j for j in range(k) if not parts[j]]
        if len(empty) >= remaining:  # every remaining item is needed to keep the parts non-empty
            j = empty[0]
        else:
            cands = [j for j in range(k) if len(parts[j]) < cap]
            j = min(cands, key=lambda j: (sums[j], j))
        parts[j].append(i)
        sums[j] += int(lens[i])
    return [sorted(p) for p in parts]
