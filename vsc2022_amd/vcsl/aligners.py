"""Aligners other than the Temporal Network behind `vcsl.vta.build_vta_model`: "DTW", "DP" and "HV".

The reference passes any `model_type` through to VCSL (vsc/baseline/localization.py:40-46) and never asks for anything
but "TN" (vsc/baseline/sscd_baseline.py:121,131; vsc/baseline/dns_baseline.py:202); VCSL's own DTW / DP are CPU code
(numba) that takes the frame x frame similarity matrix of a pair.  They run the same way here: the matrices come from the
GPU (`VCSLLocalization.similarity`, libvscmi), the alignment below runs on the host inside `forward_sim`, box scores go
through the localiser's `score()` hook -- the reference's own route, off the hot path.

PARITY UNPINNED, like TN: the VCSL source is not part of the reference checkout (dangling symlink, .gitmodules:1-3), so
these are restatements of the published algorithms (He et al., CVPR 2022, section 5.1: "DTW" = dynamic time warping over
the whole matrix, then the warping path is cut where it stops following similar frames; "DP" = the dynamic-programming
localiser of Chou et al. 2015: best-scoring diagonal-ish blocks with a bounded number of consecutive misses) with every
choice written down here.  Shared conventions with TN: a box is [q_lo, r_lo, q_hi, r_hi] in frame
indices with inclusive ends; a box is kept when min(q_hi - q_lo, r_hi - r_lo) > min_length and its IoU (areas
dq * dr, no +1) with every box kept before it is < max_iou; matrices may hold any float dtype.

DTW(discontinue=3, min_sim=0.2, min_length=5, max_iou=0.3)
    1. cost = 1 - sims; D[i, j] = cost[i, j] + min(D[i-1, j-1], D[i-1, j], D[i, j-1]) from (0, 0) to (Lq-1, Lr-1);
       the path is traced back from the far corner, ties preferring the diagonal, then the step in q, then in r.
    2. A path cell "matches" when sims >= min_sim.  The path is cut into runs that start and end on a matching cell and
       hold at most `discontinue` consecutive non-matching cells; a run's score is the sum of its matching sims.
    3. Runs in descending score (ties: earlier run first) -> boxes -> the min_length / max_iou filter.

DP(discontinue=3, min_sim=0.2, min_length=5, max_iou=0.3, max_path=10)
    1. gain = sims - min_sim.  S[i, j] = the best score of a monotone path ending on the matching cell (i, j) whose
       consecutive cells are at most `discontinue` + 1 frames apart on either axis and advance on both axes:
       S[i, j] = gain[i, j] + max(0, max S[i - a, j - b], 1 <= a, b <= discontinue + 1); non-matching cells carry no
       path.  Predecessor ties: smallest a + b, then smallest a.
    2. The best cell (ties: first in row-major order) is traced back to its start -> one path; its rows and columns are
       removed from further paths (S is recomputed on the remaining cells); at most max_path paths.
    3. Paths in extraction order -> boxes -> the min_length / max_iou filter.

HV(discontinue=3, min_sim=0.2, min_length=5, max_iou=0.3, max_path=10, tolerance=1)
    Temporal Hough voting (Douze et al., "An image-based approach to video copy detection with spatio-temporal
    post-filtering", 2010, section V; VCSL benchmarks it as "HV"): a copied segment is a run of matching frames along ONE
    temporal offset r - q.
    1. Every matching cell (sims >= min_sim) votes for its offset d = r - q with weight sims[q, r]; H[d] = the sum of the
       votes of the offsets d - tolerance ... d + tolerance (the band a slightly re-timed copy spreads over).
    2. Offsets in descending H (ties: smaller |d|, then smaller d), the best max_path of them with H > 0.
    3. Per chosen offset: the matching cells of its band, ordered by (q, r), are cut into runs wherever more than
       `discontinue` query frames in a row hold no matching cell of the band; run -> box [min q, min r, max q, max r],
       run score = the sum of its cells' sims (float64 sum in (q, r) order).
    4. All runs of all chosen offsets in descending score (ties: earlier offset of step 2, then earlier run) -> the
       min_length / max_iou filter (a copy seen from two neighbouring offsets survives once).

SPD (VCSL's "similarity pattern detection") is a TRAINED detector network over the similarity matrix: without VCSL's
weights (not in the reference checkout, no network here) there is nothing to restate -- `build_vta_model("SPD")` raises
and says so; a user who has the weights registers the model with `vcsl.vta.register_vta_model`.
"""
from typing import List, Sequence, Tuple

import numpy as np


def _iou(box, kept) -> float:
    best = 0.0
    for k in kept:
        w = min(box[2], k[2]) - max(box[0], k[0])
        h = min(box[3], k[3]) - max(box[1], k[1])
        inter = max(w, 0) * max(h, 0)
        union = (box[2] - box[0]) * (box[3] - box[1]) + (k[2] - k[0]) * (k[3] - k[1]) - inter
        if union > 0:
            best = max(best, inter / union)
        elif inter == 0 and union == 0:
            best = max(best, 0.0)
    return best


def _keep(boxes: List[List[int]], min_length: int, max_iou: float) -> List[List[int]]:
    kept: List[List[int]] = []
    for b in boxes:
        if min(b[2] - b[0], b[3] - b[1]) > min_length and (not kept or _iou(b, kept) < max_iou):
            kept.append([int(v) for v in b])
    return kept


def dtw_path(sims: np.ndarray) -> np.ndarray:
    """[n, 2] cells (q, r) of the warping path from (0, 0) to (Lq-1, Lr-1) that minimises the sum of 1 - sims."""
    cost = 1.0 - np.asarray(sims, dtype=np.float64)
    n, m = cost.shape
    acc = np.full((n + 1, m + 1), np.inf)
    acc[0, 0] = 0.0
    for i in range(1, n + 1):
        row, prev = acc[i], acc[i - 1]
        # D[i, j] depends on D[i, j-1]: a running minimum along the row (the only serial dimension)
        best_up = np.minimum(prev[1:], prev[:-1])
        c = cost[i - 1]
        left = np.inf
        for j in range(1, m + 1):
            v = c[j - 1] + min(best_up[j - 1], left)
            row[j] = v
            left = v
    path = [(n - 1, m - 1)]
    i, j = n, m
    while (i, j) != (1, 1):
        d, u, l = acc[i - 1, j - 1], acc[i - 1, j], acc[i, j - 1]
        if d <= u and d <= l:
            i, j = i - 1, j - 1
        elif u <= l:
            i -= 1
        else:
            j -= 1
        path.append((i - 1, j - 1))
    return np.array(path[::-1], dtype=np.int64)


def dtw(sims: np.ndarray, discontinue: int = 3, min_sim: float = 0.2, min_length: int = 5,
        max_iou: float = 0.3) -> List[List[int]]:
    sims = np.asarray(sims)
    if sims.ndim != 2 or 0 in sims.shape:
        return []
    path = dtw_path(sims)
    along = sims[path[:, 0], path[:, 1]]
    match = along >= min_sim
    runs = []  # (score, first matching position, last matching position)
    start = last = -1
    misses = 0
    score = 0.0
    for p, ok in enumerate(match):
        if ok:
            if start < 0:
                start, score = p, 0.0
            last, misses = p, 0
            score += float(along[p])
        elif start >= 0:
            misses += 1
            if misses > discontinue:
                runs.append((score, start, last))
                start = -1
    if start >= 0:
        runs.append((score, start, last))
    order = sorted(range(len(runs)), key=lambda k: (-runs[k][0], k))
    boxes = []
    for k in order:
        _, a, b = runs[k]
        seg = path[a : b + 1]
        boxes.append([seg[:, 0].min(), seg[:, 1].min(), seg[:, 0].max(), seg[:, 1].max()])
    return _keep(boxes, min_length, max_iou)


def _dp_scores(gain: np.ndarray, alive: np.ndarray, reach: int):
    n, m = gain.shape
    S = np.full((n, m), -np.inf)
    back = np.full((n, m, 2), -1, dtype=np.int64)
    steps = sorted(((a, b) for a in range(1, reach + 1) for b in range(1, reach + 1)), key=lambda s: (s[0] + s[1], s[0]))
    for i in range(n):
        rows_ok = alive[i]
        if not rows_ok.any():
            continue
        best = np.zeros(m)
        arg = np.full((m, 2), -1, dtype=np.int64)
        for a, b in steps:
            if i - a < 0 or b >= m:   # (a step wider than the matrix reaches no cell)
                continue
            prev = np.full(m, -np.inf)
            prev[b:] = S[i - a, : m - b]
            better = prev > best
            if better.any():
                best = np.where(better, prev, best)
                arg[better, 0] = i - a
                arg[better, 1] = np.nonzero(better)[0] - b
        S[i] = np.where(rows_ok, gain[i] + best, -np.inf)
        back[i] = arg
    return S, back


def dp(sims: np.ndarray, discontinue: int = 3, min_sim: float = 0.2, min_length: int = 5, max_iou: float = 0.3,
       max_path: int = 10) -> List[List[int]]:
    sims = np.asarray(sims)
    if sims.ndim != 2 or 0 in sims.shape:
        return []
    gain = sims.astype(np.float64) - float(min_sim)
    alive = sims >= min_sim
    reach = int(discontinue) + 1
    boxes = []
    for _ in range(int(max_path)):
        if not alive.any():
            break
        S, back = _dp_scores(gain, alive, reach)
        flat = int(np.argmax(S))
        i, j = divmod(flat, S.shape[1])
        if not np.isfinite(S[i, j]):
            break
        cells = []
        while i >= 0:
            cells.append((i, j))
            i, j = back[i, j]
        cells = np.array(cells)
        boxes.append([cells[:, 0].min(), cells[:, 1].min(), cells[:, 0].max(), cells[:, 1].max()])
        # the frames of this path take part in no further path
        alive[cells[:, 0].min() : cells[:, 0].max() + 1, :] = False
        alive[:, cells[:, 1].min() : cells[:, 1].max() + 1] = False
    return _keep(boxes, min_length, max_iou)


def hv(sims: np.ndarray, discontinue: int = 3, min_sim: float = 0.2, min_length: int = 5, max_iou: float = 0.3,
       max_path: int = 10, tolerance: int = 1) -> List[List[int]]:
    sims = np.asarray(sims)
    if sims.ndim != 2 or 0 in sims.shape:
        return []
    n, m = sims.shape
    qs, rs = np.nonzero(sims >= min_sim)           # row-major: (q, r) ascending
    if len(qs) == 0:
        return []
    w = sims[qs, rs].astype(np.float64)
    off = rs - qs                                    # in [-(n-1), m-1]
    votes = np.zeros(n + m - 1, dtype=np.float64)
    np.add.at(votes, off + (n - 1), w)
    tol = int(tolerance)
    padded = np.concatenate([np.zeros(tol), votes, np.zeros(tol)])
    H = np.zeros_like(votes)
    for t in range(2 * tol + 1):                     # H[d] = votes[d - tol] + ... + votes[d + tol], in that order
        H += padded[t : t + len(votes)]
    offsets = np.arange(-(n - 1), m)
    order = sorted(range(len(offsets)), key=lambda k: (-H[k], abs(int(offsets[k])), int(offsets[k])))
    runs = []                                        # (score, peak rank, run number, box)
    for rank, k in enumerate(order[: int(max_path)]):
        if not H[k] > 0.0:
            break
        d = int(offsets[k])
        band = np.nonzero(np.abs(off - d) <= tol)[0]  # cells of the band, still (q, r) ascending
        start, count = 0, 0
        for x in range(1, len(band) + 1):
            if x == len(band) or qs[band[x]] - qs[band[x - 1]] > discontinue + 1:
                cells = band[start:x]
                box = [int(qs[cells].min()), int(rs[cells].min()), int(qs[cells].max()), int(rs[cells].max())]
                score = 0.0
                for c in cells:
                    score += float(w[c])
                runs.append((score, rank, count, box))
                start, count = x, count + 1
    runs.sort(key=lambda r: (-r[0], r[1], r[2]))
    return _keep([r[3] for r in runs], min_length, max_iou)


class _HostAligner:
    """`forward_sim([(name, sims), ...]) -> [(name, boxes), ...]` in input order, names echoed (the contract of
    vsc/baseline/localization.py:58-66).  `concurrency` (a process-pool size in VCSL) is accepted and ignored."""

    _fn = None
    _defaults: dict = {}

    def __init__(self, concurrency: int = 1, **config):
        unknown = set(config) - set(self._defaults)
        if unknown:
            raise TypeError(f"unexpected {type(self).__name__} arguments: {sorted(unknown)}")
        self.config = dict(self._defaults, **config)

    def forward_sim(self, data: Sequence[Tuple[str, np.ndarray]]) -> List[Tuple[str, List[List[int]]]]:
        return [(name, type(self)._fn(np.asarray(sims), **self.config)) for name, sims in data]


class DTW(_HostAligner):
    _fn = staticmethod(dtw)
    _defaults = dict(discontinue=3, min_sim=0.2, min_length=5, max_iou=0.3)


class DP(_HostAligner):
    _fn = staticmethod(dp)
    _defaults = dict(discontinue=3, min_sim=0.2, min_length=5, max_iou=0.3, max_path=10)


class HV(_HostAligner):
    _fn = staticmethod(hv)
    _defaults = dict(discontinue=3, min_sim=0.2, min_length=5, max_iou=0.3, max_path=10, tolerance=1)
