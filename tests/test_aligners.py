"""CPU: the DTW / DP / HV aligners behind `vcsl.vta.build_vta_model` (vsc2022_amd/vcsl/aligners.py).
Parity with VCSL's own implementations is unpinned (their source is not in the reference checkout, like TN's): these
tests pin the `forward_sim` contract the reference relies on (vsc/baseline/localization.py:58-66), the properties of the
reference's localisation tests (tests/test_localization.py:46-66: a planted copy is found, unrelated videos yield
nothing), and the documented semantics on small hand-checked matrices."""
import numpy as np
import pytest

from hypothesis import given, settings, strategies as st

from vsc2022_amd.vcsl.aligners import dp, dtw, dtw_path, hv
from vsc2022_amd.vcsl.vta import build_vta_model


def _planted(seed=0, lq=45, lr=60, q0=20, r0=30, n=12, d=64, noise=0.05):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((lq, d))
    b = rng.standard_normal((lr, d))
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    seg = b[r0 : r0 + n] + noise * rng.standard_normal((n, d))
    a[q0 : q0 + n] = seg / np.linalg.norm(seg, axis=1, keepdims=True)
    return (a @ b.T).astype(np.float32)


@pytest.mark.parametrize("name", ["DTW", "DP", "HV"])
def test_forward_sim_contract_and_planted_copy(name):
    model = build_vta_model(name, concurrency=16, min_sim=0.5, min_length=4)   # kwargs of the reference's calls are accepted
    sims = _planted()
    rng = np.random.default_rng(5)
    unrelated = (0.1 * rng.standard_normal((40, 30))).astype(np.float32)
    out = model.forward_sim([("1-2", unrelated), ("1-3", sims), ("e", np.zeros((0, 7), np.float32))])
    assert [n for n, _ in out] == ["1-2", "1-3", "e"]
    assert out[0][1] == [] and out[2][1] == []
    boxes = out[1][1]
    assert len(boxes) == 1
    x1, y1, x2, y2 = boxes[0]
    assert all(isinstance(v, int) for v in boxes[0])
    assert (x1, y1) == (20, 30) and (x2, y2) == (31, 41)        # inclusive ends of the 12-frame copy
    with pytest.raises(TypeError):
        build_vta_model(name, no_such_parameter=1)


def test_dtw_path_is_the_optimal_monotone_path():
    rng = np.random.default_rng(1)
    for _ in range(20):
        n, m = int(rng.integers(1, 7)), int(rng.integers(1, 7))
        sims = rng.random((n, m))
        path = dtw_path(sims)
        assert tuple(path[0]) == (0, 0) and tuple(path[-1]) == (n - 1, m - 1)
        steps = np.diff(path, axis=0)
        assert ((steps >= 0) & (steps <= 1)).all() and (steps.sum(axis=1) >= 1).all()
        cost = (1 - sims[path[:, 0], path[:, 1]]).sum()

        def best(i, j, memo={}):   # brute force over all monotone paths
            key = (i, j, id(sims))
            if key in memo:
                return memo[key]
            c = 1 - sims[i, j]
            if i == 0 and j == 0:
                r = c
            else:
                r = c + min(best(a, b) for a, b in ((i - 1, j - 1), (i - 1, j), (i, j - 1)) if a >= 0 and b >= 0)
            memo[key] = r
            return r

        assert abs(cost - best(n - 1, m - 1)) < 1e-9


def test_dtw_cuts_the_path_at_discontinuities_and_filters_boxes():
    sims = np.zeros((30, 30), np.float32)
    for k in range(8):
        sims[2 + k, 2 + k] = 0.9           # first copy
    for k in range(9):
        sims[18 + k, 19 + k] = 0.8         # second copy, 8 unmatched diagonal cells after the first
    boxes = dtw(sims, discontinue=3, min_sim=0.3, min_length=5)
    assert boxes == [[18, 19, 26, 27], [2, 2, 9, 9]]            # by descending run score (fp32 sums: 9 x 0.8 > 8 x 0.9 by one ulp)
    # a tolerated gap (2 misses <= discontinue) joins two pieces into one run
    sims2 = np.zeros((20, 20), np.float32)
    for k in (0, 1, 2, 3, 6, 7, 8, 9, 10):
        sims2[3 + k, 3 + k] = 0.7
    assert dtw(sims2, discontinue=3, min_sim=0.3, min_length=5) == [[3, 3, 13, 13]]
    assert dtw(sims2, discontinue=1, min_sim=0.3, min_length=2) == [[9, 9, 13, 13], [3, 3, 6, 6]]
    assert dtw(sims2, discontinue=3, min_sim=0.3, min_length=10) == []   # min(dq, dr) must EXCEED min_length


def test_dp_extracts_disjoint_blocks_best_first():
    sims = np.zeros((40, 50), np.float32)
    for k in range(10):
        sims[5 + k, 8 + k] = 0.9            # block A: score 10 * 0.7
    for k in range(0, 14, 2):
        sims[22 + k, 30 + k] = 0.8          # block B: every other frame (1 miss between matches), score 7 * 0.6
    boxes = dp(sims, discontinue=3, min_sim=0.2, min_length=5)
    assert boxes == [[5, 8, 14, 17], [22, 30, 34, 42]]
    assert dp(sims, discontinue=0, min_sim=0.2, min_length=5) == [[5, 8, 14, 17]]   # B's matches are 2 frames apart
    assert dp(sims, discontinue=3, min_sim=0.2, min_length=5, max_path=1) == [[5, 8, 14, 17]]
    assert dp(np.full((6, 6), 0.1, np.float32)) == []


# ---------------------------------------------------------------------------------------------------------------------
# Independent checks: every aligner against a brute-force restatement of its documented semantics
# (vsc2022_amd/vcsl/aligners.py, module docstring) that shares no code with it -- exhaustive enumeration instead of
# dynamic programming, dictionaries instead of arrays -- on small hypothesis-drawn matrices.  Values are continuous, so
# the optima are unique and no tie rule is exercised here (the hand-checked cases above pin those).


def _bf_keep(boxes, min_length, max_iou):
    kept = []
    for b in boxes:
        if not min(b[2] - b[0], b[3] - b[1]) > min_length:
            continue
        ok = True
        for k in kept:
            iw = max(0, min(b[2], k[2]) - max(b[0], k[0]))
            ih = max(0, min(b[3], k[3]) - max(b[1], k[1]))
            union = (b[2] - b[0]) * (b[3] - b[1]) + (k[2] - k[0]) * (k[3] - k[1]) - iw * ih
            if union > 0 and iw * ih / union >= max_iou:
                ok = False
        if ok:
            kept.append([int(v) for v in b])
    return kept


def _sparse_matrix(seed, n, m, density, diag):
    """mostly below min_sim, `density` of the cells matching, optionally a planted diagonal run"""
    rng = np.random.default_rng(seed)
    sims = (0.15 * rng.random((n, m))).astype(np.float32)
    hot = rng.random((n, m)) < density
    sims[hot] = (0.3 + 0.6 * rng.random(int(hot.sum()))).astype(np.float32)
    if diag:
        q0, r0 = int(rng.integers(0, max(1, n - 3))), int(rng.integers(0, max(1, m - 3)))
        for k in range(min(n - q0, m - r0, int(rng.integers(3, 8)))):
            sims[q0 + k, r0 + k] = np.float32(0.5 + 0.4 * rng.random())
    return sims


def _bf_dp(sims, discontinue, min_sim, min_length, max_iou, max_path):
    n, m = sims.shape
    alive = {(i, j) for i in range(n) for j in range(m) if sims[i, j] >= min_sim}
    reach = discontinue + 1
    boxes = []
    for _ in range(max_path):
        if not alive:
            break
        cells = sorted(alive)
        best = (-1.0, None)

        def extend(chain, score):
            nonlocal best
            if score > best[0] + 1e-12:
                best = (score, list(chain))
            li, lj = chain[-1]
            for (i, j) in cells:
                if 1 <= i - li <= reach and 1 <= j - lj <= reach:
                    chain.append((i, j))
                    extend(chain, score + float(sims[i, j]) - min_sim)
                    chain.pop()

        for c in cells:
            extend([c], float(sims[c]) - min_sim)
        chain = best[1]
        qs, rs = [c[0] for c in chain], [c[1] for c in chain]
        boxes.append([min(qs), min(rs), max(qs), max(rs)])
        alive = {(i, j) for (i, j) in alive if not (min(qs) <= i <= max(qs)) and not (min(rs) <= j <= max(rs))}
    return _bf_keep(boxes, min_length, max_iou)


@settings(max_examples=80, deadline=None, derandomize=True)
@given(st.integers(0, 10**6), st.integers(1, 9), st.integers(1, 9), st.sampled_from([0, 1, 3]), st.booleans())
def test_dp_against_exhaustive_chain_enumeration(seed, n, m, discontinue, diag):
    sims = _sparse_matrix(seed, n, m, 0.12, diag)
    if int((sims >= 0.2).sum()) > 14:      # (the enumeration is exponential in the matching cells)
        return
    for min_length in (0, 2):
        got = dp(sims, discontinue=discontinue, min_sim=0.2, min_length=min_length, max_iou=0.3, max_path=4)
        assert got == _bf_dp(sims, discontinue, 0.2, min_length, 0.3, 4), (seed, n, m, discontinue)


def _bf_dtw(sims, discontinue, min_sim, min_length, max_iou):
    n, m = sims.shape
    best = (np.inf, None)

    def walk(path, cost):
        nonlocal best
        i, j = path[-1]
        if (i, j) == (n - 1, m - 1):
            if cost < best[0] - 1e-12:
                best = (cost, list(path))
            return
        for di, dj in ((1, 1), (1, 0), (0, 1)):
            a, b = i + di, j + dj
            if a < n and b < m:
                path.append((a, b))
                walk(path, cost + 1.0 - float(sims[a, b]))
                path.pop()

    walk([(0, 0)], 1.0 - float(sims[0, 0]))
    path = best[1]
    runs, cur, misses = [], [], 0
    for cell in path:
        if sims[cell] >= min_sim:
            cur.append(cell)
            misses = 0
        elif cur:
            misses += 1
            if misses > discontinue:
                runs.append(cur)
                cur = []
    if cur:
        runs.append(cur)
    # (a run's pending misses never end it at the path's end: it ends on its last matching cell either way)
    scored = sorted(range(len(runs)), key=lambda k: (-sum(float(sims[c]) for c in runs[k]), k))
    boxes = []
    for k in scored:
        qs, rs = [c[0] for c in runs[k]], [c[1] for c in runs[k]]
        boxes.append([min(qs), min(rs), max(qs), max(rs)])
    return _bf_keep(boxes, min_length, max_iou)


@settings(max_examples=80, deadline=None, derandomize=True)
@given(st.integers(0, 10**6), st.integers(1, 6), st.integers(1, 6), st.sampled_from([0, 1, 3]), st.booleans())
def test_dtw_against_exhaustive_path_enumeration(seed, n, m, discontinue, diag):
    sims = _sparse_matrix(seed, n, m, 0.3, diag)
    for min_length in (0, 1):
        got = dtw(sims, discontinue=discontinue, min_sim=0.2, min_length=min_length, max_iou=0.3)
        assert got == _bf_dtw(sims, discontinue, 0.2, min_length, 0.3), (seed, n, m, discontinue)


def _bf_hv(sims, discontinue, min_sim, min_length, max_iou, max_path, tolerance):
    n, m = sims.shape
    cells = [(q, r) for q in range(n) for r in range(m) if sims[q, r] >= min_sim]
    if not cells:
        return []
    votes = {}
    for q, r in cells:
        votes[r - q] = votes.get(r - q, 0.0) + float(sims[q, r])
    H = {d: sum(votes.get(d + t, 0.0) for t in range(-tolerance, tolerance + 1)) for d in range(-(n - 1), m)}
    peaks = sorted(H, key=lambda d: (-H[d], abs(d), d))[:max_path]
    runs = []
    for rank, d in enumerate(peaks):
        if not H[d] > 0:
            break
        band = [(q, r) for (q, r) in cells if abs(r - q - d) <= tolerance]
        cur, count = [], 0
        for c in band:
            if cur and c[0] - cur[-1][0] > discontinue + 1:
                runs.append((sum(float(sims[x]) for x in cur), rank, count, cur))
                cur, count = [], count + 1
            cur.append(c)
        if cur:
            runs.append((sum(float(sims[x]) for x in cur), rank, count, cur))
    runs.sort(key=lambda r: (-r[0], r[1], r[2]))
    boxes = [[min(c[0] for c in cs), min(c[1] for c in cs), max(c[0] for c in cs), max(c[1] for c in cs)] for _, _, _, cs in runs]
    return _bf_keep(boxes, min_length, max_iou)


@settings(max_examples=120, deadline=None, derandomize=True)
@given(st.integers(0, 10**6), st.integers(1, 14), st.integers(1, 14), st.sampled_from([0, 1, 3]), st.sampled_from([0, 1, 2]),
       st.booleans())
def test_hv_against_dictionary_restatement(seed, n, m, discontinue, tolerance, diag):
    sims = _sparse_matrix(seed, n, m, 0.15, diag)
    for min_length in (0, 2):
        got = hv(sims, discontinue=discontinue, min_sim=0.2, min_length=min_length, max_iou=0.3, max_path=5, tolerance=tolerance)
        assert got == _bf_hv(sims, discontinue, 0.2, min_length, 0.3, 5, tolerance), (seed, n, m, discontinue, tolerance)


def test_hv_votes_by_offset_and_cuts_runs():
    sims = np.zeros((40, 50), np.float32)
    for k in range(10):
        sims[5 + k, 8 + k] = 0.9             # offset +3, ten frames
    for k in range(9):
        sims[25 + k, 20 + k] = 0.8           # offset -5, nine frames
    sims[38, 41] = 0.95                      # a lone cell on offset +3, far from the run (cut: 23 empty query frames)
    assert hv(sims, min_length=5) == [[5, 8, 14, 17], [25, 20, 33, 28]]
    assert hv(sims, min_length=5, max_path=1) == [[5, 8, 14, 17]]                  # only the strongest offset
    # a re-timed copy (offset drifts by one) stays ONE run inside the tolerance band, and falls apart without it
    drift = np.zeros((30, 30), np.float32)
    for k in range(12):
        drift[4 + k, 6 + k + (1 if k >= 6 else 0)] = 0.7
    assert hv(drift, min_length=5, tolerance=1) == [[4, 6, 15, 18]]
    assert hv(drift, min_length=5, tolerance=0) == []                               # two 6-frame pieces: min(dq, dr) = 5
    assert hv(np.full((6, 6), 0.1, np.float32)) == [] and hv(np.zeros((0, 4), np.float32)) == []


def test_spd_is_declared_not_buildable():
    with pytest.raises(NotImplementedError, match="TRAINED detector network"):
        build_vta_model("SPD")
