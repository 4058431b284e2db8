"""CPU: the DTW / DP aligners behind `vcsl.vta.build_vta_model` (vsc2022_amd/vcsl/aligners.py).
Parity with VCSL's own implementations is unpinned (their source is not in the reference checkout, like TN's): these
tests pin the `forward_sim` contract the reference relies on (vsc/baseline/localization.py:58-66), the properties of the
reference's localisation tests (tests/test_localization.py:46-66: a planted copy is found, unrelated videos yield
nothing), and the documented semantics on small hand-checked matrices."""
import numpy as np
import pytest

from vsc2022_amd.vcsl.aligners import dp, dtw, dtw_path
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


@pytest.mark.parametrize("name", ["DTW", "DP"])
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
