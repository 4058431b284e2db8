"""GPU parity: similarity / search kernels vs the CPU oracle, bit-exact (integer + fp32 bit patterns)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("nq,nr,d", [(70, 300, 512), (33, 129, 64), (5, 7, 3), (130, 260, 100), (1, 1, 512)])
def test_scores_bit_exact_via_range_search(gpu, orc, nq, nr, d):
    """Every similarity equals the oracle's ascending-k fp32 fma chain, bit for bit."""
    from vsc2022_amd.vsc.index import FlatIndex

    rng = np.random.default_rng(nq * 1000 + nr)
    q, r = unit(rng, nq, d), unit(rng, nr, d)
    idx = FlatIndex(d)
    idx.add(r)
    lims, D, I = idx.range_search(q, -1e30)
    assert lims[-1] == nq * nr
    ref = orc.scores(q, r)
    assert np.array_equal(I.reshape(nq, nr), np.tile(np.arange(nr), (nq, 1)))
    assert np.array_equal(bits(D.reshape(nq, nr)), bits(ref))


def test_range_search_matches_oracle(gpu, orc):
    from vsc2022_amd.vsc.index import FlatIndex

    rng = np.random.default_rng(3)
    q, r = unit(rng, 200, 128), unit(rng, 1000, 128)
    idx = FlatIndex(128)
    idx.add(r[:400])
    idx.add(r[400:])  # incremental add
    lims, D, I = idx.range_search(q, 0.2)
    olims, oD, oI = orc.range_search(q, r, 0.2)
    assert np.array_equal(lims, olims)
    assert np.array_equal(I, oI)
    assert np.array_equal(bits(D), bits(oD))


def _check_topk(orc, q, r, K, metric=0):
    from vsc2022_amd.vsc.index import FlatIndex

    idx = FlatIndex(q.shape[1], metric)
    idx.add(r)
    i, j, s, radius = idx.global_topk(q, K)
    oi, oj, os_, info = orc.global_threshold_search(q, r, K, metric, return_info=True)
    assert len(s) == len(os_), (len(s), len(os_))
    assert np.array_equal(i, oi) and np.array_equal(j, oj)
    assert np.array_equal(bits(s), bits(os_))
    # (VSC_TOPK_SHORTCUT=2, tests/test_gpu_topk_proven.py: the proven route returns the steady run's radius, not the schedule's)
    assert idx.get_option("last_topk_route") == 1 or np.float32(radius) == np.float32(info["radius"])
    return info


@pytest.mark.parametrize("seed,nq,nr,d,K", [
    (0, 300, 900, 512, 5000),     # several re-thresholds (32+64+128+76 rows)
    (1, 1000, 1000, 64, 60000),   # config-1 shape
    (2, 40, 50, 32, 1),
    (3, 40, 50, 32, 10 ** 6),     # K larger than the matrix: everything comes back
    (4, 700, 333, 100, 1234),
    (5, 2100, 257, 16, 9000),
])
def test_global_topk_matches_oracle(gpu, orc, seed, nq, nr, d, K):
    rng = np.random.default_rng(seed)
    info = _check_topk(orc, unit(rng, nq, d), unit(rng, nr, d), K)
    if seed == 0:
        assert info["n_rethreshold"] >= 2


def test_global_topk_with_exact_ties(gpu, orc):
    """Static videos (duplicate frames) put exact ties on the re-threshold cuts; the reference
    drops every hit tied with the new radius, and so must the engine."""
    rng = np.random.default_rng(7)
    d = 64
    base_q, base_r = unit(rng, 60, d), unit(rng, 80, d)
    q = np.repeat(base_q, 5, axis=0)      # 300 rows, 5 identical copies each
    r = np.repeat(base_r, 4, axis=0)      # 320 rows
    for K in (100, 777, 5000, 20000):
        _check_topk(orc, q, r, K)
    # quantised descriptors: massive tie groups
    qq = np.round(unit(rng, 200, 8) * 2) / 2
    rr = np.round(unit(rng, 300, 8) * 2) / 2
    for K in (50, 500, 5000):
        _check_topk(orc, qq.astype(np.float32), rr.astype(np.float32), K)


def test_global_topk_l2(gpu, orc):
    rng = np.random.default_rng(11)
    q, r = unit(rng, 150, 24), unit(rng, 170, 24)
    for K in (1, 300, 4000):
        _check_topk(orc, q, r, K, metric=1)
    # tests/test_index.py fixture of the reference: 9 exact-zero distances tie at the cut -> 0 hits
    f = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9], [11, 12, 13], [14, 15, 16], [17, 18, 19],
                  [111, 112, 113], [114, 115, 116], [117, 118, 119]], dtype=np.float32)
    info = _check_topk(orc, f, f, 1, metric=1)


@pytest.mark.parametrize("nq,nr,d,k", [(300, 5000, 512, 20), (100, 700, 64, 1), (257, 130, 32, 5),
                                      (10, 3, 16, 5), (1000, 40000, 128, 10)])
def test_knn_matches_oracle(gpu, orc, nq, nr, d, k):
    from vsc2022_amd.vsc.index import FlatIndex

    rng = np.random.default_rng(nq + nr + k)
    q, r = unit(rng, nq, d), unit(rng, nr, d)
    idx = FlatIndex(d)
    idx.add(r)
    D, I = idx.search(q, k)
    oD, oI = orc.knn(q, r, k)
    assert np.array_equal(I, oI)
    assert np.array_equal(bits(D), bits(oD))


def test_knn_ties_and_l2(gpu, orc):
    from vsc2022_amd.vsc.index import FlatIndex

    rng = np.random.default_rng(5)
    r = np.repeat(unit(rng, 50, 32), 6, axis=0)  # 6 identical copies of every ref
    q = unit(rng, 90, 32)
    idx = FlatIndex(32)
    idx.add(r)
    D, I = idx.search(q, 8)
    oD, oI = orc.knn(q, r, 8)
    assert np.array_equal(I, oI) and np.array_equal(bits(D), bits(oD))
    idx2 = FlatIndex(32, 1)
    idx2.add(r)
    D, I = idx2.search(q, 4)
    oD, oI = orc.knn(q, r, 4, 1)
    assert np.array_equal(I, oI) and np.array_equal(bits(D), bits(oD))


def test_pair_max_matches_oracle(gpu, orc):
    import ctypes
    from vsc2022_amd import _lib
    from vsc2022_amd.vsc.index import FlatIndex

    rng = np.random.default_rng(9)
    nqv, nrv = 30, 40
    qlen = rng.integers(3, 20, nqv)
    rlen = rng.integers(3, 20, nrv)
    row2q = np.repeat(np.arange(nqv, dtype=np.int32), qlen)
    row2r = np.repeat(np.arange(nrv, dtype=np.int32), rlen)
    q, r = unit(rng, len(row2q), 48), unit(rng, len(row2r), 48)
    # duplicate a few rows so that pairs tie on their max score
    r[5] = r[40]
    r[6] = r[41]
    idx = FlatIndex(48)
    idx.add(r)
    i, j, s, _ = idx.global_topk(q, 3000)
    n = len(s)
    oq, orr, os_, of = orc.pair_max(i, j, s, row2q, row2r)
    gq, gr, gs = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.float32)
    gf = np.empty(n, np.int64)
    npairs = ctypes.c_int64(0)
    _lib.check(_lib.lib().vsc_pair_max(i.ctypes.data, j.ctypes.data, s.ctypes.data, n, 0, row2q.ctypes.data,
                                       len(row2q), row2r.ctypes.data, len(row2r), 0, gq.ctypes.data,
                                       gr.ctypes.data, gs.ctypes.data, gf.ctypes.data, n, 0,
                                       ctypes.byref(npairs), 0))
    m = npairs.value
    assert m == len(oq)
    assert np.array_equal(gq[:m], oq) and np.array_equal(gr[:m], orr)
    assert np.array_equal(bits(gs[:m]), bits(os_)) and np.array_equal(gf[:m], of)


def test_row_normalize(gpu, orc):
    from vsc2022_amd.vsc.baseline.score_normalization import normalize

    rng = np.random.default_rng(2)
    x = rng.standard_normal((777, 511)).astype(np.float32)
    x[13] = 0.0
    out = normalize(x)
    assert np.array_equal(bits(out), bits(orc.row_normalize(x)))
    assert np.all(out[13] == 0)
    nz = np.delete(np.arange(777), 13)
    assert np.allclose(np.linalg.norm(out[nz].astype(np.float64), axis=1), 1.0, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("metric", ["ip", "l2"])
def test_seeded_global_topk_matches_the_exact_topk(gpu, orc, metric):
    """`vsc_index_global_topk_seeded` (the local search of the query-sharded pipeline): from a radius below the K-th
    best score the steady-batch search returns the exact top-K -- what the reference's schedule returns too when no hit
    ties with a re-threshold radius --; from a radius above it, every hit strictly beyond that radius and nothing else;
    a non-finite seed is refused."""
    from vsc2022_amd import _lib
    from vsc2022_amd.vsc.index import FlatIndex

    rng = np.random.default_rng(77)
    nq, nr, d, K = 3000, 40000, 96, 50000
    q = rng.standard_normal((nq, d)).astype(np.float32)
    r = rng.standard_normal((nr, d)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    r /= np.linalg.norm(r, axis=1, keepdims=True)
    l2 = metric == "l2"
    idx = FlatIndex(d, _lib.METRIC_L2 if l2 else _lib.METRIC_INNER_PRODUCT)
    idx.add(r)
    oi, oj, os_ = orc.global_threshold_search(q, r, K, 1 if l2 else 0)
    kth = float(os_[-1])
    # a seed comfortably on the easy side of the K-th best (scores: larger is better; L2 distances: smaller)
    seed = kth + 0.02 if l2 else kth - 0.02
    i, j, s, rad = idx.global_topk(q, K, seed_radius=seed)
    assert np.array_equal(i, oi) and np.array_equal(j, oj) and np.array_equal(s.view(np.uint32), os_.view(np.uint32))
    assert (rad >= seed) if not l2 else (rad <= seed)
    # a seed on the wrong side: only the hits strictly beyond it, in order
    cut = int(K * 0.6)
    seed_hi = float(os_[cut])
    i2, j2, s2, rad2 = idx.global_topk(q, K, seed_radius=seed_hi)
    beyond = (os_ < np.float32(seed_hi)) if l2 else (os_ > np.float32(seed_hi))
    n_exp = int(np.count_nonzero(beyond))
    assert 0 < len(s2) == n_exp < K
    assert np.array_equal(i2, oi[:n_exp]) and np.array_equal(j2, oj[:n_exp])
    assert np.array_equal(s2.view(np.uint32), os_[:n_exp].view(np.uint32))
    with pytest.raises(ValueError):
        idx.global_topk(q, K, seed_radius=float("nan"))
