"""int8 pre-filter (csrc/sim_i8p.hip + quant_i8.hip) must be INVISIBLE in the results, like the fp16 one.

Batches whose hits are sparse run on v_mfma_i32_16x16x64_i8 over 8-bit images of the rows (one scale per
reference row, one per 128-row query panel); a pair goes to the exact fp32 stage when its integer score exceeds
a rigorous lower bound of (radius - eps) / (s_q s_r), eps built from the quantisation residuals that were
actually produced.  Hits, order and fp32 bit patterns must equal the CPU oracle's (vsc/index.py:142-165 and
:167-177 semantics) with the int8 kernel forced onto every pre-filtered batch (VSC_PREFILTER=2 VSC_I8=2),
chosen by the density rule, or switched off (VSC_I8=0).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


def opts(**kv):
    """{VSC_X: value-or-None} in the spelling of the environment switches -> the handle's options (set while it is empty:
    `FlatIndex(d, options=...)` -> vsc_index_set_option; no test mutates os.environ around a handle's creation)."""
    return {k[4:].lower(): float(v) for k, v in kv.items() if v is not None}


def forced_index(d):
    from vsc2022_amd.vsc.index import FlatIndex

    return FlatIndex(d, options=opts(VSC_PREFILTER="2", VSC_I8="2"))


def i8_launches(idx):
    return idx.profile_read(reset=True)["i8_launches"]


def assert_same(a, b):
    assert len(a[2]) == len(b[2]), (len(a[2]), len(b[2]))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.array_equal(bits(a[2]), bits(b[2]))


def test_quantisation_bound_is_a_bound():
    """The inequality the kernel relies on, checked in float64 on the quantisation the kernel applies:
    |x.y - s_x s_y (q_x . q_y)| <= E_x N_y + (N_x + E_x) E_y with E = ||x - s q||, N = ||x||."""
    rng = np.random.default_rng(0)
    for d, scale in ((512, 1.0), (100, 30.0), (768, 1e-4)):
        x = (rng.standard_normal((200, d)) * scale * rng.uniform(0.2, 3.0, (200, 1))).astype(np.float32)
        y = (rng.standard_normal((300, d)) * scale).astype(np.float32)
        sx = np.float32(np.abs(x).max() / 127.0)                     # one scale for the whole query panel
        sy = (np.abs(y).max(1, keepdims=True) / 127.0).astype(np.float32)
        qx = np.clip(np.rint(x / sx), -127, 127)
        qy = np.clip(np.rint(y / sy), -127, 127)
        x64, y64 = x.astype(np.float64), y.astype(np.float64)
        ex = np.linalg.norm(x64 - np.float64(sx) * qx, axis=1)
        ey = np.linalg.norm(y64 - sy.astype(np.float64) * qy, axis=1)
        nx, ny = np.linalg.norm(x64, axis=1), np.linalg.norm(y64, axis=1)
        err = np.abs(x64 @ y64.T - (np.float64(sx) * sy.astype(np.float64).T) * (qx @ qy.T))
        bound = ex[:, None] * ny[None, :] + (nx + ex)[:, None] * ey[None, :]
        assert (err <= bound * (1 + 1e-12)).all()
        assert err.max() > 0.02 * bound.min()                         # and the bound is not absurdly loose


@pytest.mark.gpu
@pytest.mark.parametrize("nq,nr,d,K", [(700, 3000, 128, 900), (300, 1000, 512, 200), (1100, 5000, 40, 3000),
                                        (64, 257, 100, 50), (260, 1500, 768, 700), (200, 900, 1000, 400)])
def test_topk_with_forced_int8_matches_oracle(gpu, orc, nq, nr, d, K):
    rng = np.random.default_rng(nq + nr + d)
    q, r = unit(rng, nq, d), unit(rng, nr, d)
    idx = forced_index(d)
    idx.profile(True)
    idx.add(r[: nr // 3 + 7])      # incremental adds that end inside a 64-row tile of the int8 image
    idx.add(r[nr // 3 + 7:])
    i, j, s, radius = idx.global_topk(q, K)
    oi, oj, os_, info = orc.global_threshold_search(q, r, K, 0, return_info=True)
    assert_same((i, j, s), (oi, oj, os_))
    assert np.float32(radius) == np.float32(info["radius"])
    assert i8_launches(idx) > 0    # the int8 kernel really ran


@pytest.mark.gpu
def test_int8_with_ties_duplicates_and_extreme_rows(gpu, orc):
    """Exact score ties on the re-threshold values; rows of very different norms inside one panel / tile; rows 8 bits
    cannot tell from zero next to large ones; inf / NaN rows; an all-zero row."""
    rng = np.random.default_rng(31)
    r = unit(rng, 2000, 64)
    r[100:400] = r[100]
    q = unit(rng, 600, 64)
    q[50:120] = q[50]
    q[300:330] = r[100]
    idx = forced_index(64)
    idx.add(r)
    assert_same(idx.global_topk(q, 1500)[:3], orc.global_threshold_search(q, r, 1500))

    q = rng.standard_normal((400, 96)).astype(np.float32) * rng.uniform(0.01, 30.0, (400, 1)).astype(np.float32)
    r = rng.standard_normal((1500, 96)).astype(np.float32) * rng.uniform(0.01, 30.0, (1500, 1)).astype(np.float32)
    r[7] *= 1e-6
    r[8, 3] = 1e6
    q[9, 0] = 7e4
    r[10, 5] = np.inf
    q[11, 2] = np.nan
    r[12] = 0.0
    q[13] = 0.0
    r[14] *= 1e-30                 # scale products far outside the normal range of their inverse
    q[15] *= 1e-25
    r[16] *= 1e25
    idx = forced_index(96)
    idx.add(r)
    assert_same(idx.global_topk(q, 2500)[:3], orc.global_threshold_search(q, r, 2500))
    idx = forced_index(96)         # the same with every row tiny: the radius sits among scores ~1e-50 (flushed to 0)
    idx.add(r * np.float32(1e-20))
    assert_same(idx.global_topk(q[16:] * np.float32(1e-20), 2500)[:3],
                orc.global_threshold_search(q[16:] * np.float32(1e-20), r * np.float32(1e-20), 2500))


@pytest.mark.gpu
@pytest.mark.parametrize("nq,nr,d,k", [(300, 5000, 128, 1), (257, 3000, 512, 20), (64, 200, 40, 64), (1000, 9000, 64, 5),
                                        (130, 2000, 768, 3)])
def test_knn_with_forced_int8_matches_oracle(gpu, orc, nq, nr, d, k):
    rng = np.random.default_rng(nq + nr + k)
    q, r = unit(rng, nq, d), unit(rng, nr, d)
    r[10:40] = r[10]               # ties: equal scores must come out in ascending ref order
    q[5] = r[10]
    idx = forced_index(d)
    idx.profile(True)
    idx.add(r)
    D, I = idx.search(q, k)
    oD, oI = orc.knn(q, r, k)
    assert np.array_equal(I, oI)
    assert np.array_equal(bits(D), bits(oD))
    assert i8_launches(idx) > 0


@pytest.mark.gpu
def test_range_search_with_forced_int8(gpu, orc):
    rng = np.random.default_rng(33)
    q, r = unit(rng, 500, 128), unit(rng, 2100, 128)
    idx = forced_index(128)
    idx.add(r[:900])
    idx.add(r[900:])
    lims, D, I = idx.range_search(q, 0.25)
    olims, oD, oI = orc.range_search(q, r, 0.25)
    assert np.array_equal(lims, olims) and np.array_equal(I, oI)
    assert np.array_equal(bits(D), bits(oD))


@pytest.mark.gpu
@pytest.mark.parametrize("seed,nq,nr,d,K", [(41, 3000, 20000, 64, 150000), (42, 2500, 33000, 128, 40000)])
def test_int8_fast_emission_path_matches_oracle(gpu, orc, seed, nq, nr, d, K):
    """emit_candidates_seg of sim_i8p.hip runs while a whole tile still fits the wave's segment (large hit buffers)."""
    rng = np.random.default_rng(seed)
    q, r = unit(rng, nq, d), unit(rng, nr, d)
    for row in rng.choice(nq, 60, replace=False):
        tgt = rng.choice(nr, 300, replace=False)
        r[tgt] = q[row] + 0.25 * rng.standard_normal((300, d)).astype(np.float32)
        r[tgt] /= np.linalg.norm(r[tgt], axis=1, keepdims=True)
    idx = forced_index(d)
    idx.set_hit_capacity(24_000_000)
    idx.add(r)
    i, j, s, radius = idx.global_topk(q, K)
    oi, oj, os_, info = orc.global_threshold_search(q, r, K, 0, return_info=True)
    assert_same((i, j, s), (oi, oj, os_))
    assert np.float32(radius) == np.float32(info["radius"])


@pytest.mark.gpu
def test_int8_chosen_by_density_equals_fp16_and_fp32_routes(gpu):
    """Large enough for the density rule to pick the int8 kernel for the late batches; the three device routes must
    agree bit for bit (hits, order, radius)."""
    from vsc2022_amd.vsc.index import FlatIndex

    rng = np.random.default_rng(35)
    q, r = unit(rng, 70000, 128), unit(rng, 120000, 128)
    K = 300000
    outs = []
    for kv in (dict(VSC_I8=None, VSC_PREFILTER=None), dict(VSC_I8="0", VSC_PREFILTER=None), dict(VSC_I8="0", VSC_PREFILTER="0")):
        idx = FlatIndex(128, options=opts(**kv))
        idx.profile(True)
        idx.add(r)
        out = idx.global_topk(q, K)
        outs.append((out, idx.profile_read(reset=True)))
    assert outs[0][1]["i8_launches"] > 0 and outs[1][1]["i8_launches"] == 0 and outs[1][1]["f16_launches"] > 0
    assert outs[2][1]["f16_launches"] == 0
    for other in outs[1:]:
        assert_same(outs[0][0][:3], other[0][:3])
        assert outs[0][0][3] == other[0][3]


def score_normalised_like(rng, nq, nr, d):
    """Descriptors shaped like the output of score_normalize (vsc/baseline/score_normalization.py:99-104): unit rows
    in d - 1 coordinates, the last coordinate = 1 for every reference and a per-row penalty for the queries."""
    q = np.concatenate([unit(rng, nq, d - 1), -rng.uniform(0.15, 0.45, (nq, 1)).astype(np.float32)], axis=1)
    r = np.concatenate([unit(rng, nr, d - 1), np.ones((nr, 1), np.float32)], axis=1)
    return np.ascontiguousarray(q), np.ascontiguousarray(r)


@pytest.mark.gpu
@pytest.mark.parametrize("nq,nr,d,K,k", [(700, 3000, 129, 900, 5), (300, 1500, 512, 400, 1), (1100, 5000, 65, 3000, 20)])
def test_constant_reference_coordinate_is_kept_out_of_the_int8_image(gpu, orc, nq, nr, d, K, k):
    """All references agree on the last coordinate (score-normalised descriptors): it is excluded from the 8-bit
    images and enters through per-row thresholds (rows of a launch sorted by threshold).  Forced int8, thresholded
    search and k-NN against the oracle."""
    rng = np.random.default_rng(nq + d)
    q, r = score_normalised_like(rng, nq, nr, d)
    q[7] = q[8]
    r[30:50] = r[30]
    idx = forced_index(d)
    idx.profile(True)
    idx.add(r[: nr // 2 + 3])
    idx.add(r[nr // 2 + 3:])
    i, j, s, radius = idx.global_topk(q, K)
    oi, oj, os_, info = orc.global_threshold_search(q, r, K, 0, return_info=True)
    assert_same((i, j, s), (oi, oj, os_))
    assert np.float32(radius) == np.float32(info["radius"])
    D, I = idx.search(q, k)
    oD, oI = orc.knn(q, r, k)
    assert np.array_equal(I, oI) and np.array_equal(bits(D), bits(oD))
    assert i8_launches(idx) > 0


@pytest.mark.gpu
def test_excluded_coordinates_follow_the_rows_that_are_added(gpu, orc):
    """The set of coordinates on which ALL references agree can only shrink as rows arrive: the image is rewritten
    before the next search when it does.  More agreeing coordinates than the 8 the kernels handle; a huge one; a NaN
    in a query's excluded coordinate."""
    rng = np.random.default_rng(77)
    d = 96
    q, r = unit(rng, 500, d), unit(rng, 2600, d)
    const = {3: 0.7, 10: -1.5, 11: 2.0, 20: 0.25, 21: 0.25, 22: 0.25, 23: 0.25, 24: 0.25, 25: 0.25, 26: 0.25, 40: 30.0}
    for c, v in const.items():
        r[:, c] = v
    r[1800:, 10] = 0.3          # the third add breaks the agreement on coordinate 10 ...
    r[1800:, 40] = -2.0         # ... and on the largest one
    q[:, 40] *= 0.05
    idx = forced_index(d)
    idx.profile(True)
    for lo, hi, K in ((0, 900, 700), (900, 1800, 1500), (1800, 2600, 2500)):
        idx.add(r[lo:hi])
        got = idx.global_topk(q, K)
        assert_same(got[:3], orc.global_threshold_search(q, r[:hi], K))
        D, I = idx.search(q, 3)
        oD, oI = orc.knn(q, r[:hi], 3)
        assert np.array_equal(I, oI) and np.array_equal(bits(D), bits(oD))
    assert i8_launches(idx) > 0
    q2 = q.copy()
    q2[5, 3] = np.nan           # excluded coordinate of a query row: its threshold is NaN, every pair must pass
    q2[6, 11] = np.inf
    assert_same(idx.global_topk(q2, 1200)[:3], orc.global_threshold_search(q2, r, 1200))


@pytest.mark.gpu
def test_int8_is_chosen_for_score_normalised_descriptors(gpu):
    """Without the exclusion the looseness gate keeps score-normalised references off the int8 kernel (one
    coordinate sets the scale of the row); with it the density rule picks int8 for the late batches, and the three
    device routes agree bit for bit."""
    from vsc2022_amd.vsc.index import FlatIndex

    rng = np.random.default_rng(36)
    q, r = score_normalised_like(rng, 70000, 120000, 128)
    K = 300000
    outs = []
    for kv in (dict(VSC_I8=None, VSC_PREFILTER=None, VSC_I8_EXCLUDE=None), dict(VSC_I8="0", VSC_PREFILTER=None, VSC_I8_EXCLUDE=None),
               dict(VSC_I8="0", VSC_PREFILTER="0", VSC_I8_EXCLUDE=None)):
        idx = FlatIndex(128, options=opts(**kv))
        idx.profile(True)
        idx.add(r)
        out = idx.global_topk(q, K)
        outs.append((out, idx.profile_read(reset=True)))
    assert outs[0][1]["i8_launches"] > 0 and outs[1][1]["i8_launches"] == 0 and outs[1][1]["f16_launches"] > 0
    for other in outs[1:]:
        assert_same(outs[0][0][:3], other[0][:3])
        assert outs[0][0][3] == other[0][3]


def shifted(rng, n, d, mean_cos, dominant=0, geometry_seed=3):
    """unit rows with a common direction: two random rows at cosine ~ mean_cos (uncentred embeddings); `dominant`
    coordinates at 4 x the scale.  Direction and dominant coordinates come from `geometry_seed`: shared by both sides."""
    geo = np.random.default_rng(geometry_seed)
    mu = unit(geo, 1, d)[0] * np.sqrt(d * mean_cos / (1.0 - mean_cos))
    x = rng.standard_normal((n, d)).astype(np.float32) + mu.astype(np.float32)
    if dominant:
        x[:, geo.permutation(d)[:dominant]] *= 4.0
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("nq,nr,d,K,k,cos,centre", [(700, 3000, 128, 900, 5, 0.3, 2), (300, 2500, 512, 400, 1, 0.5, 1),
                                                  (1100, 5000, 65, 3000, 20, 0.1, 2), (900, 4000, 256, 50000, 3, 0.0, 2)])
def test_centred_int8_image_is_invisible_in_the_results(gpu, orc, nq, nr, d, K, k, cos, centre):
    """quant_i8.hip "CENTRED references": the int8 image holds y - mu, the rows' x . mu moves their thresholds.  Forced
    onto every pre-filtered batch (centre = 2: always; 1: decided from the mean's share of the energy), rows added in two
    pieces (the centre is fixed at the first catch-up that sees >= 1024 rows), top-K / k-NN / range search against the oracle bit for bit."""
    from vsc2022_amd.vsc.index import FlatIndex

    rng = np.random.default_rng(nq + d)
    q, r = shifted(rng, nq, d, cos, 2), shifted(rng, nr, d, cos, 2)
    r[40:70] = r[40]                     # exact ties
    q[5] = r[40]
    idx = FlatIndex(d, options=opts(VSC_PREFILTER="2", VSC_I8="2", VSC_I8_CENTER=str(centre)))
    idx.profile(True)
    idx.add(r[: nr // 3])
    top = idx.global_topk(q, K)          # (first search, on a third of the rows)
    assert_same(top[:3], orc.global_threshold_search(q, r[: nr // 3], K))
    idx.add(r[nr // 3 :])
    top = idx.global_topk(q, K)
    assert_same(top[:3], orc.global_threshold_search(q, r, K))
    # (centre = 1: decided at the first catch-up over >= 1024 rows -- here the second one; the rows written before are rewritten)
    assert idx.get_option("i8_center_on") == (1.0 if (centre == 2 or cos >= 0.05) else 0.0)
    D, I = idx.search(q, k)
    Do, Io = orc.knn(q, r, k)
    assert np.array_equal(I, Io) and np.array_equal(bits(D), bits(Do))
    radius = float(np.sort(top[2])[len(top[2]) // 2]) if len(top[2]) else 0.5
    lims, Dr, Ir = idx.range_search(q, radius)
    ol, oD, oI = orc.range_search(q, r, radius)
    assert np.array_equal(lims, ol) and np.array_equal(Ir, oI) and np.array_equal(bits(Dr), bits(oD))
    assert i8_launches(idx) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("spread", [0.0, 1e-6, 1e-4, 1e-2])
def test_centred_image_of_nearly_identical_references(gpu, orc, spread):
    """The hard case for the decomposition x.y = x.(y - mu) + x.mu: references that (nearly) equal their mean.  y - mu is
    then all rounding (spread 0: exactly zero -- an empty image, E = N' = 0), every score is x.mu up to a few ulps and the
    search's radius sits inside that cloud: what decides a pair is the slack for the roundings of x.mu (a tree of fmas
    here, the ascending chain in the exact stage) and of y - mu.  Top-K, 3-NN and a range search against the oracle."""
    from vsc2022_amd.vsc.index import FlatIndex

    rng = np.random.default_rng(17)
    d, nq, nr = 200, 500, 3000
    centre = unit(rng, 1, d)
    r = np.repeat(centre, nr, axis=0) + np.float32(spread) * rng.standard_normal((nr, d)).astype(np.float32)
    r = np.ascontiguousarray(r.astype(np.float32))
    q = shifted(rng, nq, d, 0.3)
    idx = FlatIndex(d, options=opts(VSC_PREFILTER="2", VSC_I8="2", VSC_I8_CENTER="2"))
    idx.profile(True)
    idx.add(r)
    for K in (100, 5000, 400000):
        assert_same(idx.global_topk(q, K)[:3], orc.global_threshold_search(q, r, K))
    D, I = idx.search(q, 3)
    Do, Io = orc.knn(q, r, 3)
    assert np.array_equal(I, Io) and np.array_equal(bits(D), bits(Do))
    S = orc.scores(q[:64], r)
    radius = float(np.median(S))
    lims, Dr, Ir = idx.range_search(q[:64], radius)
    ol, oD, oI = orc.range_search(q[:64], r, radius)
    assert np.array_equal(lims, ol) and np.array_equal(Ir, oI) and np.array_equal(bits(Dr), bits(oD))
    assert idx.get_option("i8_center_on") == 1.0 and i8_launches(idx) > 0


@pytest.mark.gpu
def test_centring_and_excluded_coordinates_together(gpu, orc):
    """Score-normalised rows WITH a common direction: one coordinate is 1 on every reference (left out of the image, its
    contribution in the rows' thresholds) and the other coordinates are centred; rows that break the constant coordinate
    arrive later (the excluded set shrinks, the image is rewritten with the same centre)."""
    from vsc2022_amd.vsc.index import FlatIndex

    rng = np.random.default_rng(5)
    d, nq, nr = 129, 600, 4000
    q = np.concatenate([shifted(rng, nq, d - 1, 0.3), -rng.uniform(0.15, 0.45, (nq, 1)).astype(np.float32)], axis=1)
    r = np.concatenate([shifted(rng, nr, d - 1, 0.3), np.ones((nr, 1), np.float32)], axis=1)
    q, r = np.ascontiguousarray(q), np.ascontiguousarray(r)
    idx = FlatIndex(d, options=opts(VSC_PREFILTER="2", VSC_I8="2"))
    idx.profile(True)
    idx.add(r)
    for K in (500, 20000):
        assert_same(idx.global_topk(q, K)[:3], orc.global_threshold_search(q, r, K))
    assert idx.get_option("i8_center_on") == 1.0
    D, I = idx.search(q, 4)
    Do, Io = orc.knn(q, r, 4)
    assert np.array_equal(I, Io) and np.array_equal(bits(D), bits(Do))
    more = r[:500].copy()
    more[:, -1] = 0.5                    # the last coordinate is no longer constant
    idx.add(more)
    r2 = np.concatenate([r, more])
    assert_same(idx.global_topk(q, 3000)[:3], orc.global_threshold_search(q, r2, 3000))
    D, I = idx.search(q, 2)
    Do, Io = orc.knn(q, r2, 2)
    assert np.array_equal(I, Io) and np.array_equal(bits(D), bits(Do))
    assert i8_launches(idx) > 0


@pytest.mark.gpu
def test_centring_is_decided_by_the_data_and_cuts_the_candidates(gpu):
    """Default rule (i8_center = 1): isotropic rows are not centred (nothing changes for them), rows at mutual cosine 0.4
    are -- and the centred image passes fewer candidates to the exact stage than the uncentred one, with the same result."""
    from vsc2022_amd.vsc.index import FlatIndex

    rng = np.random.default_rng(9)
    iso = FlatIndex(128)
    iso.add(unit(rng, 20000, 128))
    iso.global_topk(unit(rng, 64, 128), 100)
    assert iso.get_option("i8_center_on") == 0.0 and iso.get_option("i8_center_share") < 0.01
    q, r = shifted(rng, 20000, 256, 0.4), shifted(np.random.default_rng(10), 150000, 256, 0.4)
    outs = []
    for c in (0, 1):
        idx = FlatIndex(256, options=opts(VSC_PREFILTER="2", VSC_I8="2", VSC_I8_CENTER=str(c)))
        idx.profile(True)
        idx.add(r)
        top = idx.global_topk(q, 400000)
        outs.append((top, idx.profile_read(reset=True)["candidates"], idx.get_option("i8_center_on")))
    assert outs[0][2] == 0.0 and outs[1][2] == 1.0
    assert_same(outs[0][0][:3], outs[1][0][:3])
    assert outs[0][0][3] == outs[1][0][3]
    print("candidates uncentred / centred:", outs[0][1], outs[1][1])



@pytest.mark.gpu
def test_parity_suites_with_forced_int8():
    """The search / candidate / golden / sharded / pre-filter parity suites again with the int8 kernel on every
    pre-filtered batch."""
    e = dict(os.environ, VSC_PREFILTER="2", VSC_TEST_QUICK="1", VSC_I8="2")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_gpu_search.py",
                        "tests/test_gpu_edge_cases.py", "tests/test_gpu_golden.py", "tests/test_gpu_sharded.py",
                        "tests/test_gpu_prefilter.py", "-k", "not forced_prefilter and not fp32_path_at_scale "
                        "and not chosen_by_size"],
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_parity_suites_with_a_centred_int8_image():
    """VSC_I8_CENTER=2 (every index centres its int8 reference image, whatever the data) with int8 forced onto every
    pre-filtered batch: the search / edge-case / golden / distribution suites and this file's oracle tests again."""
    e = dict(os.environ, VSC_PREFILTER="2", VSC_TEST_QUICK="1", VSC_I8="2", VSC_I8_CENTER="2")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_gpu_search.py",
                        "tests/test_gpu_edge_cases.py", "tests/test_gpu_golden.py", "tests/test_gpu_i8.py",
                        "tests/test_gpu_distributions.py", "-k", "matches_oracle or extreme_rows or range_search or golden or "
                        "edge or search or constant_reference or excluded_coordinates or small_sets"],
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_int8_route_with_the_fp16_screen():
    """VSC_I8_SCREEN=1 (off by default: measured neutral on the bench): the int8 candidates pass `f16_screen_kernel`
    before the exact stage.  This file's oracle tests and the search suite again with it on, int8 forced."""
    e = dict(os.environ, VSC_PREFILTER="2", VSC_I8="2", VSC_I8_SCREEN="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_gpu_search.py",
                        "tests/test_gpu_i8.py", "tests/test_gpu_prefilter.py", "-k",
                        "not forced_ and not fp32_path_at_scale and not chosen_by and not fp16_screen and not ring_kernel"],
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("nq,nr,d,K,k", [(1000, 6000, 512, 4000, 5), (129, 3000, 256, 700, 1), (300, 2500, 100, 900, 20),
                                          (640, 4000, 768, 1500, 3)])
def test_paired_work_items_give_the_same_candidates_and_results(gpu, orc, nq, nr, d, K, k):
    """csrc/sim_i8p.hip HALVES = 2 (work items of two 128-row panels, wave tiles of 256 rows x 32 columns): each half keeps
    its own scale, bound and thresholds, so the CANDIDATE SET is the one of the 128-row shape -- same count, same exact
    results (odd panel counts: the last item's second half is all rows past the batch; 768-d: the shape does not fit the
    LDS and the switch must change nothing)."""
    rng = np.random.default_rng(nq + d)
    q, r = unit(rng, nq, d), unit(rng, nr, d)
    outs = []
    for pair in ("0", "2"):
        from vsc2022_amd.vsc.index import FlatIndex

        idx = FlatIndex(d, options=opts(VSC_PREFILTER="2", VSC_I8="2", VSC_I8P_PAIR=pair))
        idx.profile(True)
        idx.add(r)
        top = idx.global_topk(q, K)
        st_top = idx.profile_read(reset=True)
        knn = idx.search(q, k)
        st_knn = idx.profile_read(reset=True)
        assert st_top["i8_launches"] > 0
        outs.append((top, knn, st_top["candidates"], st_knn["candidates"]))
    assert_same(outs[0][0][:3], outs[1][0][:3])
    assert outs[0][0][3] == outs[1][0][3]
    assert np.array_equal(bits(outs[0][1][0]), bits(outs[1][1][0])) and np.array_equal(outs[0][1][1], outs[1][1][1])
    assert outs[0][2] == outs[1][2] and outs[0][3] == outs[1][3], (outs[0][2:], outs[1][2:])
    assert_same(outs[1][0][:3], orc.global_threshold_search(q, r, K, 0))


@pytest.mark.gpu
def test_parity_suites_with_paired_int8_work_items():
    """The int8 / search / pre-filter suites again with the paired shape on EVERY int8 launch (VSC_I8P_PAIR=2; by default
    only launches with >= 512 items use it, i.e. none of the small cases of those files)."""
    e = dict(os.environ, VSC_PREFILTER="2", VSC_I8="2", VSC_I8P_PAIR="2")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_gpu_search.py",
                        "tests/test_gpu_i8.py", "tests/test_gpu_prefilter.py", "tests/test_gpu_edge_cases.py", "-k",
                        "not forced_prefilter and not fp32_path_at_scale and not chosen_by and not fp16_screen "
                        "and not parity_suites and not paired_work_items and not ring_kernel"],
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 5])
def test_knn_rows_per_launch_do_not_change_results(gpu, k):
    """api.hip knn_threshold_pass: over short reference ranges a k-NN threshold pass takes up to 262144 query rows per
    launch instead of 32768 (VSC_KNN_STEP / VSC_KNN_STEP_MAX).  300 000 x 40 000 x 64: the default, the fixed 32768 and the
    exact fp32 route give the same bits (300 000 query rows: more than one launch per range either way)."""
    from vsc2022_amd.vsc.index import FlatIndex

    rng = np.random.default_rng(77 + k)
    q, r = unit(rng, 300000, 64), unit(rng, 40000, 64)      # 262144 + 37856 rows: two launches per range
    outs = []
    for kv in (dict(VSC_KNN_STEP=None, VSC_PREFILTER="2", VSC_I8="2"), dict(VSC_KNN_STEP="32768", VSC_PREFILTER="2", VSC_I8="2"),
               dict(VSC_KNN_STEP=None, VSC_PREFILTER="0", VSC_I8=None)):
        idx = FlatIndex(64, options=opts(**kv))
        idx.profile(True)
        idx.add(r)
        D, I = idx.search(q, k)
        outs.append((D, I, idx.profile_read(reset=True)))
    assert outs[0][2]["i8_launches"] > 0 and outs[2][2]["i8_launches"] == 0
    assert outs[0][2]["i8_launches"] < outs[1][2]["i8_launches"]      # fewer, larger launches
    for o in outs[1:]:
        assert np.array_equal(outs[0][1], o[1]) and np.array_equal(bits(outs[0][0]), bits(o[0]))
