"""GPU, BASELINE configs[1] full size (200k query x 2M reference frames, 512-d fp32, K = 9.6M):
size-independent properties of the hot path, with oracle spot checks on samples (the oracle cannot
score 4e11 pairs)."""
import numpy as np
import pytest

from helpers import pair_scores

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fullsize(gpu):
    import torch
    from bench import plant_copies, synth_on_device
    from vsc2022_amd.engine import DeviceMatcher

    dev = torch.device("cuda", 0)
    n_qv, qf, n_rv, rf, dim = 8000, 25, 40000, 50, 512
    refs = synth_on_device(torch, dev, 1, n_rv, rf, dim)
    queries = synth_on_device(torch, dev, 1001, n_qv, qf, dim)
    gt = plant_copies(torch, dev, 2001, queries, n_qv, qf, refs, n_rv, rf)
    m = DeviceMatcher(refs, np.arange(n_rv + 1, dtype=np.int64) * rf, 0)
    m.set_queries(queries, np.arange(n_qv + 1, dtype=np.int64) * qf)
    return m, queries, refs, gt, (n_qv, qf, n_rv, rf, dim)


def test_fullsize_search_properties(fullsize, orc):
    import torch

    m, queries, refs, gt, (n_qv, qf, n_rv, rf, dim) = fullsize
    K = 1200 * n_qv
    hi, hj, hs, radius = m.search(K)
    assert hs.numel() == K
    s = hs.cpu().numpy()
    i = hi.cpu().numpy()
    j = hj.cpu().numpy()
    # ordering: score desc, then (row, ref) asc
    assert np.all(s[:-1] >= s[1:])
    same = s[:-1] == s[1:]
    key = i.astype(np.int64) * (n_rv * rf) + j
    assert np.all(key[:-1][same] < key[1:][same])
    assert np.all(s > np.float32(radius))
    assert len(np.unique(key)) == K
    # spot check: 2000 hits recomputed by the oracle chain are bit-identical
    rng = np.random.default_rng(0)
    pick = rng.choice(K, 2000, replace=False)
    q_rows = queries[torch.from_numpy(i[pick]).long().to(queries.device)].cpu().numpy()
    r_rows = refs[torch.from_numpy(j[pick]).long().to(refs.device)].cpu().numpy()
    exact = pair_scores(orc, q_rows, r_rows)
    assert np.array_equal(exact.view(np.uint32), s[pick].view(np.uint32))
    # completeness on a sample of rows: every score above the K-th best hit of the search is a hit
    rows = rng.choice(n_qv * qf, 24, replace=False)
    sub = orc.scores(queries[torch.from_numpy(rows).long().to(queries.device)].cpu().numpy(),
                     refs[: 200000].cpu().numpy())
    cut = s[-1]
    hits_set = set(zip(i.tolist(), j.tolist()))
    rr, cc = np.nonzero(sub > cut)
    assert len(rr) > 0
    for a, b in zip(rr, cc):
        assert (int(rows[a]), int(b)) in hits_set
    # idempotence
    hi2, hj2, hs2, radius2 = m.search(K)
    assert radius2 == radius and torch.equal(hs2, hs) and torch.equal(hi2, hi) and torch.equal(hj2, hj)


def _exact_index(torch, refs, dim):
    """A second index over the same rows whose every search runs on the exact fp32 MFMA kernel alone (the switches are
    options of the handle, set while it is empty: include/vscmi.h)."""
    from vsc2022_amd import _lib
    from vsc2022_amd.vsc.index import FlatIndex

    exact = FlatIndex(dim, _lib.METRIC_INNER_PRODUCT, 0, options={"prefilter": 0})
    exact.add(refs)
    return exact


def test_fullsize_routes_identical_exhaustively(fullsize):
    """the sampled checks above look at 1e-5 of the matrix; a pre-filter that loses a tile's hits
    now and then (the hazard class of sim_i8p.hip's `s_nop 4`) would pass them.  Here the default route (fp16 / int8
    pre-filters + exact stage) and the all-fp32 route (VSC_PREFILTER=0: sim_thresh_kernel alone, no bound, no
    candidate list) are compared on the WHOLE 200 k x 2 M matrix: all 9.6 M (row, ref, score bits) and the final radius;
    and the k-NN (k = 20: pre-filtered ranges vs the exact kernel over all references) likewise, all 4 M entries."""
    import torch

    m, queries, refs, gt, (n_qv, qf, n_rv, rf, dim) = fullsize
    K = 1200 * n_qv
    exact = _exact_index(torch, refs, dim)
    ei, ej, es, erad = exact.global_topk(queries, K, device_out=True)
    for rep in range(2):   # (twice: the hazard that prompted this test lost a hit in 16 % of the runs)
        hi, hj, hs, radius = m.search(K)
        assert radius == erad and hs.numel() == es.numel() == K
        assert torch.equal(hi, ei) and torch.equal(hj, ej) and torch.equal(hs.view(torch.int32), es.view(torch.int32))
    del ei, ej, es, hi, hj, hs
    De, Ie = exact.search(queries, 20, device_out=True)
    Dp, Ip = m.index.search(queries, 20, device_out=True)
    assert torch.equal(Ie, Ip) and torch.equal(De.view(torch.int32), Dp.view(torch.int32))
    del exact
    torch.cuda.empty_cache()


def test_fullsize_pipeline_finds_the_planted_copies(fullsize):
    m, queries, refs, gt, (n_qv, qf, n_rv, rf, dim) = fullsize
    res = m.match()
    assert res.n_hits == 1200 * n_qv and res.n_candidates == 25 * n_qv and res.n_localized == 5 * n_qv
    cand = set(zip(res.cand_q.cpu().tolist(), res.cand_r.cpu().tolist()))
    planted = set(gt)
    assert len(planted & cand) >= 0.99 * len(planted)
    sc = res.cand_score.cpu().numpy()
    assert np.all(sc[:-1] >= sc[1:])
    nbox = res.nbox.cpu().numpy()
    loc = set(zip(res.cand_q[: res.n_localized].cpu().numpy()[nbox > 0].tolist(),
                  res.cand_r[: res.n_localized].cpu().numpy()[nbox > 0].tolist()))
    assert len(planted & loc) >= 0.95 * len(planted)
    boxes = res.boxes.cpu().numpy()
    for p in np.nonzero(nbox > 0)[0][:200]:
        for b in range(nbox[p]):
            x1, y1, x2, y2 = boxes[p, b]
            assert 0 <= x1 < x2 < qf and 0 <= y1 < y2 < rf


def test_fullsize_knn_properties(fullsize, orc):
    """Brute-force cosine k-NN at BASELINE configs[1]'s literal shape (200k x 2M x 512, k = 1 and 20) through the
    faiss-like surface (`index.search(x, k)`, vsc/index.py:167-177, vsc/baseline/score_normalization.py:96):
    per-row order, completeness and scores on sampled rows against the oracle, k = 1 == first column of k = 20,
    idempotence."""
    import torch

    m, queries, refs, gt, (n_qv, qf, n_rv, rf, dim) = fullsize
    nq, nr = n_qv * qf, n_rv * rf
    D20, I20 = m.index.search(queries, 20)
    assert D20.shape == (nq, 20) and I20.shape == (nq, 20)
    assert np.all(D20[:, :-1] >= D20[:, 1:])
    tie = D20[:, :-1] == D20[:, 1:]
    assert np.all(I20[:, :-1][tie] < I20[:, 1:][tie])          # ties: lower reference row first
    assert I20.min() >= 0 and I20.max() < nr
    srt = np.sort(I20, axis=1)
    assert np.all(srt[:, :-1] != srt[:, 1:])                    # no reference twice in a row's list
    D1, I1 = m.index.search(queries, 1)
    assert np.array_equal(D1[:, 0].view(np.uint32), D20[:, 0].view(np.uint32)) and np.array_equal(I1[:, 0], I20[:, 0])
    # sampled rows: every listed score is the oracle's fp32 chain bit for bit, and nothing in a 200k-row slice of
    # the references beats a row's 20th score without being listed
    rng = np.random.default_rng(3)
    rows = rng.choice(nq, 16, replace=False)
    qs = queries[torch.from_numpy(rows).long().to(queries.device)].cpu().numpy()
    lo = int(rng.integers(0, nr - 200000))
    sub = orc.scores(qs, refs[lo : lo + 200000].cpu().numpy())
    for a, row in enumerate(rows):
        listed = refs[torch.from_numpy(I20[row]).long().to(refs.device)].cpu().numpy()
        exact = pair_scores(orc, np.repeat(qs[a : a + 1], 20, axis=0), listed)
        assert np.array_equal(exact.view(np.uint32), D20[row].view(np.uint32)), row
        better = np.nonzero(sub[a] > D20[row, -1])[0] + lo
        assert set(better.tolist()) <= set(I20[row].tolist()), row
        equal = np.nonzero(sub[a] == D20[row, -1])[0] + lo       # ties with the 20th: only lower ids may be listed before it
        assert all(int(e) in set(I20[row].tolist()) or e > I20[row, -1] for e in equal)
    # planted copies: the copied frames find their source video
    row2r = np.repeat(np.arange(n_rv), rf)
    planted = dict(gt)
    hit = sum(1 for qv, rv in planted.items() if rv in set(row2r[I1[qv * qf : (qv + 1) * qf, 0]].tolist()))
    assert hit >= 0.99 * len(planted)
    # idempotence
    D20b, I20b = m.index.search(queries, 20)
    assert np.array_equal(D20b.view(np.uint32), D20.view(np.uint32)) and np.array_equal(I20b, I20)


def test_fullsize_score_normalised_path(fullsize, orc):
    """BASELINE configs[3]'s extra stage at configs[1]'s size: `score_normalize`
    (vsc/baseline/score_normalization.py:31-105) of 200 k query rows against a 2 M-row noise set on the device
    (`DeviceScoreNormalizer`), then the search / candidates / localisation on the 511+1-d descriptors.
    Sampled rows against the ORACLE: the prepared rows, the 1-NN value behind the bias column (orc.knn over all
    2 M noise rows), hit scores of the normalised search bit for bit, order, completeness, planted copies."""
    import torch
    from bench import synth_on_device
    from vsc2022_amd.engine import DeviceMatcher, DeviceScoreNormalizer

    m, queries, refs, gt, (n_qv, qf, n_rv, rf, dim) = fullsize
    dev = queries.device
    nq, nr = n_qv * qf, n_rv * rf
    noise = synth_on_device(torch, dev, 77, n_rv, rf, dim, static_frac=0.0)
    beta = 1.2
    norm = DeviceScoreNormalizer(noise, beta=beta)
    qn = norm.queries(queries)
    assert qn.shape == (nq, dim) and qn.is_cuda
    # ---- the 1-NN behind the bias column, on sampled rows against all 2 M noise rows
    rng = np.random.default_rng(5)
    rows = np.sort(rng.choice(nq, 12, replace=False))
    keep = norm.sel.cpu().numpy()
    assert len(keep) == dim - 1
    noise_prep = norm._prepare(noise)                       # what the noise index holds (column dropped, row-L2)
    q_rows = queries[torch.from_numpy(rows).to(dev)].cpu().numpy()[:, keep]
    q_prep = orc.row_normalize(q_rows)
    got = qn[torch.from_numpy(rows).to(dev)].cpu().numpy()
    assert np.array_equal(got[:, : dim - 1].view(np.uint32), q_prep.view(np.uint32))
    best, best_id = orc.knn(q_prep, noise_prep.cpu().numpy(), 1)
    assert np.array_equal(got[:, dim - 1].view(np.uint32), (best[:, 0] * np.float32(-beta)).view(np.uint32))
    del noise_prep, noise
    # ---- the search on the normalised descriptors
    rn = norm.refs(refs)
    assert torch.equal(rn[:, dim - 1], torch.ones(nr, device=dev))
    m2 = DeviceMatcher(rn, np.arange(n_rv + 1, dtype=np.int64) * rf, 0)
    m2.set_queries(qn, np.arange(n_qv + 1, dtype=np.int64) * qf)
    K = 1200 * n_qv
    hi, hj, hs, radius = m2.search(K)
    assert hs.numel() == K
    s, i, j = hs.cpu().numpy(), hi.cpu().numpy(), hj.cpu().numpy()
    assert np.all(s[:-1] >= s[1:]) and np.all(s > np.float32(radius))
    same = s[:-1] == s[1:]
    key = i.astype(np.int64) * nr + j
    assert np.all(key[:-1][same] < key[1:][same]) and len(np.unique(key)) == K
    pick = rng.choice(K, 1000, replace=False)
    a = qn[torch.from_numpy(i[pick]).long().to(dev)].cpu().numpy()
    b = rn[torch.from_numpy(j[pick]).long().to(dev)].cpu().numpy()
    exact = pair_scores(orc, a, b)
    assert np.array_equal(exact.view(np.uint32), s[pick].view(np.uint32))
    sub = orc.scores(got, rn[:200000].cpu().numpy())
    hits_set = set(zip(i.tolist(), j.tolist()))
    rr, cc = np.nonzero(sub > s[-1])
    for x, y in zip(rr, cc):
        assert (int(rows[x]), int(y)) in hits_set
    # ---- candidates + localisation as the reference runs them on normalised descriptors (bias 0.5, MaxSim)
    res = m2.match(bias=0.5)
    assert res.n_hits == K and res.n_candidates == 25 * n_qv and res.n_localized == 5 * n_qv
    planted = set(gt)
    cand = set(zip(res.cand_q.cpu().tolist(), res.cand_r.cpu().tolist()))
    assert len(planted & cand) >= 0.99 * len(planted)
    nbox = res.nbox.cpu().numpy()
    loc = set(zip(res.cand_q[: res.n_localized].cpu().numpy()[nbox > 0].tolist(),
                  res.cand_r[: res.n_localized].cpu().numpy()[nbox > 0].tolist()))
    assert len(planted & loc) >= 0.95 * len(planted)
