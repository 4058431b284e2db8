"""GPU: the selection / merge entry points the sharded paths run on instead of torch.sort:
`vsc_score_histogram` + `vsc_score_pick` (exact order statistics over unsorted score lists, `dist.kth_best_unsorted`),
`vsc_argsort_scores` (stable, best first: `dist.merge_candidates`) and `vsc_merge_topk` (per-row merge of reference shards'
k-NN lists: `dist.ref_sharded_knn`) -- each against a numpy restatement, ties, signed zeros and unaligned views included."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scores(rng, n, style):
    if style == 0:
        x = rng.standard_normal(n).astype(np.float32)
    elif style == 1:                                    # a kept list: one sign, one or two exponents, many ties
        x = (0.2 + 0.05 * rng.random(n)).astype(np.float32)
        x = np.round(x * 4096) / 4096
    elif style == 2:                                    # massive ties + signed zeros
        x = rng.choice(np.array([-1.5, -0.0, 0.0, 0.25, 0.25, 3.0], dtype=np.float32), n)
    else:
        x = np.exp(rng.uniform(-30, 30, n)).astype(np.float32) * rng.choice(np.array([-1, 1], dtype=np.float32), n)
    return x.astype(np.float32)


def test_kth_best_unsorted_on_the_library(gpu):
    from vsc2022_amd.dist import kth_best_unsorted

    rng = np.random.default_rng(0)
    for trial in range(60):
        n = int(rng.choice([1, 2, 5, 63, 64, 257, 1000, 4097, 100003, 1 << 20]))
        x = _scores(rng, n + 3, trial % 4)
        off = int(rng.integers(0, 4))                   # unaligned views: the kernel's scalar head / tail
        t = torch.from_numpy(x).cuda()[off : off + n]
        want = np.sort(x[off : off + n])[::-1]
        for k in {1, n, max(1, n // 2), max(1, n - 1), min(n, 7)}:
            tau, total = kth_best_unsorted(t, k)
            assert total == n
            assert np.float32(tau) == want[k - 1], (trial, n, k, tau, want[k - 1])
        tau, total = kth_best_unsorted(t, n + 1)
        assert tau == float("-inf") and total == n
    tau, total = kth_best_unsorted(torch.zeros(0, dtype=torch.float32, device="cuda"), 1)
    assert tau == float("-inf") and total == 0


def test_argsort_scores_is_the_stable_descending_order(gpu):
    from vsc2022_amd.dist import argsort_scores_desc

    rng = np.random.default_rng(1)
    for trial in range(24):
        n = int(rng.choice([1, 3, 64, 1000, 4096, 4097, 250000]))
        x = _scores(rng, n, trial % 4)
        perm = argsort_scores_desc(torch.from_numpy(x).cuda()).cpu().numpy()
        want = np.argsort(-(x.astype(np.float64) + 0.0), kind="stable")   # (-0.0 == +0.0: both negate to a zero)
        assert np.array_equal(perm, want), (trial, n)
    assert argsort_scores_desc(torch.zeros(0, device="cuda")).numel() == 0


def test_merge_topk_equals_the_lexicographic_merge(gpu):
    from vsc2022_amd import _lib

    rng = np.random.default_rng(2)
    for trial, (nq, shards, k) in enumerate([(1, 2, 1), (37, 3, 5), (1000, 8, 20), (513, 4, 64), (300, 16, 64), (50, 2, 7)]):
        m = shards * k
        s = _scores(rng, nq * m, trial % 3).reshape(nq, m)
        ids = np.stack([rng.permutation(10 * m)[:m] for _ in range(nq)]).astype(np.int64)
        empty = rng.random((nq, m)) < (0.3 if trial % 2 else 0.0)
        if trial == 1:
            empty[0, :] = True                          # a row with no candidate at all
        ids[empty] = -1
        ts, ti = torch.from_numpy(s).cuda(), torch.from_numpy(ids).cuda()
        out_s = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        out_i = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        _lib.check(_lib.lib().vsc_merge_topk(ts.data_ptr(), ti.data_ptr(), nq, m, k, out_s.data_ptr(), out_i.data_ptr(), 0))
        got_s, got_i = out_s.cpu().numpy(), out_i.cpu().numpy()
        for x in range(nq):
            valid = np.flatnonzero(ids[x] >= 0)
            order = valid[np.lexsort((ids[x][valid], -s[x][valid].astype(np.float64)))][:k]
            n_ok = len(order)
            assert np.array_equal(got_i[x, :n_ok], ids[x][order]), (trial, x)
            assert np.array_equal(got_s[x, :n_ok].view(np.uint32), s[x][order].view(np.uint32))
            assert (got_i[x, n_ok:] == -1).all() and (got_s[x, n_ok:] == -np.finfo(np.float32).max).all()
    with pytest.raises(ValueError):
        _lib.check(_lib.lib().vsc_merge_topk(ts.data_ptr(), ti.data_ptr(), 1, 4, 5, out_s.data_ptr(), out_i.data_ptr(), 0))


def test_filter_hits_keeps_exactly_the_hits_beyond_the_radius(gpu):
    from vsc2022_amd.dist import filter_hits

    rng = np.random.default_rng(4)
    for trial in range(16):
        n = int(rng.choice([0, 1, 255, 2048, 2049, 100003, 3_000_000]))
        x = _scores(rng, n, trial % 4)
        i = rng.integers(0, 1 << 20, n).astype(np.int32)
        j = rng.integers(0, 1 << 21, n).astype(np.int32)
        radius = float(np.median(x)) if n else 0.0
        gi, gj, gs = filter_hits(torch.from_numpy(i).cuda(), torch.from_numpy(j).cuda(), torch.from_numpy(x).cuda(), radius)
        keep = x > np.float32(radius)
        want = sorted(zip(i[keep].tolist(), j[keep].tolist(), x[keep].view(np.uint32).tolist()))
        got = sorted(zip(gi.cpu().tolist(), gj.cpu().tolist(), gs.cpu().numpy().view(np.uint32).tolist()))
        assert got == want, (trial, n, len(got), len(want))
    # +-inf radii: everything / nothing
    t = torch.arange(5, dtype=torch.float32, device="cuda")
    z = torch.zeros(5, dtype=torch.int32, device="cuda")
    assert filter_hits(z, z, t, float("-inf"))[2].numel() == 5 and filter_hits(z, z, t, float("inf"))[2].numel() == 0


def test_score_histogram_rejects_bad_arguments(gpu):
    from vsc2022_amd import _lib

    st = torch.tensor([0, 0, 1, 0], dtype=torch.int64, device="cuda")
    h = torch.empty(256, dtype=torch.int64, device="cuda")
    x = torch.ones(8, device="cuda")
    with pytest.raises(ValueError):
        _lib.check(_lib.lib().vsc_score_histogram(x.data_ptr(), 8, st.data_ptr(), 12, h.data_ptr(), 0))
    with pytest.raises(ValueError):
        _lib.check(_lib.lib().vsc_score_pick(None, st.data_ptr(), 24, 0))


def test_phase_timer_reports_without_synchronising(gpu):
    from vsc2022_amd.dist import PhaseTimer, reduce_phase_report

    t = PhaseTimer(torch.device("cuda", 0))
    a = torch.randn(4096, 4096, device="cuda")
    for _ in range(3):
        with t.phase("matmul"):
            b = a @ a
    t.add_bytes("matmul", 123)
    b.sum().item()
    rep = t.collect()
    assert rep["matmul"]["calls"] == 3 and rep["matmul"]["bytes"] == 123 and rep["matmul"]["device_ms"] > 0.0
    assert reduce_phase_report(rep, torch.device("cuda", 0)) == rep
