"""CPU: the C oracle (oracle/libvscoracle.so) against the golden vectors produced by the REFERENCE's
own Python (oracle/gen_golden.py).  This is what pins the oracle before it is trusted as the
checker of the HIP path."""
import numpy as np
import pytest

from helpers import bits, load, row_maps

SEARCH_CASES = ["g2_search_plain", "g2_search_ties", "g2_search_d512", "g2_search_l2"]


@pytest.mark.parametrize("case", SEARCH_CASES)
def test_global_threshold_search_matches_reference(orc, case):
    fx = load(case)
    metric = int(fx["metric"])
    for K in fx["Ks"]:
        i, j, s = orc.global_threshold_search(fx["q_feats"], fx["r_feats"], int(K), metric)
        assert np.array_equal(i, fx[f"K{K}_i"]) and np.array_equal(j, fx[f"K{K}_j"]), (case, K)
        assert np.array_equal(bits(s), bits(fx[f"K{K}_s"])), (case, K)


@pytest.mark.parametrize("case", SEARCH_CASES[:3])
def test_pair_max_matches_reference_candidates(orc, case):
    fx = load(case)
    row2q, _ = row_maps(fx, "q")
    row2r, _ = row_maps(fx, "r")
    for K in fx["Ks"]:
        q, r, s, first = orc.pair_max(fx[f"K{K}_i"], fx[f"K{K}_j"], fx[f"K{K}_s"], row2q, row2r)
        assert np.array_equal(fx["q_ids"][q], fx[f"K{K}_cand_q"])
        assert np.array_equal(fx["r_ids"][r], fx[f"K{K}_cand_r"])
        assert np.array_equal(bits(s), bits(fx[f"K{K}_cand_s"]))
        # first-appearance order of VideoIndex.search's PairMatches list is the hit order of `first`
        order = np.argsort(first, kind="stable")
        assert np.array_equal(fx["q_ids"][q][order], fx[f"K{K}_pm_q"])
        assert np.array_equal(fx["r_ids"][r][order], fx[f"K{K}_pm_r"])


@pytest.mark.parametrize("case", SEARCH_CASES)
def test_knn_matches_reference(orc, case):
    fx = load(case)
    metric = int(fx["metric"])
    row2q, _ = row_maps(fx, "q")
    row2r, _ = row_maps(fx, "r")
    for k in fx["knn_ks"]:
        D, I = orc.knn(fx["q_feats"], fx["r_feats"], int(k), metric)
        # regroup (row, rank)-ordered hits per (query video, ref video) in first-appearance order
        qi = np.repeat(np.arange(D.shape[0]), int(k))
        key = row2q[qi].astype(np.int64) * 100000 + row2r[I.reshape(-1)]
        _, first = np.unique(key, return_index=True)
        order_pairs = np.argsort(first, kind="stable")
        uniq = np.unique(key)[order_pairs]
        assert np.array_equal(fx["q_ids"][(uniq // 100000)], fx[f"knn{k}_pm_q"])
        assert np.array_equal(fx["r_ids"][(uniq % 100000)], fx[f"knn{k}_pm_r"])
        rank = {u: n for n, u in enumerate(uniq)}
        grouped = np.argsort(np.array([rank[x] for x in key]), kind="stable")
        assert np.array_equal(bits(D.reshape(-1)[grouped]), bits(fx[f"knn{k}_pm_score32"]))


def test_g1_known_answer(orc):
    fx = load("g1_candidates")
    r = np.concatenate([fx["r5"], fx["r8"], fx["r10"]])
    row2r = np.repeat(np.arange(3, dtype=np.int32), [5, 3, 3])
    i, j, s = orc.global_threshold_search(fx["q_feat"], r, 6)
    q, rr, sc, _ = orc.pair_max(i, j, s, np.zeros(3, np.int32), row2r)
    assert np.array_equal(np.array([5, 8, 10])[rr], fx["cand_r"]) and np.array_equal(sc, fx["cand_s"])
    assert list(sc) == [2.0, 1.0, 0.25]


@pytest.mark.parametrize("case", ["g5_localization_default", "g5_localization_ref_params",
                                  "g5_localization_ref_params_nobias"])
def test_tn_matches_reference_localization(orc, case):
    """oracle pair_sims + C tn vs VCSLLocalization{MaxSim,CandidateScore}.localize_all of the reference
    (whose sims come from np.matmul: box coordinates must agree exactly, scores to fp32 round-off)."""
    fx = load(case)
    kw = {str(k): int(v) for k, v in zip(fx["kw_keys"], fx["kw_vals"]) if str(k) != "concurrency"}
    bias = float(fx["bias"])
    _, qcut = row_maps(fx, "q")
    _, rcut = row_maps(fx, "r")
    qpos = {str(v): k for k, v in enumerate(fx["q_ids"])}
    rpos = {str(v): k for k, v in enumerate(fx["r_ids"])}
    def run(b):
        rows, scores, ids = [], [], []
        for cq, cr, cs in zip(fx["cand_q"], fx["cand_r"], fx["cand_s"]):
            a, c = qpos[str(cq)], rpos[str(cr)]
            qf = fx["q_feats"][qcut[a]:qcut[a + 1]]
            rf = fx["r_feats"][rcut[c]:rcut[c + 1]]
            qts, rts = fx["q_ts"][qcut[a]:qcut[a + 1]], fx["r_ts"][rcut[c]:rcut[c + 1]]
            sims = orc.pair_sims(qf, rf, b)
            for (x1, y1, x2, y2) in orc.tn(sims, **kw):
                rows.append([qts[x1][0], qts[x2][1], rts[y1][0], rts[y2][1]])
                scores.append(sims[x1:x2, y1:y2].max() - np.float32(b))
                ids.append((str(cq), str(cr), float(cs)))
        return np.array(rows, dtype=np.float64), np.array(scores, dtype=np.float64), ids

    rows, scores, ids = run(bias)  # VCSLLocalizationMaxSim(similarity_bias=bias)
    exp = fx["maxsim_m_rows"]
    assert len(rows) == len(exp) and len(rows) > 5
    assert np.array_equal(rows, exp[:, 1:])
    assert [i[0] for i in ids] == list(fx["maxsim_m_q"]) and [i[1] for i in ids] == list(fx["maxsim_m_r"])
    assert np.allclose(scores, exp[:, 0], atol=2e-6)
    rows, _, ids = run(0.0)  # VCSLLocalizationCandidateScore (no bias)
    assert np.array_equal(rows, fx["candscore_m_rows"][:, 1:])
    assert np.allclose([i[2] for i in ids], fx["candscore_m_rows"][:, 0])


def test_row_normalize_matches_sklearn_semantics(orc):
    fx = load("g4_score_norm")
    x = np.delete(fx["q_feats"], int(fx["low_var_dim"]), axis=1)
    out = orc.row_normalize(x)
    assert np.allclose(out, fx["b10_q"][:, :-1], atol=1e-6)  # reference = sklearn.normalize
    z = np.zeros((3, 5), np.float32)
    assert np.array_equal(orc.row_normalize(z), z)


def test_score_norm_algebra(orc):
    """q' . r' == q . r - beta * max_n(q . n)  (vsc/baseline/score_normalization.py:38-60) on the
    reference's adapted descriptors."""
    fx = load("g4_score_norm")
    d = int(fx["low_var_dim"])
    for tag, beta, drop in (("b10", 1.0, True), ("b12", 1.2, True), ("b12_keepdim", 1.2, False)):
        q = orc.row_normalize(np.delete(fx["q_feats"], d, 1) if drop else fx["q_feats"])
        n = orc.row_normalize(np.delete(fx["n_feats"], d, 1) if drop else fx["n_feats"])
        D, _ = orc.knn(q, n, 1)
        assert np.allclose(fx[f"{tag}_q"][:, -1], -beta * D[:, 0], atol=2e-6)
        assert np.all(fx[f"{tag}_r"][:, -1] == 1.0)


def test_oracle_reproduces_config1_descriptor_track(orc):
    """Fixture g8 (the reference's evaluate_descriptor_track on BASELINE configs[0]'s shape): the C oracle's
    search + pair-max gives the same 1250 candidates, scores bit for bit."""
    from helpers import g8_inputs

    fx = load("g8_config1_pipeline")
    q, r, noise, gts, digest = g8_inputs()
    assert digest == str(fx["digest"])
    Q = np.concatenate([v.feature for v in q])
    R = np.concatenate([v.feature for v in r])
    row2q = np.repeat(np.arange(len(q), dtype=np.int32), [len(v.feature) for v in q])
    row2r = np.repeat(np.arange(len(r), dtype=np.int32), [len(v.feature) for v in r])
    i, j, s = orc.global_threshold_search(Q, R, 1200 * len(q))
    pq, pr, ps, _ = orc.pair_max(i, j, s, row2q, row2r)
    n = 25 * len(q)
    assert [q[k].video_id for k in pq[:n]] == list(fx["desc_cand_q"])
    assert [r[k].video_id for k in pr[:n]] == list(fx["desc_cand_r"])
    assert np.array_equal(ps[:n].view(np.uint32), fx["desc_cand_s"].view(np.uint32))
