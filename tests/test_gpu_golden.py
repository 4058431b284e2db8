"""GPU parity: the public mirrors (vsc.index / vsc.candidates / score_normalization / localization)
against the golden vectors produced by the REFERENCE's own Python (oracle/gen_golden.py)."""
import io
import os

import numpy as np
import pytest

from helpers import bits, flatten_pairmatches, load, videos

pytestmark = pytest.mark.gpu

SEARCH_CASES = ["g2_search_plain", "g2_search_ties", "g2_search_d512", "g2_search_l2"]


def test_g1_reference_known_answer(gpu):
    """tests/test_candidates.py:15-83 of the reference, verbatim expectation (int ids, mixed timestamps)."""
    from vsc2022_amd.vsc.candidates import CandidateGeneration, MaxScoreAggregation
    from vsc2022_amd.vsc.index import VideoFeature
    from vsc2022_amd.vsc.metrics import CandidatePair

    fx = load("g1_candidates")
    queries = [VideoFeature(video_id=1, feature=fx["q_feat"], timestamps=np.array([0.0, 1.0, 2.0]))]
    refs = [
        VideoFeature(video_id=5, feature=fx["r5"], timestamps=np.array([2.0, 4.0, 6.0, 8.0, 10.0])),
        VideoFeature(video_id=8, feature=fx["r8"], timestamps=np.array([[0.0, 5.0], [5.0, 10.0], [10.0, 15.0]])),
        VideoFeature(video_id=10, feature=fx["r10"], timestamps=np.array([0.0, 0.1, 0.2])),
    ]
    cg = CandidateGeneration(refs, MaxScoreAggregation())
    candidates = cg.query(queries, 2 * 3)
    assert 3 == len(candidates)
    assert candidates == [CandidatePair(query_id=1, ref_id=5, score=2.0), CandidatePair(query_id=1, ref_id=8, score=1.0),
                          CandidatePair(query_id=1, ref_id=10, score=0.25)]


def test_reference_index_test(gpu):
    """tests/test_index.py:15-53 of the reference (L2 metric, global_k = 1 and k-NN)."""
    from vsc2022_amd.vsc.index import METRIC_L2, VideoFeature, VideoIndex

    feat = np.array([[[1, 2, 3], [4, 5, 6], [7, 8, 9]], [[11, 12, 13], [14, 15, 16], [17, 18, 19]],
                     [[111, 112, 113], [114, 115, 116], [117, 118, 119]]], dtype=np.float32)
    for global_k, expect in ((1, 0), (-1, 3)):
        q = [VideoFeature(video_id=f"Q{i:06d}", feature=f, timestamps=np.arange(3, dtype=np.float32)) for i, f in enumerate(feat)]
        db = [VideoFeature(video_id=f"R{i:06d}", feature=f, timestamps=np.arange(3, dtype=np.float32)) for i, f in enumerate(feat)]
        index = VideoIndex(3, "Flat", METRIC_L2)
        index.add(db)
        results = index.search(q, global_k)
        assert len(results) == expect  # the 9 exact-zero distances tie on the cut
        for result in results:
            assert result.query_id[1:] == result.ref_id[1:]


@pytest.mark.parametrize("case", SEARCH_CASES)
def test_video_index_search_matches_reference(gpu, case):
    from vsc2022_amd.vsc.index import VideoFeature, VideoIndex

    fx = load(case)
    metric, dim = int(fx["metric"]), fx["q_feats"].shape[1]
    q, r = videos(fx, "q", VideoFeature), videos(fx, "r", VideoFeature)
    index = VideoIndex(dim, "Flat", metric)
    index.add(r[: len(r) // 2])
    index.add(r[len(r) // 2:])  # incremental add, as the API allows
    for K in fx["Ks"]:
        raw = index._global_threshold_knn_search(fx["q_feats"], int(K))
        assert np.array_equal([t[0] for t in raw], fx[f"K{K}_i"]) and np.array_equal([t[1] for t in raw], fx[f"K{K}_j"])
        assert np.array_equal(bits([t[2] for t in raw]), bits(fx[f"K{K}_s"]))
        pq, pr, pn, rows, sc = flatten_pairmatches(index.search(q, int(K)))
        assert np.array_equal(pq, fx[f"K{K}_pm_q"]) and np.array_equal(pr, fx[f"K{K}_pm_r"])
        assert np.array_equal(pn, fx[f"K{K}_pm_n"])
        assert np.array_equal(rows[:, :4], fx[f"K{K}_pm_rows"][:, :4])  # timestamps
        assert np.array_equal(bits(sc), bits(fx[f"K{K}_pm_score32"]))
    for k in fx["knn_ks"]:
        pq, pr, pn, rows, sc = flatten_pairmatches(index.search(q, -int(k)))
        assert np.array_equal(pq, fx[f"knn{k}_pm_q"]) and np.array_equal(pr, fx[f"knn{k}_pm_r"])
        assert np.array_equal(pn, fx[f"knn{k}_pm_n"]) and np.array_equal(rows[:, :4], fx[f"knn{k}_pm_rows"][:, :4])
        assert np.array_equal(bits(sc), bits(fx[f"knn{k}_pm_score32"]))


@pytest.mark.parametrize("case", SEARCH_CASES[:3])
def test_candidate_generation_matches_reference(gpu, case):
    from vsc2022_amd.vsc.candidates import CandidateGeneration, MaxScoreAggregation, ScoreAggregation
    from vsc2022_amd.vsc.index import VideoFeature

    fx = load(case)
    q, r = videos(fx, "q", VideoFeature), videos(fx, "r", VideoFeature)
    cg = CandidateGeneration(r, MaxScoreAggregation())

    class UserMax(ScoreAggregation):  # a user aggregation goes through the generic PairMatches route
        def aggregate(self, match):
            return np.max([m.score for m in match.matches])

    cg_generic = CandidateGeneration(r, UserMax())
    for K in fx["Ks"]:
        for cands in (cg.query(q, int(K)), cg_generic.query(q, int(K))):
            assert np.array_equal([c.query_id for c in cands], fx[f"K{K}_cand_q"])
            assert np.array_equal([c.ref_id for c in cands], fx[f"K{K}_cand_r"])
            assert np.array_equal(bits([c.score for c in cands]), bits(fx[f"K{K}_cand_s"]))


def test_score_normalize_matches_reference(gpu):
    from vsc2022_amd.vsc.baseline.score_normalization import score_normalize
    from vsc2022_amd.vsc.index import VideoFeature

    fx = load("g4_score_norm")
    q, r, n = (videos(fx, p, VideoFeature) for p in ("q", "r", "n"))
    for tag, kw in (("b10", dict(beta=1.0)), ("b12", dict(beta=1.2)), ("b12_keepdim", dict(beta=1.2, replace_dim=False))):
        aq, ar = score_normalize(q, r, n, **kw)
        gq = np.concatenate([v.feature for v in aq])
        gr = np.concatenate([v.feature for v in ar])
        assert gq.shape == fx[f"{tag}_q"].shape and gr.shape == fx[f"{tag}_r"].shape
        # fp tolerance: sklearn's normalize / BLAS 1-NN vs the engine's defined-order arithmetic
        assert np.allclose(gq, fx[f"{tag}_q"], atol=2e-6) and np.allclose(gr, fx[f"{tag}_r"], atol=2e-6)
        assert [v.video_id for v in aq] == [v.video_id for v in q]
    with pytest.raises(Exception, match="against VSC rules"):
        score_normalize(q, r, r)


@pytest.mark.parametrize("case", ["g5_localization_default", "g5_localization_ref_params",
                                  "g5_localization_ref_params_nobias"])
def test_localization_matches_reference(gpu, case):
    from vsc2022_amd.vsc.baseline.localization import VCSLLocalizationCandidateScore, VCSLLocalizationMaxSim
    from vsc2022_amd.vsc.index import VideoFeature
    from vsc2022_amd.vsc.metrics import CandidatePair

    fx = load(case)
    kw = {str(k): int(v) for k, v in zip(fx["kw_keys"], fx["kw_vals"])}
    bias = float(fx["bias"])
    q, r = videos(fx, "q", VideoFeature), videos(fx, "r", VideoFeature)
    cands = [CandidatePair(str(a), str(b), float(s)) for a, b, s in zip(fx["cand_q"], fx["cand_r"], fx["cand_s"])]
    for cls, tag, extra in ((VCSLLocalizationMaxSim, "maxsim", dict(similarity_bias=bias)),
                            (VCSLLocalizationCandidateScore, "candscore", {})):
        loc = cls(q, r, "TN", **extra, **kw)
        ms = loc.localize_all(cands[:100]) + loc.localize_all(cands[100:])  # batching must not matter
        exp = fx[f"{tag}_m_rows"]
        assert len(ms) == len(exp)
        assert [m.query_id for m in ms] == list(fx[f"{tag}_m_q"]) and [m.ref_id for m in ms] == list(fx[f"{tag}_m_r"])
        got = np.array([[m.query_start, m.query_end, m.ref_start, m.ref_end] for m in ms], dtype=np.float64)
        assert np.array_equal(got, exp[:, 1:])          # boxes -> timestamps: exact
        assert np.allclose([m.score for m in ms], exp[:, 0], atol=2e-6)  # np.matmul vs fma-chain round-off
    # a subclass with its own score hook takes the generic route and sees the similarity matrix
    class BoxMean(VCSLLocalizationMaxSim):
        def score(self, candidate, match, box, similarity):
            x1, y1, x2, y2 = box
            return float(similarity[x1:x2, y1:y2].mean())

    ms = BoxMean(q, r, "TN", similarity_bias=bias, **kw).localize_all(cands[:40])
    ref = VCSLLocalizationMaxSim(q, r, "TN", similarity_bias=bias, **kw).localize_all(cands[:40])
    assert [(m.query_start, m.ref_end) for m in ms] == [(m.query_start, m.ref_end) for m in ref]


def test_end_to_end_files_and_metrics(gpu, tmp_path):
    """evaluate_descriptor_track on .npz/.csv files + the matching flow; uAP and segment AP vs the
    reference within 1e-4 (north_star), candidate (query, ref) sets identical."""
    from vsc2022_amd.vsc.baseline import sscd_baseline
    from vsc2022_amd.vsc.descriptor_eval_lib import evaluate_descriptor_track
    from vsc2022_amd.vsc.index import VideoFeature
    from vsc2022_amd.vsc.metrics import Match, evaluate_matching_track
    from vsc2022_amd.vsc.storage import store_features

    fx = load("g6_end_to_end")
    q, r = videos(fx, "q", VideoFeature), videos(fx, "r", VideoFeature)
    store_features(str(tmp_path / "q.npz"), q)
    store_features(str(tmp_path / "r.npz"), r)
    gts = [Match(str(a), str(b), 1.0, *row) for a, b, row in zip(fx["gt_q"], fx["gt_r"], fx["gt_rows"])]
    Match.write_csv(gts, str(tmp_path / "gt.csv"))
    ap, cands = evaluate_descriptor_track(str(tmp_path / "q.npz"), str(tmp_path / "r.npz"), str(tmp_path / "gt.csv"))
    assert [c.query_id for c in cands] == list(fx["cand_q"]) and [c.ref_id for c in cands] == list(fx["cand_r"])
    assert np.array_equal(bits([c.score for c in cands]), bits(fx["cand_s"]))
    assert abs(ap.ap - float(fx["uap"])) < 1e-4 and abs(ap.simple_ap - float(fx["simple_ap"])) < 1e-4
    out = tmp_path / "out"
    from vsc2022_amd.vsc.storage import load_features
    from vsc2022_amd.vsc.metrics import Dataset

    cand_file, match_file = sscd_baseline.match(load_features(str(tmp_path / "q.npz"), Dataset.QUERIES),
                                                load_features(str(tmp_path / "r.npz"), Dataset.REFS), str(out))
    metrics = evaluate_matching_track(str(tmp_path / "gt.csv"), match_file)
    assert abs(metrics.segment_ap.ap - float(fx["segment_ap"])) < 1e-4
    got = Match.read_csv(match_file)
    assert [(m.query_id, m.ref_id) for m in got] == list(zip(fx["match_m_q"], fx["match_m_r"]))
    assert np.allclose([[m.query_start, m.query_end, m.ref_start, m.ref_end] for m in got], fx["match_m_rows"][:, 1:])
