"""CPU: host-side logic of the mirrors (no GPU): C-ABI exports, error behaviour without a device,
metrics / storage restatements against the reference's golden values, container types."""
import io
import os
import re

import numpy as np
import pytest

from helpers import load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    from vsc2022_amd import _lib

    header = open(os.path.join(ROOT, "include", "vscmi.h")).read()
    declared = set(re.findall(r"\b(vsc_[a-z0-9_]+)\s*\(", header))
    declared -= {"vsc_tn_params"}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), name
    assert L.vsc_version() >= 100


def test_no_device_fails_loudly_not_silently():
    """Without a gfx950 device every entry point must raise (there is no CPU fallback)."""
    from vsc2022_amd import _lib

    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    from vsc2022_amd.vsc.index import FlatIndex, VideoFeature, VideoIndex
    from vsc2022_amd.vsc.candidates import CandidateGeneration, MaxScoreAggregation
    from vsc2022_amd.vcsl.vta import build_vta_model

    with pytest.raises(RuntimeError):
        FlatIndex(8)
    with pytest.raises(RuntimeError):
        VideoIndex(8)
    vf = VideoFeature(video_id="R000001", timestamps=np.arange(3.0), feature=np.eye(3, dtype=np.float32))
    with pytest.raises(RuntimeError):
        CandidateGeneration([vf], MaxScoreAggregation())
    with pytest.raises(RuntimeError):
        build_vta_model("TN").forward_sim([("a", np.eye(4, dtype=np.float32))])
    with pytest.raises(NotImplementedError):
        build_vta_model("SPD")   # (a trained detector network: not buildable, vsc2022_amd/vcsl/aligners.py)


def test_video_feature_contract():
    from vsc2022_amd.vsc.index import PairMatch, PairMatches, VideoFeature, VideoMetadata

    with pytest.raises(AssertionError):
        VideoFeature(video_id="Q1", timestamps=np.arange(3.0), feature=np.zeros((4, 2)))
    vf = VideoFeature(video_id=7, timestamps=np.array([[0.0, 1.0], [1.0, 2.5]]), feature=np.zeros((2, 5)))
    assert len(vf) == 2 and vf.dimensions() == 5 and vf.get_timestamps(1) == (1.0, 2.5)
    md = vf.metadata()
    assert isinstance(md, VideoMetadata) and md.video_id == 7
    one_d = VideoMetadata(video_id="x", timestamps=np.array([3.0, 4.0]))
    assert one_d.get_timestamps(1) == (4.0, 4.0)
    pm = PairMatches("Q", "R", [PairMatch((0.0, 1.0), (2.0, 3.0), 0.5)])
    assert list(pm.records()) == [dict(query_id="Q", ref_id="R", query_start=0.0, query_end=1.0,
                                       ref_start=2.0, ref_end=3.0, score=0.5)]


def test_candidate_list_behaves_like_a_list():
    from vsc2022_amd.vsc.candidates import CandidateList
    from vsc2022_amd.vsc.metrics import CandidatePair

    cl = CandidateList(np.array([0, 0, 1]), np.array([2, 0, 1]), np.array([2.0, 1.0, 0.25], np.float32),
                       [1, 4], [5, 8, 10])
    assert len(cl) == 3
    assert cl[0] == CandidatePair(1, 10, 2.0)
    assert cl == [CandidatePair(1, 10, 2.0), CandidatePair(1, 5, 1.0), CandidatePair(4, 8, 0.25)]
    assert cl[:2] == [CandidatePair(1, 10, 2.0), CandidatePair(1, 5, 1.0)]
    assert [c.score for c in cl] == [2.0, 1.0, 0.25]
    assert sorted(cl, key=lambda c: c.score)[0].ref_id == 8
    buf = io.StringIO()
    CandidatePair.write_csv(cl, buf)
    buf.seek(0)
    back = CandidatePair.read_csv(buf)
    assert back[0] == CandidatePair("Q000001", "R000010", 2.0)


def test_storage_round_trip_matches_reference_schema():
    from vsc2022_amd.vsc.index import VideoFeature
    from vsc2022_amd.vsc.metrics import Dataset
    from vsc2022_amd.vsc.storage import load_features, same_value_ranges, store_features

    fx = load("g3_storage")
    loaded = load_features(io.BytesIO(_npz_bytes(fx)))
    assert [v.video_id for v in loaded] == list(fx["loaded_ids"])
    assert [len(v) for v in loaded] == list(fx["loaded_lens"])
    buf = io.BytesIO()
    store_features(buf, loaded)
    again = np.load(io.BytesIO(buf.getvalue()), allow_pickle=False)
    for k in ("video_ids", "features", "timestamps"):
        assert again[k].dtype == fx[k].dtype and np.array_equal(again[k], fx[k]), k
    # integer ids are formatted with the dataset prefix (tests/test_storage.py:28-45 of the reference)
    feats = [VideoFeature(video_id=3, timestamps=np.arange(2.0), feature=np.ones((2, 4), np.float32)),
             VideoFeature(video_id=11, timestamps=np.arange(3.0), feature=np.zeros((3, 4), np.float32))]
    buf = io.BytesIO()
    store_features(buf, feats, Dataset.QUERIES)
    back = load_features(io.BytesIO(buf.getvalue()), Dataset.QUERIES)
    assert [v.video_id for v in back] == ["Q000003", "Q000011"] and back[1].feature.shape == (3, 4)
    assert list(same_value_ranges(["a", "a", "b", "a"])) == [("a", 0, 2), ("b", 2, 3), ("a", 3, 4)]
    with pytest.raises(ValueError):
        bad = io.BytesIO()
        np.savez(bad, video_ids=np.array(["Q000001"] * 3), features=np.zeros((3, 2)), timestamps=np.zeros(2))
        load_features(io.BytesIO(bad.getvalue()))


def _npz_bytes(fx):
    buf = io.BytesIO()
    np.savez(buf, video_ids=fx["video_ids"], features=fx["features"], timestamps=fx["timestamps"])
    return buf.getvalue()


def _matches(arr, cls):
    return [cls(f"Q{int(a):06d}", f"R{int(b):06d}", float(c), float(d), float(e), float(f), float(g))
            for a, b, c, d, e, f, g in arr]


def test_metrics_match_reference_values():
    from vsc2022_amd.vsc.metrics import CandidatePair, Match, average_precision, match_metric

    fx = load("g7_metrics")
    for case in range(4):
        gts = _matches(fx[f"c{case}_gt"], Match)
        preds = _matches(fx[f"c{case}_pred"], Match)
        seg = match_metric(gts, preds)
        assert abs(seg.ap - float(fx[f"c{case}_segment_ap"])) < 1e-12
        assert np.allclose(seg.pr_curve.precisions, fx[f"c{case}_curve_p"], atol=1e-12)
        assert np.allclose(seg.pr_curve.recalls, fx[f"c{case}_curve_r"], atol=1e-12)
        ap = average_precision(CandidatePair.from_matches(gts), CandidatePair.from_matches(preds))
        assert abs(ap.ap - float(fx[f"c{case}_uap"])) < 1e-12
        assert abs(ap.simple_ap - float(fx[f"c{case}_simple_ap"])) < 1e-12


def test_metrics_known_answers():
    """Hand-checkable cases in the spirit of the reference's tests/test_metrics.py."""
    from vsc2022_amd.vsc.metrics import CandidatePair, Intervals, Match, average_precision, match_metric

    assert Intervals([(1, 5), (3, 8), (10, 12)]).total_length() == 9
    assert Intervals([(0, 10)]).intersect_length(Intervals([(5, 20), (-3, 1)])) == 6
    gt = [Match("Q000001", "R000001", 1.0, 0.0, 10.0, 20.0, 30.0)]
    assert match_metric(gt, gt).ap == pytest.approx(1.0)
    half = [Match("Q000001", "R000001", 0.9, 0.0, 5.0, 20.0, 25.0)]
    assert match_metric(gt, half).ap == pytest.approx(0.5)
    miss = [Match("Q000001", "R000001", 0.9, 50.0, 60.0, 70.0, 80.0)]
    assert match_metric(gt, miss).ap == 0.0
    gts = [CandidatePair("Q000001", "R000001", 1.0), CandidatePair("Q000002", "R000002", 1.0)]
    preds = [CandidatePair("Q000001", "R000001", 0.9), CandidatePair("Q000001", "R000005", 0.8),
             CandidatePair("Q000002", "R000002", 0.7)]
    ap = average_precision(gts, preds)
    assert ap.ap == pytest.approx((1.0 + 2.0 / 3.0) / 2.0) and ap.simple_ap == pytest.approx(ap.ap)
    with pytest.raises(AssertionError):
        average_precision(gts + gts[:1], preds)


def test_end_to_end_metrics_from_golden_candidates():
    """uAP / segment AP of the reference flow recomputed by the metrics mirror from the golden
    candidate and match tables (tolerance 1e-4 per north_star; observed ~1e-16)."""
    from vsc2022_amd.vsc.metrics import CandidatePair, Match, average_precision, match_metric

    fx = load("g6_end_to_end")
    gts = [Match(str(q), str(r), 1.0, *row) for q, r, row in zip(fx["gt_q"], fx["gt_r"], fx["gt_rows"])]
    cands = [CandidatePair(str(q), str(r), s) for q, r, s in zip(fx["cand_q"], fx["cand_r"], fx["cand_s"])]
    ap = average_precision(CandidatePair.from_matches(gts), cands)
    assert abs(ap.ap - float(fx["uap"])) < 1e-9 and abs(ap.simple_ap - float(fx["simple_ap"])) < 1e-9
    matches = [Match(str(q), str(r), row[0], *row[1:]) for q, r, row in
               zip(fx["match_m_q"], fx["match_m_r"], fx["match_m_rows"])]
    # the fixture stores rows as float64 while the reference accumulated np.float32 timestamps
    assert abs(match_metric(gts, matches).ap - float(fx["segment_ap"])) < 1e-6


def test_install_aliases_reference_module_names():
    import subprocess
    import sys

    code = ("import vsc2022_amd; vsc2022_amd.install(); import vsc.index, vsc.candidates, vcsl.vta;"
            "from vsc.baseline.localization import VCSLLocalizationMaxSim;"
            "from vsc.candidates import CandidateGeneration, MaxScoreAggregation;"
            "print(vsc.index.VideoIndex.__module__, vcsl.vta.build_vta_model.__module__)")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ["vsc2022_amd.vsc.index", "vsc2022_amd.vcsl.vta"]


def test_install_aliases_the_dns_modules():
    """`vsc.baseline.dns_baseline` / `dns_index` (reference module names) resolve to the mirrors and export the
    reference's names (dns_baseline.py:52-290, dns_index.py:36-180)."""
    import subprocess
    import sys

    code = ("import vsc2022_amd; vsc2022_amd.install();"
            "import vsc.baseline.dns_baseline as a, vsc.baseline.dns_index as b;"
            "assert {'VCSLLocalizationDnS','search','localize_and_verify','match','main','parser'} <= set(dir(a));"
            "assert {'Accelerator','index_videos','main','parser'} <= set(dir(b));"
            "assert b.Accelerator['CUDA'].get_device().type == 'cuda' and b.Accelerator['CPU'].get_device().type == 'cpu';"
            "print(a.__name__, b.__name__)")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ["vsc2022_amd.vsc.baseline.dns_baseline", "vsc2022_amd.vsc.baseline.dns_index"]
