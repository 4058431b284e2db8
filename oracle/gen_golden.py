#!/usr/bin/env python3
"""Generates tests/golden/*.npz by running the REFERENCE's own Python (imported unmodified from
/root/reference) over the oracle shims (oracle/faiss_shim, oracle/vcsl_shim).

Runs only in the development container (the reference never travels to the GPU box); the fixtures
it writes are data -- seeded inputs and the reference's outputs -- and are committed together with
this script.  Usage:  python oracle/gen_golden.py [--check]   (--check: regenerate in memory and
compare with the committed files instead of writing).

Fixtures
  g1_candidates        the known-answer case of tests/test_candidates.py:15-83
  g2_search_*          VideoIndex.search / _global_threshold_knn_search / CandidateGeneration.query on
                       seeded sets (with and without exact ties, IP and L2, several K, k-NN)
  g3_storage           .npz schema of vsc/storage.py (dtypes, shapes) after a store/load round trip
  g4_score_norm        score_normalize(): chosen low-variance dim, adapted query/ref descriptors
  g5_localization_*    VCSLLocalizationMaxSim / CandidateScore .localize_all -> Match rows
  g6_end_to_end        evaluate_descriptor_track-equivalent flow + matching flow: uAP, segment AP
  g7_metrics           random matches -> match_metric / average_precision values of vsc/metrics.py
  g9_dns_*             VCSLLocalizationDnS (vsc/baseline/dns_baseline.py:108-163) imported from the reference with a seeded
                       stand-in for the fine-grained student (tests/helpers.py:dns_standin), fg_type "att" and "bin":
                       similarity matrices of every pair + Match rows of localize_all
  g8_config1_pipeline  BASELINE configs[0] shape (50 x 20 vs 50 x 20 rows, 512-d, K = 60 000), through FILES and
                       the reference's own entry points: evaluate_descriptor_track() and
                       sscd_baseline.main() WITH --score_norm_features (beta 1.2, bias 0.5, MaxSim TN) and
                       --ground_truth.  Inputs are regenerated from the seed by vsc2022_amd/synth.py; the
                       fixture holds their checksum and the reference's outputs.
"""
import argparse
import hashlib
import io
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REFERENCE = "/root/reference"
# shims first: /root/reference/vcsl/__init__.py is an empty stub that would shadow ours
sys.path[:0] = [os.path.join(HERE, "faiss_shim"), os.path.join(HERE, "vcsl_shim"), HERE, REFERENCE, ROOT]

import faiss  # noqa: E402  (the shim)
from vsc.baseline.localization import VCSLLocalizationCandidateScore, VCSLLocalizationMaxSim  # noqa: E402
from vsc.baseline.score_normalization import score_normalize  # noqa: E402
from vsc.candidates import CandidateGeneration, MaxScoreAggregation  # noqa: E402
from vsc.index import VideoFeature, VideoIndex  # noqa: E402
from vsc.metrics import CandidatePair, Match, average_precision, match_metric  # noqa: E402
from vsc.storage import load_features, store_features  # noqa: E402

assert faiss.__file__.startswith(HERE), faiss.__file__
import vsc  # noqa: E402

assert vsc.__file__.startswith(REFERENCE), vsc.__file__

from vsc2022_amd import synth  # noqa: E402  (pure numpy generator shared with tests/bench)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def vf(videos):
    return synth.to_video_features(videos, VideoFeature)


def flat(videos):
    return dict(
        ids=np.array([v.video_id for v in videos]),
        lens=np.array([len(v.feature) for v in videos], dtype=np.int64),
        feats=np.concatenate([v.feature for v in videos]).astype(np.float32),
        ts=np.concatenate([v.timestamps for v in videos]).astype(np.float32),
    )


def pack_pairmatches(pms):
    """List[PairMatches] -> arrays (pair ids in order, per-pair run lengths, match rows)."""
    rows = [(m.query_timestamps[0], m.query_timestamps[1], m.ref_timestamps[0], m.ref_timestamps[1], m.score)
            for pm in pms for m in pm.matches]
    return dict(
        pm_q=np.array([str(pm.query_id) for pm in pms]),
        pm_r=np.array([str(pm.ref_id) for pm in pms]),
        pm_n=np.array([len(pm.matches) for pm in pms], dtype=np.int64),
        pm_rows=np.array(rows, dtype=np.float64).reshape(-1, 5),
        pm_score32=np.array([r[4] for r in rows], dtype=np.float32),
    )


def search_case(name, queries, refs, Ks, knn_ks, metric, dim):
    out = {}
    q, r = flat(queries), flat(refs)
    for k, v in q.items():
        out["q_" + k] = v
    for k, v in r.items():
        out["r_" + k] = v
    out["metric"] = np.int64(metric)
    out["Ks"] = np.array(Ks, dtype=np.int64)
    out["knn_ks"] = np.array(knn_ks, dtype=np.int64)
    qf, rf = vf(queries), vf(refs)
    for K in Ks:
        index = VideoIndex(dim, "Flat", metric)
        index.add(rf)
        raw = index._global_threshold_knn_search(np.concatenate([x.feature for x in qf]), K)
        out[f"K{K}_i"] = np.array([t[0] for t in raw], dtype=np.int64)
        out[f"K{K}_j"] = np.array([t[1] for t in raw], dtype=np.int64)
        out[f"K{K}_s"] = np.array([t[2] for t in raw], dtype=np.float32)
        for kk, vv in pack_pairmatches(index.search(qf, K)).items():
            out[f"K{K}_{kk}"] = vv
        if metric == faiss.METRIC_INNER_PRODUCT:
            cands = CandidateGeneration(rf, MaxScoreAggregation()).query(qf, K)
            out[f"K{K}_cand_q"] = np.array([str(c.query_id) for c in cands])
            out[f"K{K}_cand_r"] = np.array([str(c.ref_id) for c in cands])
            out[f"K{K}_cand_s"] = np.array([c.score for c in cands], dtype=np.float32)
    for k in knn_ks:
        index = VideoIndex(dim, "Flat", metric)
        index.add(rf)
        for kk, vv in pack_pairmatches(index.search(qf, -k)).items():
            out[f"knn{k}_{kk}"] = vv
    return name, out


def gen_g1():
    queries = [VideoFeature(video_id=1, feature=np.eye(3, dtype=np.float32), timestamps=np.array([0.0, 1.0, 2.0]))]
    r5 = np.zeros((5, 3), np.float32)
    r5[2, 1], r5[3, 1] = 1, 2
    r8 = np.zeros((3, 3), np.float32)
    r8[1, 0] = r8[2, 0] = 1
    r10 = np.zeros((3, 3), np.float32)
    r10[1, 2] = 0.25
    refs = [
        VideoFeature(video_id=5, feature=r5, timestamps=np.array([2.0, 4.0, 6.0, 8.0, 10.0])),
        VideoFeature(video_id=8, feature=r8, timestamps=np.array([[0.0, 5.0], [5.0, 10.0], [10.0, 15.0]])),
        VideoFeature(video_id=10, feature=r10, timestamps=np.array([0.0, 0.1, 0.2])),
    ]
    cands = CandidateGeneration(refs, MaxScoreAggregation()).query(queries, 6)
    # the reference test's own expectation
    assert cands == [CandidatePair(1, 5, 2.0), CandidatePair(1, 8, 1.0), CandidatePair(1, 10, 0.25)]
    return "g1_candidates", dict(
        q_feat=queries[0].feature, r5=r5, r8=r8, r10=r10,
        cand_q=np.array([c.query_id for c in cands]), cand_r=np.array([c.ref_id for c in cands]),
        cand_s=np.array([c.score for c in cands], dtype=np.float32))


def gen_g2():
    cases = []
    q, r, _ = synth.make_dataset(seed=10, n_query=12, n_ref=40, dim=64, q_frames=(8, 40), r_frames=(8, 40),
                                 planted_frac=0.4)
    cases.append(search_case("g2_search_plain", q, r, [1, 500, 1200 * 12], [1, 5], faiss.METRIC_INNER_PRODUCT, 64))
    q, r, _ = synth.make_dataset(seed=11, n_query=12, n_ref=30, dim=32, q_frames=(8, 30), r_frames=(8, 30),
                                 planted_frac=0.4, static_frac=0.3)
    # exact duplicates across videos too
    r[3].feature[:] = r[4].feature[: len(r[3].feature)] if len(r[4].feature) >= len(r[3].feature) else r[3].feature
    cases.append(search_case("g2_search_ties", q, r, [1, 300, 4000, 1200 * 12], [1, 5], faiss.METRIC_INNER_PRODUCT, 32))
    q, r, _ = synth.make_dataset(seed=12, n_query=6, n_ref=12, dim=512, q_frames=(12, 12), r_frames=(12, 12),
                                 planted_frac=0.5)
    cases.append(search_case("g2_search_d512", q, r, [50, 1200 * 6], [5], faiss.METRIC_INNER_PRODUCT, 512))
    q, r, _ = synth.make_dataset(seed=13, n_query=8, n_ref=10, dim=16, q_frames=(5, 20), r_frames=(5, 20),
                                 planted_frac=0.5, static_frac=0.2)
    cases.append(search_case("g2_search_l2", q, r, [1, 200, 3000], [1, 3], faiss.METRIC_L2, 16))
    return cases


def gen_g3():
    q, _, _ = synth.make_dataset(seed=20, n_query=4, n_ref=2, dim=8, q_frames=(2, 5), r_frames=(2, 5))
    feats = vf(q)
    feats[1] = VideoFeature(video_id=feats[1].video_id, feature=feats[1].feature,
                            timestamps=feats[1].timestamps[:, 0].copy())  # still mixed? no: per file one shape
    feats = [VideoFeature(video_id=f.video_id, feature=f.feature, timestamps=f.timestamps if f.timestamps.ndim == 2
                          else np.stack([f.timestamps, f.timestamps + 1], 1)) for f in feats]
    buf = io.BytesIO()
    store_features(buf, feats)
    raw = buf.getvalue()
    data = np.load(io.BytesIO(raw), allow_pickle=False)
    loaded = load_features(io.BytesIO(raw))
    return "g3_storage", dict(
        video_ids=data["video_ids"], features=data["features"], timestamps=data["timestamps"],
        loaded_ids=np.array([v.video_id for v in loaded]), loaded_lens=np.array([len(v) for v in loaded]))


def gen_g4():
    q, r, _ = synth.make_dataset(seed=30, n_query=6, n_ref=8, dim=24, q_frames=(4, 10), r_frames=(4, 10))
    rng = np.random.default_rng(31)
    noise = synth.make_videos(rng, 10, 24, (5, 12), "N")
    # un-normalised inputs so that the L2 step matters; one dimension with tiny variance
    for vids in (q, r, noise):
        for v in vids:
            v.feature *= rng.uniform(0.5, 2.0, size=(len(v.feature), 1)).astype(np.float32)
            v.feature[:, 7] = 0.01 * v.feature[:, 7]
    out = {}
    for k, v in flat(q).items():
        out["q_" + k] = v
    for k, v in flat(r).items():
        out["r_" + k] = v
    for k, v in flat(noise).items():
        out["n_" + k] = v
    for tag, kw in (("b10", dict(beta=1.0)), ("b12", dict(beta=1.2)), ("b12_keepdim", dict(beta=1.2, replace_dim=False))):
        aq, ar = score_normalize(vf(q), vf(r), vf(noise), **kw)
        out[f"{tag}_q"] = np.concatenate([v.feature for v in aq]).astype(np.float32)
        out[f"{tag}_r"] = np.concatenate([v.feature for v in ar]).astype(np.float32)
    sn = np.concatenate([v.feature for v in noise])
    out["low_var_dim"] = np.int64(sn.var(axis=0).argmin())
    return "g4_score_norm", out


def match_rows(matches):
    return dict(
        m_q=np.array([str(m.query_id) for m in matches]), m_r=np.array([str(m.ref_id) for m in matches]),
        m_rows=np.array([[m.score, m.query_start, m.query_end, m.ref_start, m.ref_end] for m in matches],
                        dtype=np.float64).reshape(-1, 5),
        m_score32=np.array([m.score for m in matches], dtype=np.float32))


def gen_g5():
    cases = []
    for tag, seed, kw, bias in (("default", 40, {}, 0.0), ("ref_params", 41, dict(tn_max_step=5, min_length=4), 0.5),
                                ("ref_params_nobias", 42, dict(tn_max_step=5, min_length=4, concurrency=16), 0.0)):
        q, r, gts = synth.make_dataset(seed=seed, n_query=8, n_ref=8, dim=64, q_frames=(3, 60), r_frames=(3, 70),
                                       planted_frac=1.0, noise=0.03, copy_len=(8, 40))
        # a second planted segment in one pair (multi-segment) and a very short video (Lr < top_k)
        q[0].feature[-6:] = r[1].feature[:6] if len(r[1].feature) >= 6 and len(q[0].feature) >= 6 else q[0].feature[-6:]
        r[2].feature = r[2].feature[:3]
        r[2].timestamps = r[2].timestamps[:3]
        qf, rf = vf(q), vf(r)
        cands = [CandidatePair(a.video_id, b.video_id, float(np.float32(0.1 * k)))
                 for k, (a, b) in enumerate((a, b) for a in qf for b in rf)]
        out = {}
        for k, v in flat(q).items():
            out["q_" + k] = v
        for k, v in flat(r).items():
            out["r_" + k] = v
        out["bias"] = np.float64(bias)
        out["kw_keys"] = np.array(sorted(kw))
        out["kw_vals"] = np.array([kw[k] for k in sorted(kw)], dtype=np.int64)
        out["cand_q"] = np.array([c.query_id for c in cands])
        out["cand_r"] = np.array([c.ref_id for c in cands])
        out["cand_s"] = np.array([c.score for c in cands], dtype=np.float32)
        loc = VCSLLocalizationMaxSim(qf, rf, "TN", similarity_bias=bias, **kw)
        for k, v in match_rows(loc.localize_all(cands)).items():
            out["maxsim_" + k] = v
        loc2 = VCSLLocalizationCandidateScore(qf, rf, "TN", **kw)
        for k, v in match_rows(loc2.localize_all(cands)).items():
            out["candscore_" + k] = v
        cases.append((f"g5_localization_{tag}", out))
    return cases


def gen_g6():
    """Descriptor-track and matching-track flows on a planted-copy set (the flows of
    vsc/descriptor_eval_lib.py:27-60 and vsc/baseline/sscd_baseline.py:90-176, without files)."""
    q, r, gts = synth.make_dataset(seed=50, n_query=30, n_ref=40, dim=64, q_frames=(10, 40), r_frames=(10, 40),
                                   planted_frac=0.4, static_frac=0.05)
    qf, rf = vf(q), vf(r)
    gt_matches = [Match(g.query_id, g.ref_id, 1.0, g.query_start, g.query_end, g.ref_start, g.ref_end) for g in gts]
    cands = CandidateGeneration(rf, MaxScoreAggregation()).query(qf, 1200 * len(qf))[: 25 * len(qf)]
    ap = average_precision(CandidatePair.from_matches(gt_matches), cands)
    loc = VCSLLocalizationCandidateScore(qf, rf, "TN", tn_max_step=5, min_length=4, concurrency=16)
    matches = loc.localize_all(cands[: 5 * len(qf)])
    seg = match_metric(gt_matches, matches)
    out = {}
    for k, v in flat(q).items():
        out["q_" + k] = v
    for k, v in flat(r).items():
        out["r_" + k] = v
    out["gt_q"] = np.array([g.query_id for g in gts])
    out["gt_r"] = np.array([g.ref_id for g in gts])
    out["gt_rows"] = np.array([[g.query_start, g.query_end, g.ref_start, g.ref_end] for g in gts], dtype=np.float64)
    out["cand_q"] = np.array([c.query_id for c in cands])
    out["cand_r"] = np.array([c.ref_id for c in cands])
    out["cand_s"] = np.array([c.score for c in cands], dtype=np.float32)
    out["uap"] = np.float64(ap.ap)
    out["simple_ap"] = np.float64(ap.simple_ap)
    for k, v in match_rows(matches).items():
        out["match_" + k] = v
    out["segment_ap"] = np.float64(seg.ap)
    return "g6_end_to_end", out


def gen_g7():
    """Random ground truth / predictions -> the reference's metric values (pins the metrics mirror)."""
    rng = np.random.default_rng(60)
    out = {}
    for case in range(4):
        n_gt, n_pred = int(rng.integers(5, 30)), int(rng.integers(10, 120))
        def rows(n, with_score):
            q = rng.integers(0, 6, n)
            r = rng.integers(0, 6, n)
            qs = rng.uniform(0, 50, n).round(1)
            rs = rng.uniform(0, 50, n).round(1)
            ql = rng.uniform(1, 20, n).round(1)
            rl = rng.uniform(1, 20, n).round(1)
            sc = (rng.integers(0, 20, n) / 10.0) if with_score else np.ones(n)
            return q, r, sc, qs, qs + ql, rs, rs + rl
        g = rows(n_gt, False)
        pr = rows(n_pred, True)
        gts = [Match(f"Q{a:06d}", f"R{b:06d}", float(c), float(d), float(e), float(f), float(h)) for a, b, c, d, e, f, h in zip(*g)]
        preds = [Match(f"Q{a:06d}", f"R{b:06d}", float(c), float(d), float(e), float(f), float(h)) for a, b, c, d, e, f, h in zip(*pr)]
        seg = match_metric(gts, preds)
        gt_pairs = CandidatePair.from_matches(gts)
        pred_pairs = CandidatePair.from_matches(preds)
        ap = average_precision(gt_pairs, pred_pairs)
        out[f"c{case}_gt"] = np.stack([np.asarray(x, dtype=np.float64) for x in g], 1)
        out[f"c{case}_pred"] = np.stack([np.asarray(x, dtype=np.float64) for x in pr], 1)
        out[f"c{case}_segment_ap"] = np.float64(seg.ap)
        out[f"c{case}_curve_p"] = seg.pr_curve.precisions
        out[f"c{case}_curve_r"] = seg.pr_curve.recalls
        out[f"c{case}_uap"] = np.float64(ap.ap)
        out[f"c{case}_simple_ap"] = np.float64(ap.simple_ap)
    return "g7_metrics", out


G8 = dict(seed=80, n_query=50, n_ref=50, dim=512, q_frames=(20, 20), r_frames=(20, 20), planted_frac=0.2, noise=0.05,
          copy_len=(8, 20))
G8_NOISE = dict(seed=81, n_videos=30, frames=(20, 20))


def g8_inputs():
    """(queries, refs, noise, gts) of the config-1 fixture; shared with tests/test_gpu_config1.py via synth."""
    q, r, gts = synth.make_dataset(**G8)
    noise = synth.make_videos(np.random.default_rng(G8_NOISE["seed"]), G8_NOISE["n_videos"], G8["dim"],
                              G8_NOISE["frames"], "R")
    for k, v in enumerate(noise):  # loaded as Dataset.REFS by the reference, and must not share ids with the refs
        v.video_id = f"R{900000 + k:06d}"
    return q, r, noise, gts


def inputs_digest(*video_lists):
    h = hashlib.sha256()
    for vids in video_lists:
        for v in vids:
            h.update(str(v.video_id).encode())
            h.update(np.ascontiguousarray(v.feature, dtype=np.float32).tobytes())
            h.update(np.ascontiguousarray(v.timestamps, dtype=np.float32).tobytes())
    return h.hexdigest()


def gen_g8():
    import matplotlib

    matplotlib.use("Agg")
    from vsc.baseline import sscd_baseline as ref_sscd
    from vsc.descriptor_eval_lib import evaluate_descriptor_track
    from vsc.metrics import Dataset, evaluate_matching_track

    assert ref_sscd.__file__.startswith(REFERENCE)
    q, r, noise, gts = g8_inputs()
    out = {"digest": np.array(inputs_digest(q, r, noise))}
    with tempfile.TemporaryDirectory() as tmp:
        qp, rp, npth, gtp = (os.path.join(tmp, n) for n in ("q.npz", "r.npz", "noise.npz", "gt.csv"))
        store_features(qp, vf(q))
        store_features(rp, vf(r))
        store_features(npth, vf(noise))
        Match.write_csv([Match(g.query_id, g.ref_id, 1.0, g.query_start, g.query_end, g.ref_start, g.ref_end)
                         for g in gts], gtp)
        # 1. descriptor track (vsc/descriptor_eval_lib.py:27-60)
        ap, cands = evaluate_descriptor_track(qp, rp, gtp)
        out["desc_uap"], out["desc_simple_ap"] = np.float64(ap.ap), np.float64(ap.simple_ap)
        out["desc_cand_q"] = np.array([str(c.query_id) for c in cands])
        out["desc_cand_r"] = np.array([str(c.ref_id) for c in cands])
        out["desc_cand_s"] = np.array([c.score for c in cands], dtype=np.float32)
        # 2. the matching baseline with score normalisation (vsc/baseline/sscd_baseline.py:185-231)
        outdir = os.path.join(tmp, "out")
        args = ref_sscd.parser.parse_args(["--query_features", qp, "--ref_features", rp, "--score_norm_features", npth,
                                           "--output_path", outdir, "--ground_truth", gtp])
        ref_sscd.main(args)
        sn_cands = CandidatePair.read_csv(os.path.join(outdir, "candidates.csv"))
        matches = Match.read_csv(os.path.join(outdir, "matches.csv"))
        out["sn_cand_q"] = np.array([str(c.query_id) for c in sn_cands])
        out["sn_cand_r"] = np.array([str(c.ref_id) for c in sn_cands])
        out["sn_cand_s"] = np.array([c.score for c in sn_cands], dtype=np.float64)
        for k, v in match_rows(matches).items():
            out["sn_match_" + k] = v
        gt_pairs = CandidatePair.from_matches(Match.read_csv(gtp, is_gt=True))
        out["sn_uap"] = np.float64(average_precision(gt_pairs, sn_cands).ap)
        out["sn_segment_ap"] = np.float64(evaluate_matching_track(gtp, os.path.join(outdir, "matches.csv")).segment_ap.ap)
        # a sample of the score-normalised descriptors the reference stored (every 20th row)
        for tag, ds in (("sn_queries", Dataset.QUERIES), ("sn_refs", Dataset.REFS)):
            feats = np.concatenate([v.feature for v in load_features(os.path.join(outdir, tag + ".npz"), ds)])
            out[tag + "_shape"] = np.array(feats.shape, dtype=np.int64)
            out[tag + "_sample"] = feats[::20].astype(np.float32)
        out["files"] = np.array(sorted(os.listdir(outdir)))
    return "g8_config1_pipeline", out


def gen_g9():
    import matplotlib

    matplotlib.use("Agg")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    from vsc.baseline.dns_baseline import VCSLLocalizationDnS

    import vsc.baseline.dns_baseline as ref_dns
    assert ref_dns.__file__.startswith(REFERENCE)
    cases = []
    for tag, fg_type, kw in (("att", "att", {}), ("bin", "bin", {}),
                             ("att_plain", "att", dict(symmetric=False, geometric_mean=False))):
        q, r, qfine, rfine = helpers.g9_inputs(fg_type)
        qc, rc = vf(q), vf(r)
        # the reference's callers pass the fine descriptors as Dict[str, VideoFeature] (dns_baseline.py:260-264)
        qf = {v.video_id: VideoFeature(video_id=v.video_id, timestamps=v.timestamps, feature=f) for v, f in zip(qc, qfine)}
        rf = {v.video_id: VideoFeature(video_id=v.video_id, timestamps=v.timestamps, feature=f) for v, f in zip(rc, rfine)}
        loc = VCSLLocalizationDnS(helpers.dns_standin(fg_type), qf, rf, qc, rc, model_type="TN", tn_max_step=5,
                                  min_length=4, concurrency=16, similarity_bias=0.5, device="cpu", **kw)
        cands = [CandidatePair(a.video_id, b.video_id, float(np.float32(0.1 * k)))
                 for k, (a, b) in enumerate((a, b) for a in qc for b in rc)]
        sims = [np.asarray(loc.similarity(c), dtype=np.float32) for c in cands]
        out = dict(fg_type=np.array(fg_type), symmetric=np.bool_(kw.get("symmetric", True)),
                   geometric_mean=np.bool_(kw.get("geometric_mean", True)),
                   cand_q=np.array([c.query_id for c in cands]), cand_r=np.array([c.ref_id for c in cands]),
                   cand_s=np.array([c.score for c in cands], dtype=np.float32),
                   sims=np.concatenate([x.ravel() for x in sims]),
                   sim_shapes=np.array([x.shape for x in sims], dtype=np.int64))
        out.update(match_rows(loc.localize_all(cands)))
        cases.append((f"g9_dns_{tag}", out))
    return cases


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    os.makedirs(GOLDEN, exist_ok=True)
    fixtures = [gen_g1()] + gen_g2() + [gen_g3(), gen_g4()] + gen_g5() + [gen_g6(), gen_g7(), gen_g8()] + gen_g9()
    bad = 0
    for name, arrays in fixtures:
        path = os.path.join(GOLDEN, name + ".npz")
        if args.check:
            old = np.load(path, allow_pickle=False)
            same = set(old.files) == set(arrays) and all(
                np.array_equal(old[k], np.asarray(arrays[k])) for k in arrays)
            print(("ok   " if same else "DIFF ") + name)
            bad += 0 if same else 1
        else:
            np.savez_compressed(path, **arrays)
            print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB, {len(arrays)} arrays)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
