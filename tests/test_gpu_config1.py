"""GPU: BASELINE configs[0] (descriptor_eval on 50 x 20 vs 50 x 20 rows of 512-d descriptors, K = 60 000) and the
matching baseline WITH score normalisation, through FILES and the command-line entry points, against fixture g8
= the outputs of the reference's own evaluate_descriptor_track() and sscd_baseline.main() on the same inputs
(oracle/gen_golden.py:gen_g8; vsc/descriptor_eval_lib.py:27-60, vsc/baseline/sscd_baseline.py:185-231)."""
import os

import numpy as np
import pytest

from helpers import bits, g8_inputs, load

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    from vsc2022_amd import synth
    from vsc2022_amd.vsc.index import VideoFeature
    from vsc2022_amd.vsc.metrics import Match
    from vsc2022_amd.vsc.storage import store_features

    fx = load("g8_config1_pipeline")
    q, r, noise, gts, digest = g8_inputs()
    assert digest == str(fx["digest"]), "synthetic inputs differ from the ones the fixture was generated on"
    d = tmp_path_factory.mktemp("config1")
    paths = {k: str(d / k) for k in ("q.npz", "r.npz", "noise.npz", "gt.csv", "out", "cands.csv")}
    store_features(paths["q.npz"], synth.to_video_features(q, VideoFeature))
    store_features(paths["r.npz"], synth.to_video_features(r, VideoFeature))
    store_features(paths["noise.npz"], synth.to_video_features(noise, VideoFeature))
    Match.write_csv([Match(g.query_id, g.ref_id, 1.0, g.query_start, g.query_end, g.ref_start, g.ref_end) for g in gts],
                    paths["gt.csv"])
    return fx, paths


def test_descriptor_eval_cli_on_files(gpu, files):
    from vsc2022_amd.cli import descriptor_eval
    from vsc2022_amd.vsc.metrics import CandidatePair

    fx, p = files
    ap, cands = descriptor_eval.main(["--query_features", p["q.npz"], "--ref_features", p["r.npz"],
                                      "--ground_truth", p["gt.csv"], "--candidates_output", p["cands.csv"]])
    assert len(cands) == 25 * 50
    assert [str(c.query_id) for c in cands] == list(fx["desc_cand_q"])
    assert [str(c.ref_id) for c in cands] == list(fx["desc_cand_r"])
    assert np.array_equal(bits([c.score for c in cands]), bits(fx["desc_cand_s"]))  # candidate sets AND scores bit-exact
    assert abs(ap.ap - float(fx["desc_uap"])) < 1e-4 and abs(ap.simple_ap - float(fx["desc_simple_ap"])) < 1e-4
    back = CandidatePair.read_csv(p["cands.csv"])
    assert [(str(c.query_id), str(c.ref_id)) for c in back] == list(zip(fx["desc_cand_q"], fx["desc_cand_r"]))


def test_sscd_baseline_main_with_score_normalisation(gpu, files):
    import matplotlib

    matplotlib.use("Agg")
    from vsc2022_amd.cli import matching_eval
    from vsc2022_amd.vsc.baseline import sscd_baseline
    from vsc2022_amd.vsc.metrics import CandidatePair, Dataset, Match, average_precision
    from vsc2022_amd.vsc.storage import load_features

    fx, p = files
    args = sscd_baseline.build_parser().parse_args(
        ["--query_features", p["q.npz"], "--ref_features", p["r.npz"], "--score_norm_features", p["noise.npz"],
         "--output_path", p["out"], "--ground_truth", p["gt.csv"], "--overwrite"])
    sscd_baseline.main(args)
    assert sorted(os.listdir(p["out"])) == list(fx["files"])
    # the score-normalised descriptors written next to the predictions (drop the low-variance dim, row-L2, -1.2 x
    # the 1-NN similarity to the noise set as extra dim): the reference's sklearn / BLAS values within 2e-6
    for tag, ds in (("sn_queries", Dataset.QUERIES), ("sn_refs", Dataset.REFS)):
        feats = np.concatenate([v.feature for v in load_features(os.path.join(p["out"], tag + ".npz"), ds)])
        assert list(feats.shape) == list(fx[tag + "_shape"])
        assert np.abs(feats[::20] - fx[tag + "_sample"]).max() < 2e-6
    # candidates: same pairs; scores within 2e-6 (they inherit the normalisation's rounding); where the reference
    # orders two pairs whose scores differ by less than that, either order is accepted
    cands = CandidatePair.read_csv(os.path.join(p["out"], "candidates.csv"))
    want = {(a, b): s for a, b, s in zip(fx["sn_cand_q"], fx["sn_cand_r"], fx["sn_cand_s"])}
    got = {(str(c.query_id), str(c.ref_id)): c.score for c in cands}
    assert len(cands) == len(want) == 1250
    missing = set(want) - set(got)
    cut = float(fx["sn_cand_s"][-1])
    assert all(abs(want[k] - cut) < 4e-6 for k in missing), missing  # only pairs sitting on the 25-per-query cut may swap
    for k in set(want) & set(got):
        assert abs(want[k] - got[k]) < 4e-6, (k, want[k], got[k])
    s = np.array([c.score for c in cands])
    assert np.all(s[:-1] >= s[1:])
    gt_pairs = CandidatePair.from_matches(Match.read_csv(p["gt.csv"], is_gt=True))
    assert abs(average_precision(gt_pairs, cands).ap - float(fx["sn_uap"])) < 1e-4
    # matches: MaxSim TN with bias 0.5 on the best 5 pairs per query video
    matches = Match.read_csv(os.path.join(p["out"], "matches.csv"))
    ref_rows = {}
    for a, b, row in zip(fx["sn_match_m_q"], fx["sn_match_m_r"], fx["sn_match_m_rows"]):
        ref_rows.setdefault((a, b), []).append(row)
    got_rows = {}
    for m in matches:
        got_rows.setdefault((str(m.query_id), str(m.ref_id)), []).append(
            [m.score, m.query_start, m.query_end, m.ref_start, m.ref_end])
    same_pairs = set(ref_rows) & set(got_rows)
    n_same = 0
    for k in same_pairs:
        a, b = np.array(ref_rows[k]), np.array(got_rows[k])
        if a.shape == b.shape and np.array_equal(a[:, 1:], b[:, 1:]) and np.abs(a[:, 0] - b[:, 0]).max() < 4e-6:
            n_same += 1
    # the aligner sees similarities that differ from the reference's BLAS values in the last bits: a pair whose
    # per-row top-k flips on such a difference may produce other boxes; everything else must agree exactly
    assert n_same >= 0.97 * len(ref_rows), (n_same, len(ref_rows), len(got_rows))
    seg = matching_eval.main(["--predictions", os.path.join(p["out"], "matches.csv"), "--ground_truth", p["gt.csv"]])
    assert abs(seg.segment_ap.ap - float(fx["sn_segment_ap"])) < 1e-4


# ---------------------------------------------------------------- round 5: the same entry points started as N ranks
def _run_cli(module, argv, nproc=0, port=0):
    """`python -m <module> argv` as one process, or as `nproc` ranks under torch.distributed.run (the ranks share the test
    box's one GPU, so the process group is gloo; on a multi-GPU node the same command line runs over RCCL)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["MPLBACKEND"] = "Agg"
    cmd = [sys.executable]
    if nproc:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    cmd += ["-m", module] + argv
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    return out


@pytest.mark.parametrize("nproc,score_norm", [(2, True), (3, False), (4, True)])
def test_sharded_cli_writes_the_same_files(gpu, files, tmp_path, nproc, score_norm):
    """`python -m torch.distributed.run --nproc-per-node N -m vsc2022_amd.vsc.baseline.sscd_baseline`
    shards the query videos over N ranks, gathers the Match rows and lets rank 0 write candidates.csv / matches.csv (and
    the score-normalised descriptors): the SAME BYTES as the single-process command (vsc/baseline/sscd_baseline.py:155-231;
    the reference uses all GPUs through faiss, vsc/index.py:153)."""
    fx, p = files
    base = ["--query_features", p["q.npz"], "--ref_features", p["r.npz"], "--ground_truth", p["gt.csv"], "--overwrite"]
    if score_norm:
        base += ["--score_norm_features", p["noise.npz"]]
    one, many = str(tmp_path / "one"), str(tmp_path / "many")
    _run_cli("vsc2022_amd.vsc.baseline.sscd_baseline", base + ["--output_path", one])
    log = _run_cli("vsc2022_amd.vsc.baseline.sscd_baseline", base + ["--output_path", many], nproc,
                   29100 + os.getpid() % 400 + nproc)
    assert sorted(os.listdir(one)) == sorted(os.listdir(many))
    for name in ("candidates.csv", "matches.csv"):
        a, b = open(os.path.join(one, name), "rb").read(), open(os.path.join(many, name), "rb").read()
        assert len(a) > 200 and a == b, name
    if score_norm:
        for name in ("sn_queries.npz", "sn_refs.npz"):
            a, b = np.load(os.path.join(one, name)), np.load(os.path.join(many, name))
            assert sorted(a.files) == sorted(b.files)
            for k in a.files:
                assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), (name, k)
    assert "sharded search" in log.stderr   # (the sharded route did run)


def test_sharded_descriptor_eval_cli(gpu, files, tmp_path):
    """descriptor_eval (BASELINE configs[0]) as 3 ranks: the candidate file of the single-process command, byte for byte"""
    fx, p = files
    base = ["--query_features", p["q.npz"], "--ref_features", p["r.npz"], "--ground_truth", p["gt.csv"]]
    one, many = str(tmp_path / "one.csv"), str(tmp_path / "many.csv")
    _run_cli("vsc2022_amd.cli.descriptor_eval", base + ["--candidates_output", one])
    _run_cli("vsc2022_amd.cli.descriptor_eval", base + ["--candidates_output", many], 3, 29600 + os.getpid() % 300)
    a, b = open(one, "rb").read(), open(many, "rb").read()
    assert len(a) > 1000 and a == b
