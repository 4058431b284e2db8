"""`VCSLLocalizationDnS` against the REFERENCE's class.

Fixtures g9_dns_* were produced by importing /root/reference/vsc/baseline/dns_baseline.py:108-163 unmodified
(oracle/gen_golden.py:gen_g9) with a seeded stand-in for the fine-grained student (tests/helpers.py:dns_standin):
the similarity matrix of every pair and the Match rows of `localize_all`.  The mirror gets the fine descriptors the
way the reference's callers pass them -- Dict[str, VideoFeature] (reference :183-186, 260-264) -- and as a list.
The coarse part of the matrix comes from the GPU (fp32 fma chain) where the reference calls np.matmul: matrices
agree to 2e-6, boxes must be identical, MaxSim scores within 4e-6.
"""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def build(tag, as_dict, device):
    from vsc2022_amd.vsc.baseline.dns_baseline import VCSLLocalizationDnS
    from vsc2022_amd.vsc.index import VideoFeature
    from vsc2022_amd.vsc.metrics import CandidatePair
    from vsc2022_amd import synth

    fx = helpers.load("g9_dns_" + tag)
    fg_type = str(fx["fg_type"])
    q, r, qfine, rfine = helpers.g9_inputs(fg_type)
    qc, rc = synth.to_video_features(q, VideoFeature), synth.to_video_features(r, VideoFeature)
    qf = [VideoFeature(video_id=v.video_id, timestamps=v.timestamps, feature=f) for v, f in zip(qc, qfine)]
    rf = [VideoFeature(video_id=v.video_id, timestamps=v.timestamps, feature=f) for v, f in zip(rc, rfine)]
    if as_dict:
        qf, rf = {v.video_id: v for v in qf}, {v.video_id: v for v in rf}
    loc = VCSLLocalizationDnS(helpers.dns_standin(fg_type), qf, rf, qc, rc, model_type="TN", tn_max_step=5, min_length=4,
                              concurrency=16, similarity_bias=0.5, device=device, symmetric=bool(fx["symmetric"]),
                              geometric_mean=bool(fx["geometric_mean"]))
    cands = [CandidatePair(str(a), str(b), float(s)) for a, b, s in zip(fx["cand_q"], fx["cand_r"], fx["cand_s"])]
    return fx, loc, cands


@pytest.mark.parametrize("tag", ["att", "bin", "att_plain"])
@pytest.mark.parametrize("as_dict", [True, False])
def test_dns_localization_matches_the_reference_class(gpu, tag, as_dict):
    fx, loc, cands = build(tag, as_dict, "cpu")
    cuts = np.r_[0, np.cumsum(fx["sim_shapes"].prod(axis=1))]
    for k, c in enumerate(cands):
        want = fx["sims"][cuts[k]:cuts[k + 1]].reshape(fx["sim_shapes"][k])
        got = loc.similarity(c)
        assert got.shape == want.shape and np.abs(got - want).max() < 2e-6, (k, np.abs(got - want).max())
    matches = loc.localize_all(cands)
    rows = fx["m_rows"]
    assert len(matches) == len(rows)
    assert [str(m.query_id) for m in matches] == list(fx["m_q"]) and [str(m.ref_id) for m in matches] == list(fx["m_r"])
    got = np.array([[m.score, m.query_start, m.query_end, m.ref_start, m.ref_end] for m in matches], dtype=np.float64)
    assert np.array_equal(got[:, 1:], rows[:, 1:])
    assert np.abs(got[:, 0] - rows[:, 0]).max() < 4e-6
    # one pair at a time (Localization.localize) gives the same rows
    one = [m for c in cands[:5] for m in loc.localize(c)]
    assert [(m.query_id, m.ref_id, m.query_start, m.ref_end) for m in one] == \
           [(m.query_id, m.ref_id, m.query_start, m.ref_end) for m in matches[: len(one)]]


def test_dns_fine_network_on_the_gpu(gpu):
    """The fine network on the MI355X through PyTorch-ROCm ("cuda"): same matrices to 2e-6."""
    fx, loc, cands = build("att", True, "cuda")
    cuts = np.r_[0, np.cumsum(fx["sim_shapes"].prod(axis=1))]
    for k in (0, 7, 20, 41):
        want = fx["sims"][cuts[k]:cuts[k + 1]].reshape(fx["sim_shapes"][k])
        assert np.abs(loc.similarity(cands[k]) - want).max() < 2e-6
    assert len(loc.localize_all(cands)) > 0
