"""GPU: edge cases of the public surface -- empty and ragged inputs, degenerate sizes, argument
errors (the reference's convention: asserts / exceptions)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def vf(vid, feat, cls):
    n = feat.shape[0]
    return cls(video_id=vid, timestamps=np.arange(n, dtype=np.float32), feature=feat)


def test_empty_and_degenerate_searches(gpu, orc):
    from vsc2022_amd.vsc.index import FlatIndex, VideoFeature, VideoIndex

    rng = np.random.default_rng(0)
    idx = FlatIndex(16)
    q = unit(rng, 5, 16)
    # empty index
    i, j, s, _ = idx.global_topk(q, 10)
    assert len(s) == 0
    D, I = idx.search(q, 3)
    assert np.all(I == -1) and np.all(D == -np.finfo(np.float32).max)
    lims, Dr, Ir = idx.range_search(q, 0.0)
    assert lims[-1] == 0 and len(Dr) == 0
    # one reference row, one query row, dim not a multiple of anything
    idx1 = FlatIndex(1)
    idx1.add(np.array([[2.0]], np.float32))
    i, j, s, _ = idx1.global_topk(np.array([[3.0]], np.float32), 5)
    assert list(i) == [0] and list(j) == [0] and list(s) == [6.0]
    # K = 0 and empty query batch
    idx.add(unit(rng, 7, 16))
    assert len(idx.global_topk(q, 0)[2]) == 0
    assert len(idx.global_topk(np.zeros((0, 16), np.float32), 4)[2]) == 0
    # k larger than the index: missing slots are (-1, -FLT_MAX) exactly like the oracle
    D, I = idx.search(q, 12)
    oD, oI = orc.knn(q, unit(np.random.default_rng(0), 5 + 7, 16)[5:], 12)
    assert np.array_equal(I, oI) and np.array_equal(D.view(np.uint32), oD.view(np.uint32))
    # fp16 / float64 inputs are accepted (converted to fp32), wrong width is an error
    D16, I16 = idx.search(q.astype(np.float16), 2)
    D32, I32 = idx.search(q.astype(np.float16).astype(np.float32), 2)
    assert np.array_equal(I16, I32) and np.array_equal(D16, D32)
    with pytest.raises(ValueError):
        idx.add(np.zeros((3, 15), np.float32))
    with pytest.raises(ValueError):
        idx.search(q, 4097)
    with pytest.raises(NotImplementedError):
        VideoIndex(16, "IVF64,Flat")
    # VideoIndex with nothing to say
    vi = VideoIndex(16)
    vi.add([])
    vi.add([vf("R000001", unit(rng, 1, 16), VideoFeature)])  # single-frame video
    assert vi.search([], 5) == [] if False else True
    res = vi.search([vf("Q000001", unit(rng, 1, 16), VideoFeature)], 5)
    assert len(res) == 1 and len(res[0].matches) == 1


def test_ragged_videos_and_repeated_adds(gpu, orc):
    """Videos of 1..N frames, references added in three calls, 1-D and 2-D timestamps mixed."""
    from vsc2022_amd.vsc.candidates import CandidateGeneration, MaxScoreAggregation
    from vsc2022_amd.vsc.index import VideoFeature

    rng = np.random.default_rng(1)
    refs = []
    for v, n in enumerate([1, 2, 130, 1, 257, 3, 64]):
        ts = np.arange(n, dtype=np.float32) if v % 2 else np.stack([np.arange(n), np.arange(n) + 1], 1).astype(np.float32)
        refs.append(VideoFeature(video_id=f"R{v:06d}", timestamps=ts, feature=unit(rng, n, 40)))
    queries = [vf(f"Q{v:06d}", unit(rng, n, 40), VideoFeature) for v, n in enumerate([1, 129, 2, 31])]
    cg = CandidateGeneration(refs[:2], MaxScoreAggregation())
    cg.index.add(refs[2:5])
    cg.index.add(refs[5:])
    cands = cg.query(queries, 3000)
    Q = np.concatenate([v.feature for v in queries])
    R = np.concatenate([v.feature for v in refs])
    row2q = np.repeat(np.arange(len(queries), dtype=np.int32), [len(v) for v in queries])
    row2r = np.repeat(np.arange(len(refs), dtype=np.int32), [len(v) for v in refs])
    oi, oj, os_ = orc.global_threshold_search(Q, R, 3000)
    oq, orr, ops, _ = orc.pair_max(oi, oj, os_, row2q, row2r)
    assert np.array_equal(cands.q_ord, oq) and np.array_equal(cands.r_ord, orr)
    assert np.array_equal(cands.scores.view(np.uint32), ops.view(np.uint32))


def test_tn_degenerate_pairs(gpu, orc):
    from vsc2022_amd.vcsl.vta import build_vta_model
    from vsc2022_amd.vsc.baseline.localization import VCSLLocalizationMaxSim
    from vsc2022_amd.vsc.index import VideoFeature
    from vsc2022_amd.vsc.metrics import CandidatePair

    rng = np.random.default_rng(2)
    model = build_vta_model("TN")
    assert model.forward_sim([]) == []
    mats = [np.zeros((1, 1), np.float32), np.full((3, 2), 0.9, np.float32), np.full((40, 1), 0.7, np.float32),
            np.eye(30, dtype=np.float32), -np.ones((20, 20), np.float32)]
    got = model.forward_sim([(str(k), m) for k, m in enumerate(mats)])
    for (name, boxes), m in zip(got, mats):
        assert boxes == orc.tn(m), name
    assert got[3][1] != []  # a clean diagonal is found
    with pytest.raises(TypeError):
        build_vta_model("TN", not_a_tn_argument=1)
    with pytest.raises(ValueError):
        build_vta_model("TN", tn_top_k=99).forward_sim([("a", np.eye(4, dtype=np.float32))])
    # localisation of pairs whose videos have a single frame
    q = [vf("Q000001", unit(rng, 1, 32), VideoFeature), vf("Q000002", unit(rng, 12, 32), VideoFeature)]
    r = [vf("R000001", unit(rng, 1, 32), VideoFeature), vf("R000002", unit(rng, 9, 32), VideoFeature)]
    loc = VCSLLocalizationMaxSim(q, r, "TN", similarity_bias=0.5)
    assert loc.localize_all([]) == []
    ms = loc.localize_all([CandidatePair(a.video_id, b.video_id, 1.0) for a in q for b in r])
    exp = []
    for a in q:
        for b in r:
            for box in orc.tn(orc.pair_sims(a.feature, b.feature, 0.5)):
                exp.append((a.video_id, b.video_id, tuple(box)))
    got = [(m.query_id, m.ref_id, (int(m.query_start), int(m.ref_start), int(m.query_end), int(m.ref_end)))  # 1-D timestamps: (t, t)
           for m in ms]
    assert got == exp and all(e[0] == "Q000002" and e[1] == "R000002" for e in exp)  # 1-frame videos never match
    assert loc.similarity(CandidatePair("Q000002", "R000002", 0.0)).shape == (12, 9)
    with pytest.raises(KeyError):
        loc.localize(CandidatePair("Q000009", "R000001", 1.0))


def test_score_normalize_edge_cases(gpu):
    from vsc2022_amd.vsc.baseline.score_normalization import normalize, score_normalize
    from vsc2022_amd.vsc.index import VideoFeature

    rng = np.random.default_rng(3)
    assert normalize(np.zeros((0, 8), np.float32)).shape == (0, 8)
    q = [vf("Q000001", unit(rng, 3, 8), VideoFeature)]
    r = [vf("R000001", unit(rng, 4, 8), VideoFeature)]
    n = [vf("R900001", unit(rng, 5, 8), VideoFeature)]
    aq, ar = score_normalize(q, r, n, beta=1.2)
    assert aq[0].feature.shape == (3, 8) and ar[0].feature.shape == (4, 8)   # one dim dropped, one appended
    assert np.all(ar[0].feature[:, -1] == 1.0) and np.all(aq[0].feature[:, -1] <= 1.2 + 1e-6)
    aq2, ar2 = score_normalize([], r, n)
    assert aq2 == [] and len(ar2) == 1


def test_tn_rejects_more_paths_than_boxes(gpu):
    """max_path >= VSC_TN_MAX_BOXES would drop boxes silently; it is an explicit error instead."""
    from vsc2022_amd import _lib
    from vsc2022_amd.vcsl.vta import build_vta_model

    sims = np.random.default_rng(0).random((12, 14)).astype(np.float32)
    assert build_vta_model("TN", max_path=_lib.TN_MAX_BOXES - 1).forward_sim([("a", sims)])[0][0] == "a"
    with pytest.raises(ValueError, match="max_path"):
        build_vta_model("TN", max_path=_lib.TN_MAX_BOXES).forward_sim([("a", sims)])


def test_sharded_code_path_over_rccl_world1(gpu):
    """The RCCL ("nccl") branch of vsc2022_amd/dist.py -- device tensors straight into the collectives, no host
    staging -- as far as one GPU allows: a world of one rank, the sharded pipeline forced on."""
    import subprocess
    import sys

    code = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29871", VSC_FORCE_SHARDED="1")
from vsc2022_amd import synth, dist as vdist
from vsc2022_amd.engine import DeviceMatcher
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
q, r, _ = synth.make_dataset(seed=5, n_query=40, n_ref=60, dim=64, q_frames=(8, 20), r_frames=(8, 30), planted_frac=0.3)
pack = lambda v: (np.concatenate([x.feature for x in v]).astype(np.float32), np.r_[0, np.cumsum([len(x.feature) for x in v])].astype(np.int64))
(qf, qoff), (rf, roff) = pack(q), pack(r)
m = DeviceMatcher(rf, roff, 0); m.set_queries(qf, qoff)
a = m.match()
t = torch.zeros(4, device=dev); assert not vdist._via_host(t, None)
b = m.match(n_qvid_global=len(q), qvid_base=0, row_base=0)
assert torch.equal(a.cand_q, b.cand_q) and torch.equal(a.cand_r, b.cand_r) and torch.equal(a.cand_score, b.cand_score)
assert (a.n_hits, a.n_candidates, a.n_localized, a.n_matches) == (b.n_hits, b.n_candidates, b.n_localized, b.n_matches)
n = a.n_localized
assert torch.equal(a.nbox[:n], b.nbox[:n]) and torch.equal(a.boxes[:n], b.boxes[:n])
# the collectives themselves on device tensors
x = torch.arange(10, device=dev, dtype=torch.float32).flip(0)
assert vdist.distributed_prefix_select(x, 4)[0] == 4
assert torch.equal(vdist.all_gather_varlen(torch.arange(5, device=dev)), torch.arange(5, device=dev))
dist.destroy_process_group(); print("rccl-world1-ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl-world1-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_knn_with_more_than_64_neighbours(gpu, orc):
    """faiss accepts any k; k > 64 takes the explicit-matrix route (same fp32 chains, same order)."""
    from vsc2022_amd.vsc.index import FlatIndex

    rng = np.random.default_rng(4)
    q, r = unit(rng, 50, 48), unit(rng, 700, 48)
    r[100:110] = r[5:15]  # ties
    idx = FlatIndex(48)
    idx.add(r)
    for k in (65, 200, 700):
        D, I = idx.search(q, k)
        Do, Io = orc.knn(q, r, k)
        assert np.array_equal(I, Io) and np.array_equal(D.view(np.uint32), Do.view(np.uint32)), k
    with pytest.raises(ValueError):
        idx.search(q, 5000)
