"""Frame inference: model contract on CPU, device hand-off on GPU."""
import numpy as np
import pytest
import torch


def test_sscd_model_contract_cpu():
    from vsc2022_amd.vsc.baseline.inference import SSCDModel, SyntheticVideos, build_sscd_model, run_inference, \
        to_video_features

    model = build_sscd_model(dims=512, seed=0, device="cpu", channels_last=False)
    n_params = sum(p.numel() for p in model.parameters())
    assert 24_000_000 < n_params < 26_000_000  # ResNet-50 trunk (23.5M) + Linear(2048 -> 512)
    src = SyntheticVideos(n_videos=5, frames=(2, 4), size=64, seed=3)
    mine = list(run_inference(model, src, "cpu", batch_size=3, rank=1, world_size=2, channels_last=False))
    assert [i for i, _ in mine] == [1, 3]                       # video_idx % world == rank
    lens = src.lengths()
    for idx, desc in mine:
        assert desc.shape == (lens[idx], 512) and desc.dtype == torch.float32 and torch.isfinite(desc).all()
    vfs = to_video_features(mine, src)
    assert vfs[0].video_id == "Q000001" and vfs[0].timestamps.shape == (lens[1], 2)
    assert np.array_equal(vfs[0].timestamps[:, 1] - vfs[0].timestamps[:, 0], np.ones(lens[1], np.float32))
    # batches never mix videos and batching does not change the result
    again = dict(run_inference(model, src, "cpu", batch_size=32, rank=1, world_size=2, channels_last=False))
    assert torch.allclose(again[1], mine[0][1], atol=1e-5)


@pytest.mark.gpu
def test_inference_feeds_the_engine_on_device(gpu):
    from vsc2022_amd.engine import DeviceMatcher
    from vsc2022_amd.vsc.baseline.inference import SyntheticVideos, build_sscd_model, run_inference, to_flat

    dev = torch.device("cuda", 0)
    model = build_sscd_model(device=dev)
    refs = SyntheticVideos(n_videos=6, frames=(4, 6), size=96, seed=5, prefix="R")
    qs = SyntheticVideos(n_videos=3, frames=(4, 6), size=96, seed=5, prefix="Q")   # same seed: videos 0..2 are copies
    rf, roff, _ = to_flat(run_inference(model, refs, dev, autocast_dtype=torch.bfloat16))
    qf, qoff, _ = to_flat(run_inference(model, qs, dev, autocast_dtype=torch.bfloat16))
    assert rf.is_cuda and rf.shape[1] == 512
    rf = rf / rf.norm(dim=1, keepdim=True)
    qf = qf / qf.norm(dim=1, keepdim=True)
    m = DeviceMatcher(rf, roff, 0)
    m.set_queries(qf, qoff)
    hi, hj, hs, _ = m.search(50)
    pq, pr, ps, _ = m.pair_max(hi, hj, hs)
    best = {}
    for a, b, s in zip(pq.cpu().tolist(), pr.cpu().tolist(), ps.cpu().tolist()):
        best.setdefault(a, (b, s))
    assert all(best[v][0] == v for v in range(3))   # every query video retrieves its own copy first


def test_packed_inference_equals_per_video_cpu():
    from vsc2022_amd.vsc.baseline.inference import SyntheticVideos, build_sscd_model, run_inference, run_inference_packed

    model = build_sscd_model(dims=32, seed=1, device="cpu", channels_last=False)
    src = SyntheticVideos(n_videos=7, frames=(1, 5), size=32, seed=4)
    a = dict(run_inference(model, src, "cpu", batch_size=4, channels_last=False))
    b = list(run_inference_packed(model, src, "cpu", batch_size=6, channels_last=False))
    assert [i for i, _ in b] == sorted(a)
    for idx, desc in b:
        assert desc.shape == a[idx].shape and torch.allclose(desc, a[idx], atol=1e-5)
    c = dict(run_inference_packed(model, src, "cpu", batch_size=6, rank=1, world_size=3, channels_last=False))
    assert sorted(c) == [1, 4] and torch.allclose(c[4], a[4], atol=1e-5)
