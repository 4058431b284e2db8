"""Device-resident score normalisation (engine.score_normalize_device) against the list-of-VideoFeature
mirror of vsc/baseline/score_normalization.py (itself pinned to the reference by fixture g4)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("beta,replace_dim,l2", [(1.2, True, True), (1.0, False, True), (1.0, True, False)])
def test_device_score_normalize_equals_list_mirror(gpu, beta, replace_dim, l2):
    import torch

    from vsc2022_amd.engine import score_normalize_device
    from vsc2022_amd.vsc.baseline.score_normalization import score_normalize
    from vsc2022_amd.vsc.index import VideoFeature

    rng = np.random.default_rng(7)
    dim = 64

    def videos(prefix, lens):
        out = []
        for v, n in enumerate(lens):
            f = rng.standard_normal((n, dim)).astype(np.float32)
            f[:, 17] *= 0.01  # a clear lowest-variance coordinate (no near-tie between fp32 and fp64 variance)
            out.append(VideoFeature(video_id=f"{prefix}{v:05d}", timestamps=np.arange(n, dtype=np.float32), feature=f))
        return out

    q, r, noise = videos("Q", [5, 9, 1, 30]), videos("R", [12, 40, 3]), videos("N", [50, 70, 30])
    q2, r2 = score_normalize(q, r, noise, l2_normalize=l2, replace_dim=replace_dim, beta=beta)
    dev = torch.device("cuda", 0)
    stack = lambda vs: torch.from_numpy(np.concatenate([v.feature for v in vs])).to(dev)
    dq, dr = score_normalize_device(stack(q), stack(r), stack(noise), beta=beta, l2_normalize=l2,
                                    replace_dim=replace_dim)
    want_q = np.concatenate([v.feature for v in q2])
    want_r = np.concatenate([v.feature for v in r2])
    assert dq.shape == want_q.shape and dr.shape == want_r.shape
    assert np.array_equal(dq.cpu().numpy().view(np.uint32), want_q.astype(np.float32).view(np.uint32))
    assert np.array_equal(dr.cpu().numpy().view(np.uint32), want_r.astype(np.float32).view(np.uint32))
