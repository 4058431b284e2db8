"""Seeded synthetic descriptors in the shape of the VSC datasets.

Descriptors are `standard_normal` rows, L2-normalised (cosine == inner product); videos have a
random number of 1 fps frames with [i, i+1] timestamps (vsc/baseline/video_reader/
ffmpeg_video_reader.py:54 of the reference); a fraction of the query videos carries a planted,
noised copy of a reference segment (these are the ground truth), and a fraction of the videos is
static (all frames identical) to exercise exact score ties.

Distribution classes (real SSCD descriptors are neither isotropic nor independent; the reference
itself drops a LOWEST-VARIANCE coordinate, vsc/baseline/score_normalization.py:73-80, because real data has one).
`Geometry(dist, dim, seed)` fixes what queries, references and noise rows share -- cluster centres, a spectrum and its
basis, a mean -- and `make_videos(..., geometry=...)` / `device_rows(...)` draw rows from it:

  "gaussian"  isotropic (the default; `geometry=None` keeps the historical random stream bit for bit -- fixtures)
  "clusters"  a mixture of n_centres directions; a row is sqrt(rho) centre + sqrt(1 - rho) noise with rho (the cosine
              between two rows of one cluster) drawn per cluster from [0.6, 0.9]; a video lives in 1-3 clusters
  "powerlaw"  coordinate k of a random orthogonal basis scaled by k^(-alpha/2): a few directions carry the energy
  "offset"    a non-zero mean (rows share a direction: every cosine is shifted up) + a few dominant coordinates
  "temporal"  AR(1) inside a video (consecutive frames nearly equal: blocky frame x frame similarity matrices)
  "neardup"   isotropic rows; a tenth of the REFERENCE videos are noised copies of other reference videos
"""
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np


@dataclass
class SynthVideo:
    video_id: str
    timestamps: np.ndarray  # [n, 2] float32
    feature: np.ndarray     # [n, dim] float32


@dataclass
class SynthGT:
    query_id: str
    ref_id: str
    query_start: float
    query_end: float
    ref_start: float
    ref_end: float


DISTRIBUTIONS = ("gaussian", "clusters", "powerlaw", "offset", "temporal", "neardup")


def _unit_rows(rng, n, dim):
    x = rng.standard_normal((n, dim)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x


class Geometry:
    """What the rows of one dataset share (module docstring); numpy arrays drawn from `seed` alone, so every process --
    and the numpy and the on-device generators -- sees the same one."""

    def __init__(self, dist: str, dim: int, seed: int = 0, n_centres: int = 2000, rho=(0.6, 0.9), alpha: float = 1.0,
                 mean_cos: float = 0.2, n_dominant: int = 4, dominant_scale: float = 4.0, ar: float = 0.9,
                 dup_frac: float = 0.1, dup_noise: float = 0.02):
        if dist not in DISTRIBUTIONS:
            raise ValueError(f"unknown distribution {dist!r}: one of {DISTRIBUTIONS}")
        self.dist, self.dim, self.seed = dist, int(dim), int(seed)
        self.ar, self.dup_frac, self.dup_noise = float(ar), float(dup_frac), float(dup_noise)
        rng = np.random.default_rng([self.seed, 0x5EED])
        self.centres = self.rho = self.basis = self.scale = self.mean = self.gain = None
        if dist == "clusters":
            self.centres = _unit_rows(rng, int(n_centres), dim)
            self.rho = rng.uniform(rho[0], rho[1], int(n_centres)).astype(np.float32)
        elif dist == "powerlaw":
            qmat, _ = np.linalg.qr(rng.standard_normal((dim, dim)))
            self.basis = np.ascontiguousarray(qmat.astype(np.float32))
            self.scale = (np.arange(1, dim + 1, dtype=np.float64) ** (-0.5 * alpha)).astype(np.float32)
        elif dist == "offset":
            # two rows z + mu with |z|^2 ~ dim have cosine ~ |mu|^2 / (dim + |mu|^2) =: mean_cos
            mu = _unit_rows(rng, 1, dim)[0]
            self.mean = (mu * np.sqrt(dim * mean_cos / (1.0 - mean_cos))).astype(np.float32)
            self.gain = np.ones(dim, dtype=np.float32)
            self.gain[rng.permutation(dim)[: int(n_dominant)]] = float(dominant_scale)

    # ---- numpy rows of one video
    def video_rows(self, rng, n: int) -> np.ndarray:
        dim = self.dim
        if self.dist in ("gaussian", "neardup"):
            return _unit_rows(rng, n, dim)
        z = rng.standard_normal((n, dim)).astype(np.float32)
        if self.dist == "clusters":
            home = rng.integers(0, len(self.centres), int(rng.integers(1, 4)))
            c = home[rng.integers(0, len(home), n)]
            z /= np.linalg.norm(z, axis=1, keepdims=True)
            rho = self.rho[c][:, None]
            x = np.sqrt(rho) * self.centres[c] + np.sqrt(1.0 - rho) * z
        elif self.dist == "powerlaw":
            x = (z * self.scale) @ self.basis.T
        elif self.dist == "offset":
            x = (z + self.mean) * self.gain
        else:  # temporal
            x = z
            a, b = np.float32(self.ar), np.float32(np.sqrt(1.0 - self.ar ** 2))
            for t in range(1, n):
                x[t] = a * x[t - 1] + b * z[t]
        x = x.astype(np.float32)
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        return x


def make_videos(rng, n_videos, dim, frames, prefix, static_frac=0.0, geometry: Optional[Geometry] = None,
                duplicates: bool = False) -> List[SynthVideo]:
    """geometry=None: isotropic rows from the historical random stream (the committed fixtures depend on it).
    duplicates (geometry "neardup", reference side): a tenth of the videos are noised copies of earlier ones."""
    lo, hi = frames
    out = []
    for v in range(n_videos):
        n = int(rng.integers(lo, hi + 1)) if hi > lo else int(lo)
        if static_frac > 0 and rng.random() < static_frac:
            one = _unit_rows(rng, 1, dim) if geometry is None else geometry.video_rows(rng, 1)
            feat = np.repeat(one, n, axis=0)
        elif geometry is None:
            feat = _unit_rows(rng, n, dim)
        elif duplicates and geometry.dist == "neardup" and v > 0 and rng.random() < geometry.dup_frac:
            src = out[int(rng.integers(0, v))].feature
            feat = np.resize(src, (n, dim)).astype(np.float32) + \
                np.float32(geometry.dup_noise) * rng.standard_normal((n, dim)).astype(np.float32)
            feat /= np.linalg.norm(feat, axis=1, keepdims=True)
        else:
            feat = geometry.video_rows(rng, n)
        ts = np.stack([np.arange(n, dtype=np.float32), np.arange(1, n + 1, dtype=np.float32)], axis=1)
        out.append(SynthVideo(f"{prefix}{v:06d}", ts, feat))
    return out


def make_dataset(seed=0, n_query=50, n_ref=50, dim=512, q_frames=(20, 20), r_frames=(20, 20),
                 planted_frac=0.2, static_frac=0.0, noise=0.05, copy_len=(8, 30), dist: str = "gaussian"
                 ) -> Tuple[List[SynthVideo], List[SynthVideo], List[SynthGT]]:
    """Queries, refs and the ground-truth copied segments.  dist: the distribution class of the rows (module docstring;
    "gaussian" is the historical generator, random stream unchanged)."""
    rng = np.random.default_rng(seed)
    geo = None if dist == "gaussian" else Geometry(dist, dim, seed)
    refs = make_videos(rng, n_ref, dim, r_frames, "R", static_frac, geo, duplicates=True)
    queries = make_videos(rng, n_query, dim, q_frames, "Q", static_frac, geo)
    gts: List[SynthGT] = []
    n_planted = int(round(planted_frac * n_query))
    for qv in rng.permutation(n_query)[:n_planted]:
        q = queries[int(qv)]
        r = refs[int(rng.integers(0, n_ref))]
        length = int(min(rng.integers(copy_len[0], copy_len[1] + 1), len(q.feature), len(r.feature)))
        if length < 2:
            continue
        q0 = int(rng.integers(0, len(q.feature) - length + 1))
        r0 = int(rng.integers(0, len(r.feature) - length + 1))
        seg = r.feature[r0 : r0 + length] + noise * rng.standard_normal((length, dim)).astype(np.float32)
        seg /= np.linalg.norm(seg, axis=1, keepdims=True)
        q.feature[q0 : q0 + length] = seg.astype(np.float32)
        gts.append(SynthGT(q.video_id, r.video_id, float(q.timestamps[q0, 0]),
                           float(q.timestamps[q0 + length - 1, 1]), float(r.timestamps[r0, 0]),
                           float(r.timestamps[r0 + length - 1, 1])))
    return queries, refs, gts


def to_video_features(videos: List[SynthVideo], cls):
    """Build the caller's VideoFeature type (the mirror's, or the reference's for goldens)."""
    return [cls(video_id=v.video_id, timestamps=v.timestamps, feature=v.feature) for v in videos]


# ---- the same classes generated in HBM (bench.py, the full-size tests): fixed frames per video, torch's device generator
def device_rows(torch, dev, seed: int, n_vid: int, frames: int, dim: int, static_frac: float = 0.01, dist: str = "gaussian",
                geometry: Optional[Geometry] = None, duplicates: bool = False):
    """[n_vid * frames, dim] unit rows in HBM, a function of (seed, shape, dist, geometry) alone -- every rank of a job,
    whatever the world size, generates the same tensor and slices its share.  dist "gaussian": the generator bench.py
    has always used (same random stream)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    n = n_vid * frames
    x = torch.randn((n, dim), generator=g, device=dev, dtype=torch.float32)
    if dist not in ("gaussian", "neardup"):
        geo = geometry if geometry is not None else Geometry(dist, dim, seed)
        assert geo.dist == dist and geo.dim == dim
        if dist == "clusters":
            centres = torch.from_numpy(geo.centres).to(dev)
            rho_c = torch.from_numpy(geo.rho).to(dev)
            n_home = torch.randint(1, 4, (n_vid,), generator=g, device=dev)
            home = torch.randint(0, centres.shape[0], (n_vid, 3), generator=g, device=dev)
            slot = torch.randint(0, 6, (n_vid, frames), generator=g, device=dev) % n_home.unsqueeze(1)
            c = torch.gather(home, 1, slot).reshape(-1)
            x /= x.norm(dim=1, keepdim=True)
            rho = rho_c[c].unsqueeze(1)
            for a in range(0, n, 1 << 20):   # (chunks: the gathered centres of 2 M rows are a second 4 GB tensor)
                b = min(n, a + (1 << 20))
                x[a:b] = torch.sqrt(rho[a:b]) * centres[c[a:b]] + torch.sqrt(1.0 - rho[a:b]) * x[a:b]
        elif dist == "powerlaw":
            basis_t = torch.from_numpy(np.ascontiguousarray(geo.basis.T)).to(dev)
            scale = torch.from_numpy(geo.scale).to(dev)
            for a in range(0, n, 1 << 20):
                b = min(n, a + (1 << 20))
                x[a:b] = (x[a:b] * scale) @ basis_t
        elif dist == "offset":
            x += torch.from_numpy(geo.mean).to(dev)
            x *= torch.from_numpy(geo.gain).to(dev)
        elif dist == "temporal":
            xv = x.view(n_vid, frames, dim)
            a_, b_ = float(geo.ar), float(np.sqrt(1.0 - geo.ar ** 2))
            for t in range(1, frames):
                xv[:, t] = a_ * xv[:, t - 1] + b_ * xv[:, t]
    x /= x.norm(dim=1, keepdim=True)
    xv = x.view(n_vid, frames, dim)
    if dist == "neardup" and duplicates and n_vid > 1:
        geo = geometry if geometry is not None else Geometry(dist, dim, seed)
        n_dup = int(round(geo.dup_frac * n_vid))
        perm = torch.randperm(n_vid, generator=g, device=dev)
        dst, src = perm[:n_dup], perm[n_dup : 2 * n_dup] if 2 * n_dup <= n_vid else perm[-n_dup:]
        dup = xv[src] + geo.dup_noise * torch.randn((n_dup, frames, dim), generator=g, device=dev)
        xv[dst] = dup / dup.norm(dim=2, keepdim=True)
    n_static = int(round(static_frac * n_vid))
    if n_static:
        vids = torch.randperm(n_vid, generator=g, device=dev)[:n_static]
        xv[vids] = xv[vids, :1].expand(-1, frames, -1).clone()
    return x
