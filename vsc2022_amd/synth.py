"""Seeded synthetic descriptors in the shape of the VSC datasets.

Descriptors are `standard_normal` rows, L2-normalised (cosine == inner product); videos have a
random number of 1 fps frames with [i, i+1] timestamps (vsc/baseline/video_reader/
ffmpeg_video_reader.py:54 of the reference); a fraction of the query videos carries a planted,
noised copy of a reference segment (these are the ground truth), and a fraction of the videos is
static (all frames identical) to exercise exact score ties.
"""
from dataclasses import dataclass
from typing import List, Tuple

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


def _unit_rows(rng, n, dim):
    x = rng.standard_normal((n, dim)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x


def make_videos(rng, n_videos, dim, frames, prefix, static_frac=0.0) -> List[SynthVideo]:
    lo, hi = frames
    out = []
    for v in range(n_videos):
        n = int(rng.integers(lo, hi + 1)) if hi > lo else int(lo)
        if static_frac > 0 and rng.random() < static_frac:
            feat = np.repeat(_unit_rows(rng, 1, dim), n, axis=0)
        else:
            feat = _unit_rows(rng, n, dim)
        ts = np.stack([np.arange(n, dtype=np.float32), np.arange(1, n + 1, dtype=np.float32)], axis=1)
        out.append(SynthVideo(f"{prefix}{v:06d}", ts, feat))
    return out


def make_dataset(seed=0, n_query=50, n_ref=50, dim=512, q_frames=(20, 20), r_frames=(20, 20),
                 planted_frac=0.2, static_frac=0.0, noise=0.05, copy_len=(8, 30)
                 ) -> Tuple[List[SynthVideo], List[SynthVideo], List[SynthGT]]:
    """Queries, refs and the ground-truth copied segments."""
    rng = np.random.default_rng(seed)
    refs = make_videos(rng, n_ref, dim, r_frames, "R", static_frac)
    queries = make_videos(rng, n_query, dim, q_frames, "Q", static_frac)
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
