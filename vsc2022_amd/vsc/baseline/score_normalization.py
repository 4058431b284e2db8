"""Score normalisation against a "noise" descriptor set, on the MI355X engine.

Same entry points as the reference module (`transform_features`, `score_normalize`;
/root/reference/vsc/baseline/score_normalization.py:22-105).  The idea: a query frame that is close
to *everything* (including an unrelated noise set) should be trusted less, so every similarity is
corrected by the query frame's best similarity to the noise set,

    sim'(q, r) = q . r  -  beta * max_n (q . n).

Because the correction depends on the query frame only, it can be folded into the descriptors --
append `-beta * max_n(q . n)` to every query row and `1` to every reference row -- and the search
path stays a plain inner product.  To keep the dimension at 512 the least informative coordinate
(lowest variance over the noise set) is dropped first.

GPU work here: row L2 normalisation (`vsc_row_normalize`) and the 1-nearest-neighbour search of all
query frames against the noise set (`vsc_index_knn`, ONE call instead of one faiss call per video;
rows are independent so the result is the same).  The variance/argmin stays a single host numpy call,
identical to the reference's, so the dropped coordinate is the same by construction.
"""
import dataclasses
import logging
from typing import Callable, List, Sequence, Tuple

import numpy as np

from vsc2022_amd import _lib
from vsc2022_amd.vsc.candidates import CandidateGeneration, MaxScoreAggregation
from vsc2022_amd.vsc.index import VideoFeature

logger = logging.getLogger("score_normalization.py")
logger.setLevel(logging.INFO)


def transform_features(features: List[VideoFeature], transform: Callable) -> List[VideoFeature]:
    """New VideoFeature objects whose descriptor blocks went through `transform`."""
    return [dataclasses.replace(video, feature=transform(video.feature)) for video in features]


def normalize(x: np.ndarray, device=None) -> np.ndarray:
    """Rows scaled to unit L2 norm on the GPU; all-zero rows stay zero (sklearn `normalize` semantics)."""
    x = _lib.f32c(x)
    if x.ndim != 2:
        raise ValueError("normalize expects a 2-D array")
    out = np.empty_like(x)
    if x.shape[0]:
        _lib.check(_lib.lib().vsc_row_normalize(x.ctypes.data, x.shape[0], x.shape[1], _lib.MEM_HOST,
                                                out.ctypes.data, _lib.MEM_HOST,
                                                _lib.default_device() if device is None else device))
    return out


def _split_like(videos: Sequence[VideoFeature], flat: np.ndarray) -> List[VideoFeature]:
    edges = np.cumsum([0] + [len(v) for v in videos])
    return [dataclasses.replace(v, feature=flat[a:b]) for v, a, b in zip(videos, edges[:-1], edges[1:])]


def _stack(videos: Sequence[VideoFeature]) -> np.ndarray:
    return np.concatenate([_lib.f32c(v.feature) for v in videos], axis=0)


def _normalize_videos(videos: List[VideoFeature]) -> List[VideoFeature]:
    """Unit-norm rows for a whole list of videos with one kernel launch."""
    return _split_like(videos, normalize(_stack(videos))) if videos else []


def _drop_column(videos: List[VideoFeature], column: int) -> List[VideoFeature]:
    return transform_features(videos, lambda block: np.delete(block, column, axis=1))


def _append_column(video: VideoFeature, column: np.ndarray) -> VideoFeature:
    return dataclasses.replace(video, feature=np.concatenate([video.feature, column], axis=1))


def score_normalize(
    queries: List[VideoFeature],
    refs: List[VideoFeature],
    score_norm_refs: List[VideoFeature],
    l2_normalize: bool = True,
    replace_dim: bool = True,
    beta: float = 1.0,
) -> Tuple[List[VideoFeature], List[VideoFeature]]:
    """(adapted queries, adapted refs) whose inner product is the score-normalised similarity."""
    if {v.video_id for v in refs} & {v.video_id for v in score_norm_refs}:
        raise Exception(
            "Normalizing on the dataset we're evaluating on is against VSC rules. "
            "An independent dataset is needed."
        )
    if score_norm_refs is not None and replace_dim:
        weakest = np.concatenate([v.feature for v in score_norm_refs], axis=0).var(axis=0).argmin()
        logger.info("dropping coordinate %d (lowest variance over the noise set)", int(weakest))
        queries, refs, score_norm_refs = (_drop_column(group, weakest) for group in (queries, refs, score_norm_refs))
    if l2_normalize:
        queries, refs, score_norm_refs = (_normalize_videos(group) for group in (queries, refs, score_norm_refs))

    # the faiss-like flat index of the noise set, reached the way the reference reaches it
    noise_index = CandidateGeneration(score_norm_refs, MaxScoreAggregation()).index.index
    adapted_queries: List[VideoFeature] = []
    if queries:
        best_noise_sim, _ = noise_index.search(_stack(queries), 1)
        penalty = -beta * best_noise_sim[:, :1]
        edges = np.cumsum([0] + [len(v) for v in queries])
        adapted_queries = [_append_column(v, penalty[a:b]) for v, a, b in zip(queries, edges[:-1], edges[1:])]
    adapted_refs = [_append_column(v, np.ones_like(v.feature[:, :1])) for v in refs]
    logger.info("score normalisation applied to %d query and %d reference videos", len(queries), len(refs))
    return adapted_queries, adapted_refs
