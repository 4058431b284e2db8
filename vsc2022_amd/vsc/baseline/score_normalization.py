"""CSLS-style score normalisation: mirror of the reference's
`vsc/baseline/score_normalization.py` (paths relative to /root/reference).

sim_sn(query, ref) = query.ref - beta * max_n(query.noise) is folded into one extra descriptor
dimension (query' = [query, bias], ref' = [ref, 1]) so the search path is unchanged.  The row
normalisation and the 1-NN search against the noise set run on the GPU (libvscmi); all query
videos are searched in ONE call instead of one faiss call per video (rows are independent, so the
result is the same).
"""
import ctypes
import dataclasses
import logging
from typing import Callable, List, Tuple

import numpy as np

from vsc2022_amd import _lib
from vsc2022_amd.vsc.candidates import CandidateGeneration, MaxScoreAggregation
from vsc2022_amd.vsc.index import VideoFeature

logger = logging.getLogger("score_normalization.py")
logger.setLevel(logging.INFO)


def transform_features(features: List[VideoFeature], transform: Callable) -> List[VideoFeature]:
    """vsc/baseline/score_normalization.py:22-28"""
    return [dataclasses.replace(f, feature=transform(f.feature)) for f in features]


def normalize(x: np.ndarray, device=None) -> np.ndarray:
    """Row L2 normalisation with sklearn.preprocessing.normalize semantics (zero rows stay zero),
    on the GPU.  Output fp32."""
    x = _lib.f32c(x)
    if x.ndim != 2:
        raise ValueError("normalize expects a 2-D array")
    out = np.empty_like(x)
    if x.shape[0]:
        dev = _lib.default_device() if device is None else device
        _lib.check(_lib.lib().vsc_row_normalize(x.ctypes.data, x.shape[0], x.shape[1], _lib.MEM_HOST,
                                                out.ctypes.data, _lib.MEM_HOST, dev))
    return out


def _normalize_videos(features: List[VideoFeature]) -> List[VideoFeature]:
    """One kernel launch for the whole list instead of one per video."""
    if not features:
        return []
    lens = [len(f) for f in features]
    flat = normalize(np.concatenate([_lib.f32c(f.feature) for f in features], axis=0))
    cuts = np.cumsum([0] + lens)
    return [dataclasses.replace(f, feature=flat[a:b]) for f, a, b in zip(features, cuts[:-1], cuts[1:])]


def score_normalize(
    queries: List[VideoFeature],
    refs: List[VideoFeature],
    score_norm_refs: List[VideoFeature],
    l2_normalize: bool = True,
    replace_dim: bool = True,
    beta: float = 1.0,
) -> Tuple[List[VideoFeature], List[VideoFeature]]:
    """vsc/baseline/score_normalization.py:31-105"""
    if {f.video_id for f in refs}.intersection({f.video_id for f in score_norm_refs}):
        raise Exception(
            "Normalizing on the dataset we're evaluating on is against VSC rules. "
            "An independent dataset is needed."
        )
    if score_norm_refs is not None and replace_dim:
        # Make space for the additional score normalization dimension: drop the dimension with the
        # lowest variance over the noise set (host numpy, one pass; score_normalization.py:68-80).
        logger.info("Replacing dimension")
        sn_features = np.concatenate([ref.feature for ref in score_norm_refs], axis=0)
        low_var_dim = sn_features.var(axis=0).argmin()
        queries, refs, score_norm_refs = [
            transform_features(x, lambda feature: np.delete(feature, low_var_dim, axis=1))
            for x in [queries, refs, score_norm_refs]
        ]
    if l2_normalize:
        logger.info("L2 normalizing")
        queries, refs, score_norm_refs = [_normalize_videos(x) for x in [queries, refs, score_norm_refs]]
    logger.info("Applying score normalization")
    index = CandidateGeneration(score_norm_refs, MaxScoreAggregation()).index.index

    # KNN search is ok here (versus a threshold/radius/range search) since we're not searching
    # the dataset we're evaluating on (score_normalization.py:94-96).
    adapted_queries = []
    if queries:
        lens = [len(q) for q in queries]
        flat = np.concatenate([_lib.f32c(q.feature) for q in queries], axis=0)
        similarity, _ = index.search(flat, 1)
        norm_term = -beta * similarity[:, :1]
        cuts = np.cumsum([0] + lens)
        for q, a, b in zip(queries, cuts[:-1], cuts[1:]):
            feature = np.concatenate([q.feature, norm_term[a:b]], axis=1)
            adapted_queries.append(dataclasses.replace(q, feature=feature))
    adapted_refs = []
    for ref in refs:
        ones = np.ones_like(ref.feature[:, :1])
        adapted_refs.append(dataclasses.replace(ref, feature=np.concatenate([ref.feature, ones], axis=1)))
    return adapted_queries, adapted_refs
