"""Descriptor-track evaluation on the MI355X engine.

Drop-in for the reference module of the same name (`evaluate_descriptor_track`, the two
per-query constants; /root/reference/vsc/descriptor_eval_lib.py:23-24,27-60): descriptors are read
from two `.npz` files, every query frame is searched against every reference frame on the GPU, the
best `25 x #queries` (query video, reference video) pairs are kept and -- when a ground-truth CSV is
given -- scored with the challenge's micro-AP.
"""
import logging
import time
from typing import List, Optional, Tuple

from vsc2022_amd.vsc import candidates as _cand
from vsc2022_amd.vsc import metrics as _metrics
from vsc2022_amd.vsc import storage as _storage

# Frame hits retrieved / candidate pairs kept, per query video (part of the parity contract).
RETRIEVAL_CANDIDATES_PER_QUERY = 1200
AGGREGATED_CANDIDATES_PER_QUERY = 25

logger = logging.getLogger("descriptor_eval_lib.py")
logger.setLevel(logging.INFO)


def _budgets(n_query_videos: int) -> Tuple[int, int]:
    return (int(RETRIEVAL_CANDIDATES_PER_QUERY * n_query_videos),
            int(AGGREGATED_CANDIDATES_PER_QUERY * n_query_videos))


def _score(candidates, ground_truth_filename: str) -> _metrics.AveragePrecision:
    truth = _metrics.CandidatePair.from_matches(_metrics.Match.read_csv(ground_truth_filename, is_gt=True))
    result = _metrics.average_precision(truth, candidates)
    logger.info("micro-AP over %d ground-truth pairs: %.4f", len(truth), result.ap)
    return result


def evaluate_descriptor_track(
    query_feature_filename: str,
    ref_feature_filename: str,
    ground_truth_filename: Optional[str],
) -> Tuple[Optional[_metrics.AveragePrecision], List[_metrics.CandidatePair]]:
    """Returns (micro-AP or None, candidate pairs best first)."""
    queries = _storage.load_features(query_feature_filename, _metrics.Dataset.QUERIES)
    refs = _storage.load_features(ref_feature_filename, _metrics.Dataset.REFS)
    logger.info("descriptors: %d query videos, %d reference videos", len(queries), len(refs))
    n_hits, n_keep = _budgets(len(queries))

    started = time.perf_counter()
    from vsc2022_amd.vsc.baseline import sharded

    if sharded.requested():
        # started as N ranks (python -m torch.distributed.run ... -m vsc2022_amd.cli.descriptor_eval): the query videos
        # are sharded, every rank ends up with the single-process candidate table (vsc/baseline/sharded.py)
        sharded.init()
        pairs, _, _ = sharded.match_sharded(queries, refs, False, RETRIEVAL_CANDIDATES_PER_QUERY,
                                            AGGREGATED_CANDIDATES_PER_QUERY, 5, 5, 4, 0.0, localize=False)
    else:
        generator = _cand.CandidateGeneration(refs, _cand.MaxScoreAggregation())
        pairs = generator.query(queries, global_k=n_hits)
    logger.info("search for the %d best frame hits -> %d distinct video pairs in %.2f s", n_hits, len(pairs),
                time.perf_counter() - started)
    kept = pairs[:n_keep] if len(pairs) > n_keep else pairs
    if ground_truth_filename is None:
        return None, kept
    return _score(kept, ground_truth_filename), kept
