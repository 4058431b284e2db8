"""End-to-end matching baseline on SSCD descriptors: mirror of the reference's
`vsc/baseline/sscd_baseline.py` (same functions, flags, constants and output files; paths relative
to /root/reference), running search, candidate generation, score normalisation and Temporal-Network
localisation on the MI355X engine.

    python -m vsc2022_amd.vsc.baseline.sscd_baseline --query_features q.npz --ref_features r.npz \
        --output_path out/ [--score_norm_features noise.npz] [--ground_truth gt.csv] [--overwrite]
"""
import argparse
import logging
import os
from typing import List, Tuple

from vsc2022_amd.vsc.baseline.localization import VCSLLocalizationCandidateScore, VCSLLocalizationMaxSim
from vsc2022_amd.vsc.baseline.score_normalization import _normalize_videos, score_normalize
from vsc2022_amd.vsc.candidates import CandidateGeneration, MaxScoreAggregation
from vsc2022_amd.vsc.index import VideoFeature
from vsc2022_amd.vsc.metrics import (AveragePrecision, CandidatePair, Dataset, Match, average_precision,
                                     evaluate_matching_track)
from vsc2022_amd.vsc.storage import load_features, store_features

logger = logging.getLogger("sscd_baseline.py")
logger.setLevel(logging.INFO)

# pipeline constants of the reference (sscd_baseline.py:93-94,111,139,118-135,198)
RETRIEVE_PER_QUERY = 1200.0
CANDIDATES_PER_QUERY = 25.0
LOCALIZE_PER_QUERY = 5.0
BATCH_SIZE = 512
TN_ARGS = dict(model_type="TN", tn_max_step=5, min_length=4, concurrency=16)
SCORE_NORM_BIAS = 0.5
SCORE_NORM_BETA = 1.2


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser()
    p.add_argument("--query_features", help="Path to query descriptors", type=str, required=True)
    p.add_argument("--ref_features", help="Path to reference descriptors", type=str, required=True)
    p.add_argument("--score_norm_features", help="Path to score normalization descriptors", type=str)
    p.add_argument("--output_path", help="The path to write match predictions.", type=str, required=True)
    p.add_argument("--ground_truth", help="Path to the ground truth (labels) CSV file.", type=str)
    p.add_argument("--overwrite", help="Overwrite prediction files, if found.", action="store_true")
    return p


def search(queries: List[VideoFeature], refs: List[VideoFeature], retrieve_per_query: float = RETRIEVE_PER_QUERY,
           candidates_per_query: float = CANDIDATES_PER_QUERY) -> List[CandidatePair]:
    """sscd_baseline.py:90-104"""
    logger.info("Searching")
    cg = CandidateGeneration(refs, MaxScoreAggregation())
    candidates = cg.query(queries, global_k=int(retrieve_per_query * len(queries)))
    candidates = candidates[: int(candidates_per_query * len(queries))]
    logger.info("Got %d candidates", len(candidates))
    return candidates


def localize_and_verify(queries: List[VideoFeature], refs: List[VideoFeature], candidates: List[CandidatePair],
                        localize_per_query: float = LOCALIZE_PER_QUERY, score_normalization: bool = False
                        ) -> List[Match]:
    """sscd_baseline.py:107-152"""
    candidates = candidates[: int(len(queries) * localize_per_query)]
    if score_normalization:
        alignment = VCSLLocalizationMaxSim(queries, refs, similarity_bias=SCORE_NORM_BIAS, **TN_ARGS)
    else:
        alignment = VCSLLocalizationCandidateScore(_normalize_videos(queries), _normalize_videos(refs), **TN_ARGS)
    matches: List[Match] = []
    logger.info("Aligning %s candidate pairs", len(candidates))
    done = 0
    while done < len(candidates):
        batch = candidates[done : done + BATCH_SIZE]
        matches.extend(alignment.localize_all(batch))
        done += len(batch)
        logger.info("Aligned %d pairs of %d; %d predictions so far", done, len(candidates), len(matches))
    return matches


def match(queries: List[VideoFeature], refs: List[VideoFeature], output_path: str,
          score_normalization: bool = False) -> Tuple[str, str]:
    """sscd_baseline.py:155-176: writes candidates.csv and matches.csv."""
    candidates = search(queries, refs)
    os.makedirs(output_path, exist_ok=True)
    candidate_file = os.path.join(output_path, "candidates.csv")
    CandidatePair.write_csv(candidates, candidate_file)
    matches = localize_and_verify(queries, refs, candidates, score_normalization=score_normalization)
    matches_file = os.path.join(output_path, "matches.csv")
    Match.write_csv(matches, matches_file)
    return candidate_file, matches_file


def create_pr_plot(ap: AveragePrecision, filename: str):
    import matplotlib.pyplot as plt

    ap.pr_curve.plot(linewidth=1)
    plt.savefig(filename)
    plt.show()


def main(args):
    """sscd_baseline.py:185-231"""
    if os.path.exists(args.output_path) and not args.overwrite:
        raise Exception(f"Output path already exists: {args.output_path}. Do you want to --overwrite?")
    queries = load_features(args.query_features, Dataset.QUERIES)
    refs = load_features(args.ref_features, Dataset.REFS)
    score_normalization = False
    if args.score_norm_features:
        queries, refs = score_normalize(queries, refs, load_features(args.score_norm_features, Dataset.REFS),
                                        beta=SCORE_NORM_BETA)
        score_normalization = True
        os.makedirs(args.output_path, exist_ok=True)
        store_features(os.path.join(args.output_path, "sn_queries.npz"), queries)
        store_features(os.path.join(args.output_path, "sn_refs.npz"), refs)
    candidate_file, match_file = match(queries, refs, args.output_path, score_normalization=score_normalization)
    if not args.ground_truth:
        return
    gt_pairs = CandidatePair.from_matches(Match.read_csv(args.ground_truth, is_gt=True))
    candidate_uap = average_precision(gt_pairs, CandidatePair.read_csv(candidate_file))
    logger.info(f"Candidate uAP: {candidate_uap.ap:.4f}")
    candidate_pr_file = os.path.join(args.output_path, "candidate_precision_recall.pdf")
    create_pr_plot(candidate_uap, candidate_pr_file)
    match_metrics = evaluate_matching_track(args.ground_truth, match_file)
    logger.info(f"Matching track metric: {match_metrics.segment_ap.ap:.4f}")
    matching_pr_file = os.path.join(args.output_path, "precision_recall.pdf")
    create_pr_plot(match_metrics.segment_ap, matching_pr_file)
    logger.info(f"Candidates: {candidate_file}")
    logger.info(f"Matches: {match_file}")
    logger.info(f"Candidate PR plot: {candidate_pr_file}")
    logger.info(f"Match PR plot: {matching_pr_file}")


if __name__ == "__main__":
    logging.basicConfig(format="%(asctime)s %(levelname)-8s %(message)s", level=logging.INFO,
                        datefmt="%Y-%m-%d %H:%M:%S")
    main(build_parser().parse_args())
