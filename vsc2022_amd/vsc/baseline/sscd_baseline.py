"""SSCD matching baseline on the MI355X engine: search -> candidates -> (score normalisation) ->
Temporal-Network localisation -> CSV files -> metrics.

Drop-in for `python -m vsc.baseline.sscd_baseline` of the reference (same flags, same output files
`candidates.csv` / `matches.csv` / `sn_*.npz`, same module-level functions `search`,
`localize_and_verify`, `match`, `main`; /root/reference/vsc/baseline/sscd_baseline.py:54-231).

    python -m vsc2022_amd.vsc.baseline.sscd_baseline --query_features q.npz --ref_features r.npz \
        --output_path out/ [--score_norm_features noise.npz] [--ground_truth gt.csv] [--overwrite]

Several GPUs: start the same module as N ranks (`python -m torch.distributed.run --nproc-per-node N -m
vsc2022_amd.vsc.baseline.sscd_baseline ...`): the query videos are sharded over the ranks, the files rank 0 writes are the
single-process run's, byte for byte (vsc/baseline/sharded.py; the reference's analogue is FAISS spreading its index over
all visible GPUs, /root/reference/vsc/index.py:153).
"""
import argparse
import logging
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

from vsc2022_amd.vsc import metrics as M
from vsc2022_amd.vsc import storage
from vsc2022_amd.vsc.baseline import localization as loc
from vsc2022_amd.vsc.baseline import sharded
from vsc2022_amd.vsc.baseline.score_normalization import _normalize_videos, score_normalize
from vsc2022_amd.vsc.candidates import CandidateGeneration, MaxScoreAggregation
from vsc2022_amd.vsc.index import VideoFeature

logger = logging.getLogger("sscd_baseline.py")
logger.setLevel(logging.INFO)


@dataclass(frozen=True)
class PipelineConstants:
    """The reference's hard-coded tuning (sscd_baseline.py:93-94,111,121-125,139,198)."""

    retrieve_per_query: float = 1200.0   # frame hits searched for, per query video
    candidates_per_query: float = 25.0   # (query, ref) pairs kept
    localize_per_query: float = 5.0      # pairs handed to the aligner
    batch_size: int = 512                # pairs per localize_all call
    tn_max_step: int = 5
    tn_min_length: int = 4
    tn_concurrency: int = 16             # accepted for compatibility; the GPU aligns a whole batch at once
    score_norm_bias: float = 0.5
    score_norm_beta: float = 1.2


CONSTANTS = PipelineConstants()


def search(queries: List[VideoFeature], refs: List[VideoFeature],
           retrieve_per_query: float = CONSTANTS.retrieve_per_query,
           candidates_per_query: float = CONSTANTS.candidates_per_query) -> List[M.CandidatePair]:
    """Best `candidates_per_query * len(queries)` video pairs by max frame similarity."""
    n_hits = int(retrieve_per_query * len(queries))
    n_keep = int(candidates_per_query * len(queries))
    pairs = CandidateGeneration(refs, MaxScoreAggregation()).query(queries, global_k=n_hits)[:n_keep]
    logger.info("search: %d candidate pairs from the %d best frame hits", len(pairs), n_hits)
    return pairs


def _aligner(queries, refs, score_normalization: bool) -> loc.VCSLLocalization:
    tn = dict(model_type="TN", tn_max_step=CONSTANTS.tn_max_step, min_length=CONSTANTS.tn_min_length,
              concurrency=CONSTANTS.tn_concurrency)
    if score_normalization:
        # score-normalised similarities live around zero: shift them for the aligner, score boxes by
        # their best frame similarity
        return loc.VCSLLocalizationMaxSim(queries, refs, similarity_bias=CONSTANTS.score_norm_bias, **tn)
    # plain descriptors: align on cosine similarity, score boxes with the candidate's retrieval score
    return loc.VCSLLocalizationCandidateScore(_normalize_videos(queries), _normalize_videos(refs), **tn)


def localize_and_verify(queries: List[VideoFeature], refs: List[VideoFeature], candidates: Sequence[M.CandidatePair],
                        localize_per_query: float = CONSTANTS.localize_per_query,
                        score_normalization: bool = False) -> List[M.Match]:
    """Temporal-Network boxes for the best `localize_per_query * len(queries)` candidates."""
    todo = candidates[: int(len(queries) * localize_per_query)]
    aligner = _aligner(queries, refs, score_normalization)
    found: List[M.Match] = []
    for start in range(0, len(todo), CONSTANTS.batch_size):
        found += aligner.localize_all(todo[start : start + CONSTANTS.batch_size])
        logger.info("localised %d / %d pairs, %d segments so far", min(start + CONSTANTS.batch_size, len(todo)),
                    len(todo), len(found))
    return found


def match(queries: List[VideoFeature], refs: List[VideoFeature], output_path: str,
          score_normalization: bool = False) -> Tuple[str, str]:
    """Runs the matching pipeline; returns the paths of candidates.csv and matches.csv."""
    os.makedirs(output_path, exist_ok=True)
    candidate_file = os.path.join(output_path, "candidates.csv")
    matches_file = os.path.join(output_path, "matches.csv")
    if sharded.requested():
        # N ranks: each searches and localises its range of the query videos; rank 0 writes the (identical) files
        n_keep = int(CONSTANTS.candidates_per_query * len(queries))
        pairs, found, res = sharded.match_sharded(
            queries, refs, score_normalization, CONSTANTS.retrieve_per_query, CONSTANTS.candidates_per_query,
            CONSTANTS.localize_per_query, CONSTANTS.tn_max_step, CONSTANTS.tn_min_length, CONSTANTS.score_norm_bias)
        if sharded.is_main():
            logger.info("sharded search: %d candidate pairs, %d segments (tie on the K cut: %s%s)", len(pairs), len(found),
                        res.tie_on_cut, ", tied hits dropped as the reference drops them" if res.ties_dropped else "")
            M.CandidatePair.write_csv(pairs[:n_keep], candidate_file)
            M.Match.write_csv(found, matches_file)
        sharded.barrier()
        return candidate_file, matches_file
    pairs = search(queries, refs)
    M.CandidatePair.write_csv(pairs, candidate_file)
    M.Match.write_csv(localize_and_verify(queries, refs, pairs, score_normalization=score_normalization), matches_file)
    return candidate_file, matches_file


def create_pr_plot(ap: M.AveragePrecision, filename: str):
    import matplotlib.pyplot as plt

    ap.pr_curve.plot(linewidth=1)
    plt.savefig(filename)
    plt.show()


def _report(output_path: str, ground_truth: str, candidate_file: str, match_file: str):
    truth = M.CandidatePair.from_matches(M.Match.read_csv(ground_truth, is_gt=True))
    uap = M.average_precision(truth, M.CandidatePair.read_csv(candidate_file))
    tracks = M.evaluate_matching_track(ground_truth, match_file)
    logger.info("Candidate uAP: %.4f", uap.ap)
    logger.info("Matching track metric: %.4f", tracks.segment_ap.ap)
    for curve, name in ((uap, "candidate_precision_recall.pdf"), (tracks.segment_ap, "precision_recall.pdf")):
        target = os.path.join(output_path, name)
        create_pr_plot(curve, target)
        logger.info("PR plot: %s", target)
    logger.info("Candidates: %s", candidate_file)
    logger.info("Matches: %s", match_file)


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="SSCD matching baseline on MI355X")
    for flag, text, required in (("--query_features", "Path to query descriptors", True),
                                 ("--ref_features", "Path to reference descriptors", True),
                                 ("--score_norm_features", "Path to score normalization descriptors", False),
                                 ("--output_path", "The path to write match predictions.", True),
                                 ("--ground_truth", "Path to the ground truth (labels) CSV file.", False)):
        p.add_argument(flag, help=text, type=str, required=required)
    p.add_argument("--overwrite", help="Overwrite prediction files, if found.", action="store_true")
    return p


def main(args):
    multi = sharded.requested()
    if multi:
        sharded.init()
    if os.path.exists(args.output_path) and not args.overwrite:
        raise Exception(f"Output path already exists: {args.output_path}. Do you want to --overwrite?")
    if multi:
        sharded.barrier()  # (nobody creates the directory before everybody has looked)
    queries = storage.load_features(args.query_features, M.Dataset.QUERIES)
    refs = storage.load_features(args.ref_features, M.Dataset.REFS)
    normalised = bool(args.score_norm_features)
    if normalised:
        noise = storage.load_features(args.score_norm_features, M.Dataset.REFS)
        if multi:
            # the 1-NN against the noise set is per query row: every rank adapts its own range of the query videos
            # (and the references, which are cheap), then the adapted query descriptors are gathered
            lo, hi = sharded.shard_of(queries)
            mine, refs = score_normalize(queries[lo:hi], refs, noise, beta=CONSTANTS.score_norm_beta)
            queries = sharded.gather_videos(mine, queries)
        else:
            queries, refs = score_normalize(queries, refs, noise, beta=CONSTANTS.score_norm_beta)
        if sharded.is_main():
            os.makedirs(args.output_path, exist_ok=True)
            storage.store_features(os.path.join(args.output_path, "sn_queries.npz"), queries)
            storage.store_features(os.path.join(args.output_path, "sn_refs.npz"), refs)
    candidate_file, match_file = match(queries, refs, args.output_path, score_normalization=normalised)
    if args.ground_truth and sharded.is_main():
        _report(args.output_path, args.ground_truth, candidate_file, match_file)
    if multi:
        sharded.barrier()


if __name__ == "__main__":
    logging.basicConfig(format="%(asctime)s %(levelname)-8s %(message)s", level=logging.INFO,
                        datefmt="%Y-%m-%d %H:%M:%S")
    main(build_parser().parse_args())
