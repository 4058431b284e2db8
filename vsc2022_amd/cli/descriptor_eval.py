"""Descriptor-track evaluation on the MI355X engine.

Same flags and outputs as the reference's `descriptor_eval.py` (:16-58):
    python -m vsc2022_amd.cli.descriptor_eval --query_features q.npz --ref_features r.npz \
        [--ground_truth gt.csv] [--candidates_output candidates.csv]
"""
import argparse
import logging

from vsc2022_amd.vsc.descriptor_eval_lib import evaluate_descriptor_track
from vsc2022_amd.vsc.metrics import CandidatePair

log = logging.getLogger("descriptor_eval_lib.py")


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    p.add_argument("--query_features", required=True, type=str, help="Path containing query features")
    p.add_argument("--ref_features", required=True, type=str, help="Path containing reference features")
    p.add_argument("--candidates_output", type=str, help="Path to write candidates (optional)")
    p.add_argument("--ground_truth", type=str, help="Path containing Groundtruth")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    logging.basicConfig(format="%(asctime)s %(levelname)-8s %(message)s", level=logging.INFO,
                        datefmt="%Y-%m-%d %H:%M:%S")
    log.setLevel(logging.INFO)
    ap, candidates = evaluate_descriptor_track(args.query_features, args.ref_features, args.ground_truth)
    if args.candidates_output:
        log.info(f"Storing candidates to {args.candidates_output}")
        CandidatePair.write_csv(candidates, args.candidates_output)
    return ap, candidates


if __name__ == "__main__":
    main()
