"""Descriptor-track evaluation from the command line, on the MI355X engine.

Accepts the flags of the reference's `descriptor_eval.py` (:16-35):

    python -m vsc2022_amd.cli.descriptor_eval --query_features q.npz --ref_features r.npz \
        [--ground_truth gt.csv] [--candidates_output candidates.csv]
"""
import argparse
import logging

from vsc2022_amd.vsc import descriptor_eval_lib, metrics

FLAGS = (
    # name, required, help
    ("--query_features", True, "query descriptors (.npz)"),
    ("--ref_features", True, "reference descriptors (.npz)"),
    ("--candidates_output", False, "where to write the candidate pairs (CSV, optional)"),
    ("--ground_truth", False, "ground-truth CSV; enables the micro-AP report"),
)


def main(argv=None):
    parser = argparse.ArgumentParser(description="descriptor track: search + candidate micro-AP")
    for name, required, text in FLAGS:
        parser.add_argument(name, required=required, type=str, help=text)
    args = parser.parse_args(argv)
    logging.basicConfig(level=logging.INFO, datefmt="%Y-%m-%d %H:%M:%S",
                        format="%(asctime)s %(levelname)-8s %(message)s")
    ap, candidates = descriptor_eval_lib.evaluate_descriptor_track(args.query_features, args.ref_features,
                                                                   args.ground_truth)
    from vsc2022_amd.vsc.baseline import sharded

    if args.candidates_output and sharded.is_main():
        metrics.CandidatePair.write_csv(candidates, args.candidates_output)
        logging.getLogger("descriptor_eval").info("wrote %d candidates to %s", len(candidates), args.candidates_output)
    return ap, candidates


if __name__ == "__main__":
    main()
