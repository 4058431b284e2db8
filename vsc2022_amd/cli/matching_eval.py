"""Matching-track evaluation (segment AP) -- flags of the reference's `matching_eval.py` (:16-48):
    python -m vsc2022_amd.cli.matching_eval --predictions matches.csv --ground_truth gt.csv
"""
import argparse

from vsc2022_amd.vsc.metrics import evaluate_matching_track


def main(argv=None):
    p = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    p.add_argument("--predictions", required=True, type=str, help="Path containing match predictions")
    p.add_argument("--ground_truth", required=True, type=str, help="Path containing ground truth labels")
    args = p.parse_args(argv)
    metrics = evaluate_matching_track(args.ground_truth, args.predictions)
    print(f"Matching track segment AP: {metrics.segment_ap.ap:.4f}")
    return metrics


if __name__ == "__main__":
    main()
