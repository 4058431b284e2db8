#!/usr/bin/env python3
"""Sensitivity of the engine to the descriptor distribution: BASELINE configs[1]'s shape on every
class of vsc2022_amd/synth.py, without and with score normalisation; one JSON object per line.  Every leg also runs the
exhaustive default-route vs all-fp32-route comparison (all K hits: row, reference, score bits, radius).

    python scripts/bench_distributions.py [--classes clusters,powerlaw,...] [--score-norm] > gpurun_out/dist.jsonl
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    import bench
    from vsc2022_amd import synth

    ap = argparse.ArgumentParser()
    ap.add_argument("--classes", default=",".join(synth.DISTRIBUTIONS))
    ap.add_argument("--score-norm", action="store_true", help="also the score-normalised form of every class")
    ap.add_argument("--query-videos", type=int, default=8000)
    ap.add_argument("--no-exhaustive", action="store_true")
    ap.add_argument("--geo", default="", help='JSON keyword arguments of synth.Geometry, e.g. \'{"n_dominant": 0}\'')
    a = ap.parse_args()
    sys.argv = sys.argv[:1]
    args = bench.parse()
    args.query_videos = a.query_videos
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    for dist in a.classes.split(","):
        for sn in ((False, True) if a.score_norm else (False,)):
            try:
                rep = bench.distribution_leg(args, torch, dev, args.dim, dist, score_norm=sn, exhaustive=not a.no_exhaustive,
                                             geo_kw=json.loads(a.geo) if a.geo else None)
            except Exception as exc:  # noqa: BLE001
                rep = {"data": dist, "score_normalised": sn, "error": f"{type(exc).__name__}: {exc}"}
            print(json.dumps(rep), flush=True)
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
