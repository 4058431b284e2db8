#!/usr/bin/env python3
"""End-to-end timing of the PUBLIC drop-in surface (numpy in, Python objects / CSV out):
sscd_baseline.match = CandidateGeneration.query + VCSLLocalization.localize_all + CSV writers."""
import argparse
import cProfile
import os
import pstats
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from vsc2022_amd import synth
from vsc2022_amd.vsc.baseline import sscd_baseline
from vsc2022_amd.vsc.index import VideoFeature

ap = argparse.ArgumentParser()
ap.add_argument("--nq", type=int, default=2000)
ap.add_argument("--nr", type=int, default=10000)
ap.add_argument("--profile", action="store_true")
args = ap.parse_args()
t0 = time.perf_counter()
q, r, gts = synth.make_dataset(seed=3, n_query=args.nq, n_ref=args.nr, dim=512, q_frames=(25, 25), r_frames=(50, 50),
                               planted_frac=0.2, static_frac=0.01)
qf, rf = synth.to_video_features(q, VideoFeature), synth.to_video_features(r, VideoFeature)
print(f"synthetic data: {time.perf_counter() - t0:.1f} s ({args.nq * 25} x {args.nr * 50} frames)")
out = tempfile.mkdtemp()
sscd_baseline.match(qf[:50], rf[:200], os.path.join(out, "warm"))
pr = cProfile.Profile() if args.profile else None
t0 = time.perf_counter()
if pr:
    pr.enable()
cand_file, match_file = sscd_baseline.match(qf, rf, os.path.join(out, "run"))
if pr:
    pr.disable()
dt = time.perf_counter() - t0
print(f"sscd_baseline.match: {dt:.2f} s -> {args.nq / dt:.0f} query videos/s "
      f"({sum(1 for _ in open(cand_file)) - 1} candidates, {sum(1 for _ in open(match_file)) - 1} matches)")
if pr:
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
