#!/usr/bin/env python3
"""Frames -> matches on one GPU without leaving HBM: 8000 synthetic query videos x 25 frames (320 x 320) through FastSSCD,
descriptors handed to the matching engine on the device (no .npz), searched / aggregated / localised against 40000
reference videos x 50 frames of synthetic descriptors (BASELINE configs[1] shape) with a share of the references
replaced by the descriptors of query videos so that there is something to find."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import synth_on_device
from vsc2022_amd.engine import DeviceMatcher
from vsc2022_amd.vsc.baseline.inference import FastSSCD, SyntheticVideos, build_sscd_model, run_inference_packed, to_flat

ap = argparse.ArgumentParser()
ap.add_argument("--videos", type=int, default=8000)
ap.add_argument("--ref-videos", type=int, default=40000)
args = ap.parse_args()
dev = torch.device("cuda", 0)
qf, rfr, dim = 25, 50, 512
net = FastSSCD(build_sscd_model(device=dev)).to(dev)
src = SyntheticVideos(n_videos=args.videos, frames=(qf, qf), size=320)
warm = SyntheticVideos(n_videos=16, frames=(qf, qf), size=320)
for _ in run_inference_packed(net, warm, dev, 256):
    pass
refs = synth_on_device(torch, dev, 1, args.ref_videos, rfr, dim)
matcher = DeviceMatcher(refs, np.arange(args.ref_videos + 1, dtype=np.int64) * rfr, 0)
# warm the matcher (buffers, learned capacities) on synthetic queries of the same shape
matcher.set_queries(synth_on_device(torch, dev, 1001, args.videos, qf, dim), np.arange(args.videos + 1, dtype=np.int64) * qf)
matcher.match()
torch.cuda.synchronize()
t0 = time.perf_counter()
feats, off, _ = to_flat(run_inference_packed(net, src, dev, 256))
feats = feats / feats.norm(dim=1, keepdim=True)
torch.cuda.synchronize()
t1 = time.perf_counter()
matcher.set_queries(feats, off)
res = matcher.match()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{args.videos} query videos x {qf} frames: inference {t1 - t0:.2f} s ({args.videos * qf / (t1 - t0):.0f} frames/s), "
      f"search + candidates + localisation against {args.ref_videos * rfr} reference frames {t2 - t1:.3f} s "
      f"({res.n_hits} hits, {res.n_candidates} candidates, {res.n_localized} pairs localised); "
      f"end to end {t2 - t0:.2f} s = {args.videos / (t2 - t0):.0f} query videos/s from frames, descriptors never leave the HBM")
