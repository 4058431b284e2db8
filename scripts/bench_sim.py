#!/usr/bin/env python3
"""Micro-benchmark of the similarity kernel alone: one global-threshold search on a big batch.
Prints TFLOP/s of the fp32 MFMA kernel measured with HIP events inside libvscmi."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from vsc2022_amd.engine import DeviceMatcher

ap = argparse.ArgumentParser()
ap.add_argument("--nq", type=int, default=65536)
ap.add_argument("--nr", type=int, default=1000000)
ap.add_argument("--dim", type=int, default=512)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--K", type=int, default=2000000)
args = ap.parse_args()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(0)
r = torch.randn((args.nr, args.dim), generator=g, device=dev)
r /= r.norm(dim=1, keepdim=True)
q = torch.randn((args.nq, args.dim), generator=g, device=dev)
q /= q.norm(dim=1, keepdim=True)
m = DeviceMatcher(r, np.arange(0, args.nr + 1, 50, dtype=np.int64), 0)
m.set_queries(q, np.arange(0, args.nq + 1, 32, dtype=np.int64))
m.search(args.K)
m.index.profile(True)
m.index.profile_read(True)
t0 = time.perf_counter()
for _ in range(args.reps):
    hi, hj, hs, rad = m.search(args.K)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
p = m.index.profile_read(True)
print(f"nq={args.nq} nr={args.nr} d={args.dim} hits={hs.numel()} wall/search={dt/args.reps*1e3:.1f} ms "
      f"sim_kernel={p['sim_ms']/args.reps:.1f} ms launches={p['sim_launches']//args.reps} "
      f"TFLOP/s={p['sim_flops']/1e12/max(p['sim_ms'],1e-9)*1e3:.1f} | f16 {p['f16_ms']/args.reps:.1f} ms "
      f"launches={p['f16_launches']//args.reps} TFLOP/s={p['f16_flops']/1e12/max(p['f16_ms'],1e-9)*1e3:.1f} | "
      f"rescore {p['rescore_ms']/args.reps:.1f} ms candidates={p['candidates']}")
