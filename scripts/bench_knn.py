#!/usr/bin/env python3
"""Micro-benchmark of the k-NN similarity kernel (sim_knn_kernel + merge)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from vsc2022_amd.vsc.index import FlatIndex

ap = argparse.ArgumentParser()
ap.add_argument("--nq", type=int, default=65536)
ap.add_argument("--nr", type=int, default=1000000)
ap.add_argument("--dim", type=int, default=512)
ap.add_argument("--k", type=int, nargs="+", default=[1, 20])
args = ap.parse_args()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(0)
r = torch.randn((args.nr, args.dim), generator=g, device=dev)
r /= r.norm(dim=1, keepdim=True)
q = torch.randn((args.nq, args.dim), generator=g, device=dev)
q /= q.norm(dim=1, keepdim=True)
idx = FlatIndex(args.dim)
torch.cuda.synchronize()
idx.add(r)
for k in args.k:
    idx.search(q[:1024], k)
    idx.profile(True)
    idx.profile_read(True)
    t0 = time.perf_counter()
    D, I = idx.search(q, k)
    dt = time.perf_counter() - t0
    p = idx.profile_read(True)
    eff = 2.0 * args.nq * args.nr * args.dim / dt / 1e12
    print(f"k={k} nq={args.nq} nr={args.nr} wall={dt*1e3:.1f} ms (= {eff:.0f} effective TFLOP/s) | exact fp32 kernel "
          f"{p['sim_ms']:.1f} ms {p['sim_flops']/1e12/max(p['sim_ms'],1e-9)*1e3:.1f} TFLOP/s | f16 {p['f16_ms']:.1f} ms "
          f"{p['f16_flops']/1e12/max(p['f16_ms'],1e-9)*1e3:.1f} TFLOP/s | i8 {p['i8_ms']:.1f} ms | rescore {p['rescore_ms']:.1f} ms "
          f"candidates={p['candidates']}")
