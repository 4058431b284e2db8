#!/usr/bin/env python3
"""BASELINE config 5 on ONE GPU, for sizing: a 16M x 512 fp32 reference index (33 GB packed) searched
with k-NN (k = 20) and with the global-threshold search; reports times and the similarity TFLOP/s.
(The 8-shard variant of config 5 holds 2M rows per GPU and merges with dist.ref_sharded_knn.)"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from vsc2022_amd.vsc.index import FlatIndex

ap = argparse.ArgumentParser()
ap.add_argument("--nr", type=int, default=16_000_000)
ap.add_argument("--nq", type=int, default=8192)
ap.add_argument("--dim", type=int, default=512)
args = ap.parse_args()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(4)
idx = FlatIndex(args.dim)
chunk = 2_000_000
t0 = time.perf_counter()
for r0 in range(0, args.nr, chunk):
    x = torch.randn((min(chunk, args.nr - r0), args.dim), generator=g, device=dev)
    x /= x.norm(dim=1, keepdim=True)
    torch.cuda.synchronize()
    idx.add(x)
del x
print(f"index: {idx.ntotal} rows added in {time.perf_counter() - t0:.1f} s (incremental, 2M-row chunks)")
q = torch.randn((args.nq, args.dim), generator=g, device=dev)
q /= q.norm(dim=1, keepdim=True)
torch.cuda.synchronize()
idx.profile(True)
for k in (1, 20):
    idx.profile_read(True)
    t0 = time.perf_counter()
    D, I = idx.search(q, k)
    dt = time.perf_counter() - t0
    p = idx.profile_read(True)
    print(f"knn k={k}: {dt * 1e3:.0f} ms wall, sim kernel {p['sim_ms']:.0f} ms, "
          f"{p['sim_flops'] / 1e12 / (p['sim_ms'] / 1e3):.1f} TFLOP/s; row 0 best = {I[0, 0]} ({D[0, 0]:.4f})")
K = 1200 * (args.nq // 25)
t0 = time.perf_counter()
i, j, s, rad = idx.global_topk(q, K, device_out=True)
dt = time.perf_counter() - t0
p = idx.profile_read(True)
print(f"global top-K (K={K}): {dt * 1e3:.0f} ms wall, sim kernel {p['sim_ms']:.0f} ms, "
      f"{p['sim_flops'] / 1e12 / (p['sim_ms'] / 1e3):.1f} TFLOP/s; hits={s.numel()} radius={rad:.4f}")
print(f"HBM in use: {torch.cuda.mem_get_info()[1] / 2**30 - torch.cuda.mem_get_info()[0] / 2**30:.1f} GiB")
