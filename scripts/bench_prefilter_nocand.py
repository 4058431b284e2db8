#!/usr/bin/env python3
"""The pre-filter kernels on a batch that yields no / few candidates (range search at a high radius): separates the cost
of the K loop + epilogue test from the cost of emitting candidates.  VSC_I8=2 -> int8 kernel, VSC_I8=0 -> fp16 kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vsc2022_amd.vsc.index import FlatIndex
nq, nr, d = 32768, 2000000, 512
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(0)
r = torch.randn((nr, d), generator=g, device=dev); r /= r.norm(dim=1, keepdim=True)
q = torch.randn((nq, d), generator=g, device=dev); q /= q.norm(dim=1, keepdim=True)
os.environ["VSC_PREFILTER"] = "2"
idx = FlatIndex(d); idx.add(r); idx.profile(True)
for radius in (0.9, 0.22, 0.18, 0.16):
    idx.range_search(q, radius); idx.profile_read(True)
    t0 = time.perf_counter(); lims, D, I = idx.range_search(q, radius); dt = time.perf_counter() - t0
    p = idx.profile_read(True)
    ms = p["i8_ms"] + p["f16_ms"]; fl = p["i8_flops"] + p["f16_flops"]
    print(f"radius {radius}: hits {len(D)} candidates {p['candidates']} pre-filter {ms:.2f} ms = {fl/ms/1e9:.0f} T(FL)OP/s "
          f"(int8 {p['i8_launches']} launches, fp16 {p['f16_launches']}) rescore {p['rescore_ms']:.2f} ms", flush=True)
