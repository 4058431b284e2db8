#!/usr/bin/env python3
"""FastSSCD forward rate against the batch size (fixed batch in HBM): do smaller activations (Infinity Cache) help?"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vsc2022_amd.vsc.baseline.inference import FastSSCD, build_sscd_model, preprocess

dev = torch.device("cuda", 0)
fast = FastSSCD(build_sscd_model(device=dev)).to(dev)
g = torch.Generator(device=dev)
g.manual_seed(5)
for bs in (16, 32, 64, 128, 256):
    x = preprocess(torch.randint(0, 256, (bs, 3, 320, 320), generator=g, device=dev, dtype=torch.uint8))
    with torch.no_grad():
        for _ in range(3):
            fast(x)
        torch.cuda.synchronize()
        it = max(4, 2048 // bs)
        t0 = time.perf_counter()
        for _ in range(it):
            fast(x)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / it
    print(f"batch {bs:4d}: {bs / dt:9.1f} frames/s ({dt * 1e3:.2f} ms per pass)", flush=True)

print("with the forward pass captured in a HIP graph (torch.cuda.CUDAGraph):")
for bs in (16, 32, 64, 128, 256):
    x = preprocess(torch.randint(0, 256, (bs, 3, 320, 320), generator=g, device=dev, dtype=torch.uint8))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.no_grad(), torch.cuda.stream(s):
        for _ in range(3):
            fast(x)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(graph):
        y = fast(x)
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = fast(x)
    graph.replay()
    torch.cuda.synchronize()
    same = torch.equal(ref, y)
    it = max(4, 2048 // bs)
    t0 = time.perf_counter()
    for _ in range(it):
        graph.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / it
    print(f"batch {bs:4d}: {bs / dt:9.1f} frames/s ({dt * 1e3:.2f} ms per pass)  replay == eager: {same}", flush=True)
