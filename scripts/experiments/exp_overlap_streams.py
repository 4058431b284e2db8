#!/usr/bin/env python3
"""Round 5 experiment: how much of the exact stage (fabric-bound) hides behind the int8 pre-filter (power-bound) when two
searches run on two streams at once?  Two handles on their own torch streams, two host threads (ctypes releases the GIL):
A = the 1-NN of 1 M query rows against a 2 M-row noise index (int8 kernel, little exact stage), B = the global-threshold search
of the same rows against 2 M references (int8 + fp16 kernels + 0.36 s of exact stage).  Sequential vs concurrent wall time.
Not part of the product: the schedule's batches depend on each other, only independent calls can overlap like this."""
import sys
import os
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from vsc2022_amd.vsc.index import FlatIndex

dev = torch.device("cuda", 0)
NQ, NR, D, K = 1_000_000, 2_000_000, 512, 48_000_000
g = torch.Generator(device=dev).manual_seed(0)


def unit(n):
    x = torch.randn((n, D), generator=g, device=dev)
    return x / x.norm(dim=1, keepdim=True)


q, refs, noise = unit(NQ), unit(NR), unit(NR)
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
with torch.cuda.stream(sa):
    ia = FlatIndex(D)
    ia.use_torch_stream()
    ia.add(noise)
with torch.cuda.stream(sb):
    ib = FlatIndex(D)
    ib.use_torch_stream()
    ib.add(refs)
torch.cuda.synchronize()
del noise, refs
times = {}


def run_a(tag):
    with torch.cuda.stream(sa):
        t0 = time.perf_counter()
        ia.search(q, 1, device_out=True)
        sa.synchronize()
        times[tag] = time.perf_counter() - t0


def run_b(tag):
    with torch.cuda.stream(sb):
        t0 = time.perf_counter()
        ib.global_topk(q, K, device_out=True)
        sb.synchronize()
        times[tag] = time.perf_counter() - t0


run_a("warm_a"), run_b("warm_b")
for rep in range(3):
    t0 = time.perf_counter()
    run_a("a"), run_b("b")
    seq = time.perf_counter() - t0
    ta, tb = threading.Thread(target=run_a, args=("ca",)), threading.Thread(target=run_b, args=("cb",))
    t0 = time.perf_counter()
    ta.start(), tb.start()
    ta.join(), tb.join()
    con = time.perf_counter() - t0
    print(f"rep {rep}: sequential {seq*1e3:.0f} ms (1-NN {times['a']*1e3:.0f} + search {times['b']*1e3:.0f}); concurrent {con*1e3:.0f} ms "
          f"(1-NN {times['ca']*1e3:.0f}, search {times['cb']*1e3:.0f}): {100*(1-con/seq):+.1f} %", flush=True)
