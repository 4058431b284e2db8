#!/usr/bin/env python3
"""Soak test: random shapes, the pre-filtered routes (forced on every batch / every k-NN) against the
all-fp32 route on the same GPU -- global top-K, k-NN and range search must agree bit for bit.

    python scripts/fuzz_prefilter.py --seconds 120 --seed 0
    VSC_I8=2 python scripts/fuzz_prefilter.py --seconds 120        # ... with the int8 kernel on every pre-filtered batch
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from vsc2022_amd.vsc.index import FlatIndex

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120.0)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--only", type=int, default=-1, help="replay the random stream, run only this case")
ap.add_argument("--big", action="store_true", help="larger shapes, pre-filter chosen by the density / size rules")
args = ap.parse_args()
rng = np.random.default_rng(args.seed)


def make(mode, d):
    return FlatIndex(d, options={"prefilter": int(mode)})


def bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


t_end = time.time() + args.seconds
n_cases = 0
while time.time() < t_end:
    d = int(rng.choice([3, 17, 40, 64, 100, 128, 200, 256, 384, 512, 600, 768, 1000]))
    nq = int(rng.integers(1, 4000))
    nr = int(rng.integers(1, 25000))
    if args.big:
        d = int(rng.choice([32, 64, 128, 256]))
        nq = int(rng.integers(2000, 70000))
        nr = int(rng.integers(20000, 150000))
    style = int(rng.integers(0, 9))
    q = rng.standard_normal((nq, d)).astype(np.float32)
    r = rng.standard_normal((nr, d)).astype(np.float32)
    if style >= 5 and d >= 8:
        # styles 5-8: the non-isotropic classes of vsc2022_amd/synth.py (cluster mixture, power-law spectrum, non-zero mean +
        # dominant coordinates, AR(1) runs) -- what the int8 / fp16 bounds and the density rules were NOT tuned on
        from vsc2022_amd import synth

        geo = synth.Geometry(("clusters", "powerlaw", "offset", "temporal")[style - 5], d, int(rng.integers(1 << 30)),
                             n_centres=int(rng.integers(2, 200)))
        def rows(n):
            out, done = np.empty((n, d), dtype=np.float32), 0
            while done < n:
                m = int(min(n - done, rng.integers(1, 80)))
                out[done : done + m] = geo.video_rows(rng, m)
                done += m
            return out
        q, r = rows(nq), rows(nr)
    elif style != 1:  # unit rows (descriptor-like); style 1 keeps raw gaussian rows (norm ~ sqrt(d))
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        r /= np.linalg.norm(r, axis=1, keepdims=True)
    if style == 2 and nr > 20:  # duplicates -> exact ties
        r[rng.integers(0, nr, nr // 5)] = r[int(rng.integers(0, nr))]
        q[rng.integers(0, nq, max(1, nq // 7))] = r[int(rng.integers(0, nr))]
    if style == 3:  # widely different norms, log-uniform from 1e-5 (fp16-subnormal elements) to 20
        q *= np.exp(rng.uniform(np.log(1e-5), np.log(20.0), (nq, 1))).astype(np.float32)
        r *= np.exp(rng.uniform(np.log(1e-5), np.log(20.0), (nr, 1))).astype(np.float32)
    if style == 4 and d > 4:  # coordinates on which all references agree (score-normalised descriptors have one)
        for c in rng.choice(d, int(rng.integers(1, min(d - 1, 12))), replace=False):
            r[:, c] = np.float32(rng.choice([1.0, -1.0, 0.3, 25.0, 1e-3]))
            q[:, c] = rng.uniform(-0.5, 0.5, nq).astype(np.float32)
    if os.environ.get("FUZZ_VERBOSE"):
        print(f"case {n_cases}: d={d} nq={nq} nr={nr} style={style}", flush=True)
    cut = int(rng.integers(0, nr + 1))
    K = int(rng.integers(1, max(2, min(nq * nr, 3000000 if args.big else 200000))))
    k = int(rng.integers(1, min(64, nr) + 1))
    if args.only >= 0 and n_cases != args.only:
        n_cases += 1
        if n_cases > args.only:
            break
        continue
    a, b = make("1" if args.big else "2", d), make("0", d)
    for idx in (a, b):
        idx.add(r[:cut])
        idx.add(r[cut:])
    if os.environ.get("FUZZ_VERBOSE"):
        print(f"   cut={cut} K={K}", flush=True)
    ra, rb = a.global_topk(q, K), b.global_topk(q, K)
    assert len(ra[2]) == len(rb[2]) and np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1]) \
        and np.array_equal(bits(ra[2]), bits(rb[2])) and ra[3] == rb[3], ("topk", d, nq, nr, K, style)
    if os.environ.get("FUZZ_VERBOSE"):
        print(f"   topk ok; k={k}", flush=True)
    Da, Ia = a.search(q, k)
    Db, Ib = b.search(q, k)
    assert np.array_equal(Ia, Ib) and np.array_equal(bits(Da), bits(Db)), ("knn", d, nq, nr, k, style)
    if os.environ.get("FUZZ_VERBOSE"):
        print("   knn ok", flush=True)
    if nq * nr <= 4_000_000:
        radius = float(np.quantile(ra[2], 0.5)) if len(ra[2]) else 0.0
        la, xa, ya = a.range_search(q, radius)
        lb, xb, yb = b.range_search(q, radius)
        assert np.array_equal(la, lb) and np.array_equal(ya, yb) and np.array_equal(bits(xa), bits(xb)), \
            ("range", d, nq, nr, radius, style)
    n_cases += 1
print(f"fuzz ok: {n_cases} random cases (top-K, k-NN, range search) bit-identical between the pre-filtered and the "
      f"fp32 routes")
