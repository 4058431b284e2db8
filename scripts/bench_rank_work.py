#!/usr/bin/env python3
"""What ONE rank of an N-GPU configs[3] run has to do, measured on one GPU (round 5; the pool hands out one GPU per call, so
no N-GPU number can be taken -- this is the per-rank GPU work of the sharded pipeline, phase by phase, not a scaling run).

For world sizes W in --worlds the script builds configs[3]'s data once (40 000 query videos x 25 frames, 2 M reference rows,
2 M noise rows, 512-d, score normalisation) and then walks the sharded pipeline of engine.DeviceMatcher.match as rank 0 of W
would, with the other ranks' slices of every batch searched one after the other in the same process:

  * score normalisation of the rank's 1/W of the query rows (row L2 + 1-NN against the noise index) -- timed for rank 0;
  * the reference's batch schedule (dist.emulate_schedule, world 1): every batch is searched slice by slice against the W column
    slices of the references at exactly the schedule's radius; the time of EACH slice is accumulated separately -- the largest of
    them is the rank's search time, their sum what all ranks do together; events and counts run on the union of the lists (the
    ranks of a real run each sort 1/W of them);
  * pair-max of the hits and localisation of 1/W of the candidate pairs.

Transfers (the all-gather of 2 GB of query rows, the all-to-all of ~1 GB of kept hits, small collectives) are NOT in these
numbers: they do not exist inside one process."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from vsc2022_amd import _lib, dist as vdist
from vsc2022_amd.engine import DeviceMatcher, DeviceScoreNormalizer, sort_hits_device
from vsc2022_amd.vsc.index import FlatIndex

ap = argparse.ArgumentParser()
ap.add_argument("--worlds", type=int, nargs="+", default=[1, 2, 4, 8])
ap.add_argument("--query-videos", type=int, default=40000)
ap.add_argument("--ref-videos", type=int, default=40000)
args = ap.parse_args()
dev = torch.device("cuda", 0)
qf, rf, dim = 25, 50, 512
n_qv, n_rv = args.query_videos, args.ref_videos
refs = bench.synth_on_device(torch, dev, 1, n_rv, rf, dim)
queries = bench.synth_on_device(torch, dev, 1001, n_qv, qf, dim)
bench.plant_copies(torch, dev, 2001, queries, n_qv, qf, refs, n_rv, rf)
noise = bench.synth_on_device(torch, dev, 78, n_rv * rf, 1, dim, static_frac=0.0)
norm = DeviceScoreNormalizer(noise, beta=1.2)
del noise
r_off = np.arange(n_rv + 1, dtype=np.int64) * rf
q_off = np.arange(n_qv + 1, dtype=np.int64) * qf
matcher = DeviceMatcher(norm.refs(refs), r_off, 0)
del refs
qn = norm.queries(queries)
matcher.set_queries(qn, q_off)
K = 1200 * n_qv
nq, nr = int(qn.shape[0]), matcher.index.ntotal


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


# single process, for reference
matcher.match(bias=0.5)
t0 = sync()
norm.queries(queries)
t_norm1 = sync() - t0
t0 = sync()
matcher.search(K)
t_search1 = sync() - t0
out = {"single_process": {"score_norm_ms": round(t_norm1 * 1e3, 1), "search_ms": round(t_search1 * 1e3, 1)}, "worlds": {}}
print(json.dumps(out["single_process"]), flush=True)

for W in args.worlds:
    rows_r0 = vdist.shard_ranges(n_qv, W)[0]
    q0 = queries[rows_r0[0] * qf: rows_r0[1] * qf]
    norm.queries(q0)
    t0 = sync()
    norm.queries(q0)
    t_norm = sync() - t0
    # the W column slices (64-row aligned, as in DeviceMatcher.sharded_schedule_search)
    slices = []
    for r in range(W):
        c0, c1 = [(x // 64) * 64 for x in vdist.shard_ranges(nr, W)[r]]
        if r == W - 1:
            c1 = nr
        idx = FlatIndex(matcher.dim, _lib.METRIC_INNER_PRODUCT, 0)
        idx.use_torch_stream()
        idx.set_option("sort_hits", 0)
        idx.add(matcher.ref_feats[c0:c1])
        slices.append((c0, c1, idx))
    t_slice = [0.0] * W
    per_batch = {}
    t_misc = {"events_and_counts": 0.0}

    def head_budget(r0, n_here, share):
        return int(min(2.5 * K, 4.0 * K * n_here / max(r0, n_here)) * share) + (1 << 20)

    def search_rows(r0, r1, radius):
        parts = []
        for r, (c0, c1, idx) in enumerate(slices):
            idx.set_option("density_hint", min(1.0, K / (float(r0) * nr)) if r0 > 0 else 1.0)   # (as the engine does)
            t0 = sync()
            i, j, s = matcher._rows_above(qn[r0:r1], radius, head_budget(r0, r1 - r0, 1.3 / W), index=idx)
            dt = sync() - t0
            t_slice[r] += dt
            if r == 0:
                per_batch[(r0, r1)] = dt
            parts.append((i + r0, j + c0, s))
        return (torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts]), torch.cat([p[2] for p in parts]))

    for rep in range(2):       # (the first walk warms every slice index: its int8 image, its buffers)
        for r in range(W):
            t_slice[r] = 0.0
        t0 = sync()
        radius, hi, hj, hs = vdist.emulate_schedule(search_rows, nq, K, None, dev)
        t_emulate = sync() - t0
    t_misc["events_and_counts"] = t_emulate - sum(t_slice)
    t0 = sync()
    hi, hj, hs = sort_hits_device(hi, hj, hs, nq, nr)
    hi, hj, hs = hi[:K], hj[:K], hs[:K]
    t_sort = sync() - t0
    # candidate generation + localisation of the rank's share (1/W of the hits' rows, 1/W of the pairs)
    mine = hi < (rows_r0[1] * qf)
    t0 = sync()
    pq, pr, ps, pf = matcher.pair_max(hi[mine], hj[mine], hs[mine])
    t_pair = sync() - t0
    n_loc = min(int(ps.numel()), 5 * n_qv // W)
    matcher.localize(pq[:n_loc], pr[:n_loc], 0.5)
    t0 = sync()
    matcher.localize(pq[:n_loc], pr[:n_loc], 0.5)
    t_tn = sync() - t0
    rec = {
        "score_norm_ms": round(t_norm * 1e3, 1),
        "search_slice_max_ms": round(max(t_slice) * 1e3, 1), "search_slices_sum_ms": round(sum(t_slice) * 1e3, 1),
        "events_and_counts_all_lists_ms": round(t_misc["events_and_counts"] * 1e3, 1),
        "final_sort_all_hits_ms": round(t_sort * 1e3, 1), "pair_max_ms": round(t_pair * 1e3, 1), "tn_ms": round(t_tn * 1e3, 1),
        "hits": int(hs.numel()), "radius": float(radius),
    }
    rec["rank_gpu_work_ms"] = round(rec["score_norm_ms"] + rec["search_slice_max_ms"] + (rec["events_and_counts_all_lists_ms"]
                                    + rec["final_sort_all_hits_ms"]) / W + rec["pair_max_ms"] + rec["tn_ms"], 1)
    out["worlds"][str(W)] = rec
    if os.environ.get("VSC_RANK_WORK_BATCHES") == "1":   # slice 0's time per batch of the schedule (second walk)
        print(W, "batches", " ".join(f"{a}:{dt * 1e3:.2f}" for (a, b), dt in sorted(per_batch.items())), flush=True)
    print(W, json.dumps(rec), flush=True)
    del slices
    torch.cuda.empty_cache()
print(json.dumps(out))
