#!/usr/bin/env python3
"""Soak test of the query-sharded engine: random datasets, 2-4 ranks sharing the one GPU of the box (gloo) -- the global
candidate table and the localisation results must equal the single-process engine's bit for bit.  Both designs of the
sharded search are drawn (engine.DeviceMatcher.sharded_schedule_search): the default column mode (every batch of the
reference's schedule split by reference columns) and, for cases with `seed_rows`, the row-list mode with lists prepared
from a row sample of random size (VSC_SHARD_SPEC_START: incl. samples so small that the predicted floor is often too
high and batches are searched on demand).  Static videos and descriptors on a coarse grid are part of the draw -- exact
score ties, also ON the K cut, where the result depends on what the reference's schedule does with the tied hits
(vsc2022_amd/dist.py, module docstring); every run must report `matches_reference`.

    python scripts/fuzz_sharded.py --seconds 120 --seed 0
"""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def pack(videos):
    feats = np.concatenate([v.feature for v in videos]).astype(np.float32)
    off = np.r_[0, np.cumsum([len(v.feature) for v in videos])].astype(np.int64)
    return feats, off


def arrays(res):
    return dict(cq=res.cand_q.cpu().numpy(), cr=res.cand_r.cpu().numpy(), cs=res.cand_score.cpu().numpy(),
                loc=res.loc_index.cpu().numpy(), nbox=res.nbox.cpu().numpy(), boxes=res.boxes.cpu().numpy(),
                bscore=res.box_score.cpu().numpy(),
                n=np.array([res.n_hits, res.n_candidates, res.n_localized, res.n_matches]),
                flags=np.array([res.matches_reference, res.tie_on_cut, res.ties_dropped]))


def dataset(case):
    from vsc2022_amd import synth

    q, r = synth.make_dataset(seed=case["seed"], n_query=case["n_query"], n_ref=case["n_ref"], dim=case["dim"],
                              q_frames=case["qf"], r_frames=case["rf"], planted_frac=case["planted"],
                              static_frac=case["static"], dist=case.get("dist", "gaussian"))[:2]
    if case.get("grid"):
        for v in q + r:
            v.feature[:] = np.round(v.feature * case["grid"]) / case["grid"]
    return q, r


def worker(rank, world, port, out_dir, case):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0")
    if case["seed_rows"]:
        os.environ["VSC_SHARD_SPEC_START"] = str(case["seed_rows"])
        os.environ["VSC_SHARD_MODE"] = "rows"   # cases with prepared row lists; the others run the default column mode
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vsc2022_amd import dist as vdist
        from vsc2022_amd.engine import DeviceMatcher

        q, r = dataset(case)
        rf, roff = pack(r)
        lo, hi = vdist.shard_ranges(len(q), world)[rank]
        qf, qoff = pack(q[lo:hi])
        m = DeviceMatcher(rf, roff, 0)
        m.set_queries(qf, qoff)
        res = m.match(n_qvid_global=len(q), qvid_base=lo, row_base=sum(len(v.feature) for v in q[:lo]), bias=case["bias"])
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **arrays(res))
    finally:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    from vsc2022_amd.engine import DeviceMatcher

    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    n_cases = n_seeded = n_tie = n_drop = 0
    while time.time() - t0 < args.seconds:
        lo_f = int(rng.integers(4, 20))
        case = dict(seed=int(rng.integers(1 << 30)), n_query=int(rng.integers(12, 160)), n_ref=int(rng.integers(20, 300)),
                    dim=int(rng.choice([32, 64, 128, 256, 512])), qf=(lo_f, lo_f + int(rng.integers(0, 30))),
                    rf=(lo_f, lo_f + int(rng.integers(0, 40))), planted=float(rng.uniform(0.0, 0.5)),
                    static=float(rng.choice([0.0, 0.05, 0.3])), grid=int(rng.choice([0, 0, 2, 4, 8])),
                    bias=float(rng.choice([0.0, 0.0, 0.5])),
                    seed_rows=int(rng.choice([0, 0, 0, 8, 40, 150, 600])),
                    # the distribution class of the rows (vsc2022_amd/synth.py): half of the cases are not isotropic
                    dist=str(rng.choice(["gaussian"] * 5 + ["clusters", "powerlaw", "offset", "temporal", "neardup"])))
        world = int(rng.integers(2, 5))
        q, r = dataset(case)
        if len(q) < world:
            continue
        rf, roff = pack(r)
        qf, qoff = pack(q)
        m = DeviceMatcher(rf, roff, 0)
        m.set_queries(qf, qoff)
        single = arrays(m.match(bias=case["bias"]))
        del m
        torch.cuda.empty_cache()
        with tempfile.TemporaryDirectory() as td:
            mp.spawn(worker, args=(world, 29000 + int(rng.integers(0, 900)), td, case), nprocs=world, join=True)
            parts = [dict(np.load(os.path.join(td, f"rank{k}.npz"))) for k in range(world)]
        for p in parts:
            ok = (np.array_equal(p["cq"], single["cq"]) and np.array_equal(p["cr"], single["cr"]) and
                  np.array_equal(p["cs"].view(np.uint32), single["cs"].view(np.uint32)) and np.array_equal(p["n"], single["n"]))
            assert ok, f"case {n_cases} {case} world {world}: candidate tables differ"
            assert p["flags"][0], f"case {n_cases} {case} world {world}: result not proven to be the reference's"
        n_loc = int(single["n"][2])
        nbox = np.full(n_loc, -1, dtype=np.int64)
        boxes = np.zeros((n_loc, 16, 4), dtype=np.int64)
        bscore = np.zeros((n_loc, 16), dtype=np.float32)
        for p in parts:
            nbox[p["loc"]] = p["nbox"]
            boxes[p["loc"]] = p["boxes"]
            bscore[p["loc"]] = p["bscore"]
        assert np.array_equal(nbox, single["nbox"]), f"case {n_cases} {case} world {world}: box counts differ"
        for k in range(n_loc):
            assert np.array_equal(boxes[k, : nbox[k]], single["boxes"][k, : nbox[k]])
            assert np.array_equal(bscore[k, : nbox[k]].view(np.uint32), single["bscore"][k, : nbox[k]].view(np.uint32))
        n_cases += 1
        n_seeded += 1 if case["seed_rows"] else 0
        n_tie += int(parts[0]["flags"][1])
        n_drop += int(parts[0]["flags"][2])
    print(f"fuzz ok: {n_cases} random datasets ({n_seeded} in row-list mode with prepared lists, the others in column mode; {n_tie} with a tie on the K cut, "
          f"{n_drop} of them with the tied hits dropped as the reference drops them), 2-4 ranks on one GPU, candidate "
          f"tables and localisation equal to the single-process engine")


if __name__ == "__main__":
    main()
