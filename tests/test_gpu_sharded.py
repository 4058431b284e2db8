"""GPU: the query-sharded engine (DeviceMatcher.match with world_size 2) equals the single-process
engine.  Both ranks share the one GPU of the test box, so the collectives run over gloo (staged
through the host); on a multi-GPU node the same code runs over RCCL."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _data(seed=5):
    sys.path.insert(0, ROOT)
    from vsc2022_amd import synth

    shape = {5: (64, 120, 128), 6: (37, 80, 64), 7: (90, 60, 256)}[seed]
    q, r, gts = synth.make_dataset(seed=seed, n_query=shape[0], n_ref=shape[1], dim=shape[2], q_frames=(8, 30),
                                   r_frames=(8, 40), planted_frac=0.3, static_frac=0.0)
    return q, r


def _pack(videos):
    feats = np.concatenate([v.feature for v in videos]).astype(np.float32)
    off = np.r_[0, np.cumsum([len(v.feature) for v in videos])].astype(np.int64)
    return feats, off


def _result_arrays(res):
    nbox = res.nbox.cpu().numpy()
    return dict(cq=res.cand_q.cpu().numpy(), cr=res.cand_r.cpu().numpy(), cs=res.cand_score.cpu().numpy(),
                loc=res.loc_index.cpu().numpy(), nbox=nbox, boxes=res.boxes.cpu().numpy(),
                bscore=res.box_score.cpu().numpy(), n=np.array([res.n_hits, res.n_candidates, res.n_localized, res.n_matches]))


def _worker(rank, world, port, out_dir, seed, seed_rows):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0")
    if seed_rows:
        # a sample of `seed_rows` rows over all ranks seeds every rank's local search (engine.DeviceMatcher.seed_radius);
        # by default these query sets are too small for the sample to be taken
        os.environ["VSC_SHARD_SEED_ROWS"] = str(seed_rows)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        from vsc2022_amd import dist as vdist
        from vsc2022_amd.engine import DeviceMatcher

        q, r = _data(seed)
        rf, roff = _pack(r)
        lo, hi = vdist.shard_ranges(len(q), world)[rank]
        qf, qoff = _pack(q[lo:hi])
        m = DeviceMatcher(rf, roff, 0)
        m.set_queries(qf, qoff)
        row_base = sum(len(v.feature) for v in q[:lo])
        res = m.match(n_qvid_global=len(q), qvid_base=lo, row_base=row_base)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **_result_arrays(res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("seed,world,seed_rows", [(5, 2, 0), (6, 3, 0), (7, 2, 0), (5, 2, 120), (6, 3, 40), (7, 2, 300), (7, 3, 12)])
def test_sharded_engine_equals_single_process(gpu, tmp_path, seed, world, seed_rows):
    from vsc2022_amd.engine import DeviceMatcher

    q, r = _data(seed)
    rf, roff = _pack(r)
    qf, qoff = _pack(q)
    m = DeviceMatcher(rf, roff, 0)
    m.set_queries(qf, qoff)
    single = _result_arrays(m.match())
    del m
    torch.cuda.empty_cache()
    port = 29650 + os.getpid() % 500
    mp.spawn(_worker, args=(world, port, str(tmp_path), seed, seed_rows), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"rank{k}.npz") for k in range(world)]
    for p in parts:  # every rank holds the same global candidate table = the single-process one
        assert np.array_equal(p["cq"], single["cq"]) and np.array_equal(p["cr"], single["cr"])
        assert np.array_equal(p["cs"].view(np.uint32), single["cs"].view(np.uint32))
        assert np.array_equal(p["n"], single["n"])
    # localisation results, reassembled by candidate index
    n_loc = int(single["n"][2])
    nbox = np.full(n_loc, -1, dtype=np.int64)
    boxes = np.zeros((n_loc, 16, 4), dtype=np.int64)
    bscore = np.zeros((n_loc, 16), dtype=np.float32)
    for p in parts:
        nbox[p["loc"]] = p["nbox"]
        boxes[p["loc"]] = p["boxes"]
        bscore[p["loc"]] = p["bscore"]
    assert np.array_equal(nbox, single["nbox"])
    for k in range(n_loc):
        assert np.array_equal(boxes[k, : nbox[k]], single["boxes"][k, : nbox[k]])
        assert np.array_equal(bscore[k, : nbox[k]].view(np.uint32), single["bscore"][k, : nbox[k]].view(np.uint32))
    assert single["n"][3] > 0
