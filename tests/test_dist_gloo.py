"""CPU, world_size 2, gloo: the query-sharded merge (vsc2022_amd/dist.py) reproduces the
single-process pipeline.  The per-rank search results are supplied by the CPU oracle (there is no
GPU here); what is under test is the distributed logic: the exact global-K selection, the local
budget doubling for skewed shards, the candidate cut and the variable-length all-gather."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dataset(skewed):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
    from vsc2022_amd import synth

    q, r, _ = synth.make_dataset(seed=77, n_query=16, n_ref=20, dim=32, q_frames=(6, 14), r_frames=(6, 14),
                                 planted_frac=0.5 if not skewed else 0.0)
    if skewed:  # every high score sits in the first shard
        for k in range(4):
            n = min(len(q[k].feature), len(r[k].feature))
            q[k].feature[:n] = r[k].feature[:n]
    return q, r


def _worker(rank, world, port, skewed, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
        import oracle as orc
        from vsc2022_amd import dist as vdist

        q, r, = _dataset(skewed)
        R = np.concatenate([v.feature for v in r])
        row2r = np.repeat(np.arange(len(r), dtype=np.int32), [len(v.feature) for v in r])
        nqv = len(q)
        K = 40 * nqv          # scaled-down 1200/video
        M = 3 * nqv           # scaled-down 25/video
        lo, hi = vdist.shard_ranges(nqv, world)[rank]
        mine = q[lo:hi]
        Q = np.concatenate([v.feature for v in mine])
        row_base = sum(len(v.feature) for v in q[:lo])
        row2q = np.repeat(np.arange(len(mine), dtype=np.int32), [len(v.feature) for v in mine])
        calls = []

        def local_search(k_local):
            calls.append(k_local)
            i, j, s, info = orc.global_threshold_search(Q, R, k_local, return_info=True)
            return (torch.from_numpy(i), torch.from_numpy(j), torch.from_numpy(s), info["radius"])

        hi_, hj_, hs_, tau = vdist.sharded_hits(local_search, Q.shape[0] * R.shape[0], K,
                                                k_local_start=K // 8 if skewed else None)
        pq, pr, ps, pf = orc.pair_max(hi_.numpy(), hj_.numpy(), hs_.numpy(), row2q, row2r)
        first_i = torch.from_numpy(hi_.numpy()[pf] + row_base) if len(pf) else torch.zeros(0, dtype=torch.int64)
        first_j = torch.from_numpy(hj_.numpy()[pf]) if len(pf) else torch.zeros(0, dtype=torch.int64)
        cands = vdist.merge_candidates(torch.from_numpy(pq + lo), torch.from_numpy(pr), torch.from_numpy(ps),
                                       first_i, first_j, M)
        n_hits = torch.tensor([int(hs_.numel())])
        dist.all_reduce(n_hits)
        if rank == 0:
            np.savez(out_path, q=cands.q_vid.numpy(), r=cands.r_vid.numpy(), s=cands.score.numpy(),
                     n_hits=n_hits.numpy(), calls=np.array(calls), tau=np.float32(tau))
    finally:
        dist.destroy_process_group()


def _single(skewed):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
    import oracle as orc

    q, r = _dataset(skewed)
    Q = np.concatenate([v.feature for v in q])
    R = np.concatenate([v.feature for v in r])
    row2q = np.repeat(np.arange(len(q), dtype=np.int32), [len(v.feature) for v in q])
    row2r = np.repeat(np.arange(len(r), dtype=np.int32), [len(v.feature) for v in r])
    K, M = 40 * len(q), 3 * len(q)
    i, j, s = orc.global_threshold_search(Q, R, K)
    pq, pr, ps, _ = orc.pair_max(i, j, s, row2q, row2r)
    return len(s), pq[:M], pr[:M], ps[:M], s[-1]


@pytest.mark.parametrize("skewed", [False, True])
def test_sharded_merge_equals_single_process(tmp_path, skewed):
    port = 29500 + (os.getpid() % 2000) + (7 if skewed else 0)
    out = str(tmp_path / "rank0.npz")
    mp.spawn(_worker, args=(2, port, skewed, out), nprocs=2, join=True)
    got = np.load(out)
    n_hits, pq, pr, ps, tau = _single(skewed)
    assert int(got["n_hits"][0]) == n_hits
    assert np.float32(got["tau"]) == np.float32(tau)
    assert np.array_equal(got["q"], pq) and np.array_equal(got["r"], pr)
    assert np.array_equal(got["s"].view(np.uint32), ps.view(np.uint32))
    if skewed:
        assert len(got["calls"]) >= 2, "the skewed shard must have forced a larger local budget"


def test_prefix_select_single_process_properties():
    sys.path[:0] = [ROOT]
    from vsc2022_amd.dist import distributed_prefix_select

    rng = np.random.default_rng(0)
    for trial in range(100):
        n = int(rng.integers(0, 400))
        x = rng.normal(size=n).astype(np.float32)
        if trial % 3 == 0:
            x = np.round(x * 2) / 2
        x = np.sort(x)[::-1].copy()
        k = int(rng.integers(0, 500))
        n_take, tau = distributed_prefix_select(torch.from_numpy(x), k)
        assert n_take == min(n, k)
        if 0 < k < n:
            assert tau == x[k - 1]


def _knn_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
        import oracle as orc
        from vsc2022_amd import dist as vdist

        rng = np.random.default_rng(3)
        q = rng.standard_normal((40, 16)).astype(np.float32)
        r = rng.standard_normal((90, 16)).astype(np.float32)
        r[50:60] = r[10:20]  # exact duplicates across the shard boundary -> ties resolved by global id
        lo, hi = vdist.shard_ranges(len(r), world)[rank]
        D, I = orc.knn(q, r[lo:hi], 7)
        I = np.where(I >= 0, I + lo, -1)
        gD, gI = vdist.ref_sharded_knn(torch.from_numpy(D), torch.from_numpy(I), 7)
        if rank == 0:
            np.savez(out_path, D=gD.numpy(), I=gI.numpy())
    finally:
        dist.destroy_process_group()


def test_ref_sharded_knn_merge_equals_single_index(tmp_path):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
    import oracle as orc

    out = str(tmp_path / "knn.npz")
    mp.spawn(_knn_worker, args=(3, 29900 + os.getpid() % 1000, out), nprocs=3, join=True)
    got = np.load(out)
    rng = np.random.default_rng(3)
    q = rng.standard_normal((40, 16)).astype(np.float32)
    r = rng.standard_normal((90, 16)).astype(np.float32)
    r[50:60] = r[10:20]
    D, I = orc.knn(q, r, 7)
    assert np.array_equal(got["I"], I) and np.array_equal(got["D"].view(np.uint32), D.view(np.uint32))


def _select_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path[:0] = [ROOT]
        from vsc2022_amd.dist import distributed_prefix_select

        rng = np.random.default_rng(100 + rank)
        res = []
        for case in range(12):
            crng = np.random.default_rng(1000 + case)  # same on every rank
            k = int(crng.integers(1, 120))
            n = int(rng.integers(0, 200))              # some lists are longer than k, some empty
            x = rng.normal(size=n).astype(np.float32)
            if case % 2 == 0:
                x = np.round(x * 3) / 3                # heavy ties, also across ranks
            x = np.sort(x)[::-1].copy()
            n_take, tau = distributed_prefix_select(torch.from_numpy(x), k)
            res.append((case, k, x, n_take, tau))
        torch.save(res, f"{out_path}.{rank}")
    finally:
        dist.destroy_process_group()


def test_prefix_select_world3_ties_and_long_lists(tmp_path):
    """Exact global top-k prefix per rank under (score desc, rank asc, position asc), lists sorted, with
    ties across ranks and lists longer than k (only their leading k elements can matter)."""
    out = str(tmp_path / "sel")
    world = 3
    mp.spawn(_select_worker, args=(world, 29950 + os.getpid() % 1000, out), nprocs=world, join=True)
    per_rank = [torch.load(f"{out}.{r}", weights_only=False) for r in range(world)]
    for case in range(12):
        k = per_rank[0][case][1]
        lists = [per_rank[r][case][2] for r in range(world)]
        order = sorted(((-float(v), r, p) for r in range(world) for p, v in enumerate(lists[r])))
        top = order[:k]
        for r in range(world):
            want = sum(1 for _, rr, _ in top if rr == r)
            assert per_rank[r][case][3] == want, (case, r, per_rank[r][case][3], want)
        if len(order) > k:
            assert all(np.float32(per_rank[r][case][4]) == np.float32(-top[-1][0]) for r in range(world))


# ---------------------------------------------------------------- one shard holds the whole top-K
def _skew_all_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path[:0] = [ROOT]
        from vsc2022_amd import dist as vdist

        K = 100
        rng = np.random.default_rng(50 + rank)
        # rank 0: 400 scores near 1 (the entire global top-K and more); the others: noise near 0
        full = (0.9 + 0.1 * rng.random(400) if rank == 0 else 0.1 * rng.random(300)).astype(np.float32)
        full = np.sort(full)[::-1].copy()
        calls = []

        def local_search(k_local):
            calls.append(k_local)
            s = torch.from_numpy(full[:k_local].copy())
            z = torch.zeros(len(s), dtype=torch.int32)
            return z, z, s, float("-inf")

        hi, hj, hs, tau = vdist.sharded_hits(local_search, 10 ** 9, K, k_local_start=63)
        torch.save((hs.numpy(), tau, calls), f"{out_path}.{rank}")
    finally:
        dist.destroy_process_group()


def test_sharded_hits_terminates_when_one_shard_holds_the_whole_topk(tmp_path):
    """Rank 0's list is cut at k_local == K with the global cut equal to its last element (never 'exact' by the
    tau > cut rule); rank 1 is exact from the start.  Every rank must leave the loop on the same iteration."""
    out = str(tmp_path / "skew")
    mp.spawn(_skew_all_worker, args=(2, 29300 + os.getpid() % 500, out), nprocs=2, join=True)
    r0 = torch.load(f"{out}.0", weights_only=False)
    r1 = torch.load(f"{out}.1", weights_only=False)
    assert len(r0[0]) == 100 and len(r1[0]) == 0
    assert r0[2][-1] == 101, r0[2]       # rank 0 doubled its budget up to K + 1 (the selection looks one past the cut)
    assert r1[2] == [63], r1[2]          # the exact rank kept its first result (no repeated search)
    assert np.float32(r0[1]) == np.float32(r1[1]) == r0[0][-1]


# ---------------------------------------------------------------- seeded local searches (round 4: no schedule replay per rank)
def _seeded_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path[:0] = [ROOT]
        from vsc2022_amd import dist as vdist

        rng = np.random.default_rng(70 + rank)
        full = np.sort(rng.random(500 + 100 * rank).astype(np.float32))[::-1].copy()
        full[40:44] = full[40]  # ties inside every list
        out = {}
        for name, K, seed in (("good", 300, 0.55), ("too_high", 300, 0.93), ("one_rank_full", 120, 0.05)):
            calls = []

            def local_search(k_local, seed=seed, K=K, calls=calls):
                # first call: seeded -- every local score beyond the seed, cut at K; later calls: the plain budgeted search
                if not calls:
                    calls.append(("seeded", K))
                    s = torch.from_numpy(full[full > np.float32(seed)][:K].copy())
                    z = torch.zeros(len(s), dtype=torch.int32)
                    return z, z, s, float(seed), True
                calls.append(("plain", k_local))
                s = torch.from_numpy(full[:k_local].copy())
                z = torch.zeros(len(s), dtype=torch.int32)
                return z, z, s, float("-inf")

            hi, hj, hs, tau = vdist.sharded_hits(local_search, 10 ** 9, K)
            out[name] = (hs.numpy(), tau, calls)
        torch.save((full, out), f"{out_path}.{rank}")
    finally:
        dist.destroy_process_group()


def test_sharded_hits_with_seeded_local_searches(tmp_path):
    """world 3: a seed below the global cut gives the exact top-K in ONE seeded search per rank; a seed above it (fewer
    than K hits over all ranks) sends every rank through the plain search; a rank whose seeded list is full (K hits) is
    exact although its own cut equals its last element."""
    out = str(tmp_path / "seeded")
    world = 3
    mp.spawn(_seeded_worker, args=(world, 29700 + os.getpid() % 200, out), nprocs=world, join=True)
    got = [torch.load(f"{out}.{r}", weights_only=False) for r in range(world)]
    lists = [g[0] for g in got]
    for name, K in (("good", 300), ("too_high", 300), ("one_rank_full", 120)):
        order = sorted(((-float(v), r, p) for r in range(world) for p, v in enumerate(lists[r])))[:K]
        for r in range(world):
            want = [lists[r][p] for _, rr, p in order if rr == r]
            hs, tau, calls = got[r][1][name]
            assert np.array_equal(hs, np.array(want, dtype=np.float32)), (name, r)
            assert np.float32(tau) == np.float32(-order[-1][0])
            if name == "good":
                assert calls == [("seeded", K)], calls
            if name == "too_high":
                assert calls[0] == ("seeded", K) and calls[1][0] == "plain", calls
    # 'one_rank_full': the seed 0.05 leaves > 120 scores on every rank: all lists are full, one search each
    assert all(got[r][1]["one_rank_full"][2] == [("seeded", 120)] for r in range(world))


# ---------------------------------------------------------------- reference-sharded index (configs[4])
class _OracleIndex:
    """FlatIndex stand-in over the CPU oracle (there is no GPU here): what is under test is refshard.py."""

    def __init__(self, rows, metric_type=0):
        self.rows = np.ascontiguousarray(rows, dtype=np.float32)
        self.ntotal = len(self.rows)
        self.metric_type = metric_type

    def search(self, x, k):
        import oracle as orc

        return orc.knn(np.ascontiguousarray(x, dtype=np.float32), self.rows, k, self.metric_type)

    def global_topk(self, x, K, device_out=False):
        import oracle as orc

        i, j, s, info = orc.global_threshold_search(np.ascontiguousarray(x, dtype=np.float32), self.rows, K,
                                                    return_info=True)
        return i, j, s, info["radius"]


def _refshard_data():
    rng = np.random.default_rng(11)
    q = rng.standard_normal((70, 24)).astype(np.float32)
    r = rng.standard_normal((260, 24)).astype(np.float32)
    r[200:215] = r[20:35]      # duplicate rows in different shards: exact score ties across ranks
    q[5] = q[6]                # and duplicate query rows: ties resolved by (row, ref)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    r /= np.linalg.norm(r, axis=1, keepdims=True)
    return q, r


def _refshard_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
        from vsc2022_amd import dist as vdist
        from vsc2022_amd.refshard import RefShardedIndex

        q, r = _refshard_data()
        lo, hi = vdist.shard_ranges(len(r), world)[rank]
        idx = RefShardedIndex(_OracleIndex(r[lo:hi]), lo, len(r))
        D, I = idx.search(q, 9)
        res = {"D": D, "I": I}
        # the same shards behind an L2 index: distances ascend, ties by id (vsc/index.py:171 with faiss.METRIC_L2)
        res["D2"], res["I2"] = RefShardedIndex(_OracleIndex(r[lo:hi], 1), lo, len(r)).search(q, 7)
        for K in (50, 700, 5000):
            i, j, s, tau = idx.global_topk(q, K, k_local_start=K // 6 + 1)
            res[f"i{K}"], res[f"j{K}"], res[f"s{K}"] = i.numpy(), j.numpy(), s.numpy()
        np.savez(f"{out_path}.{rank}.npz", **res)
    finally:
        dist.destroy_process_group()


def test_ref_sharded_index_equals_single_index(tmp_path):
    """world 3: per-row k-NN and the global top-K over column shards equal the single-index results bit for
    bit (ties across shards included); every rank holds the same merged result."""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
    import oracle as orc

    out = str(tmp_path / "rs")
    world = 3
    mp.spawn(_refshard_worker, args=(world, 29400 + os.getpid() % 500, out), nprocs=world, join=True)
    q, r = _refshard_data()
    D, I = orc.knn(q, r, 9)
    D2, I2 = orc.knn(q, r, 7, 1)
    for rank in range(world):
        got = np.load(f"{out}.{rank}.npz")
        assert np.array_equal(got["I"], I) and np.array_equal(got["D"].view(np.uint32), D.view(np.uint32))
        assert np.array_equal(got["I2"], I2) and np.array_equal(got["D2"].view(np.uint32), D2.view(np.uint32))
        for K in (50, 700, 5000):
            # K small enough that the reference's schedule never re-thresholds on a tie: plain exact top-K
            S = (q @ r.T).astype(np.float32)
            i, j, s = orc.global_threshold_search(q, r, K)
            assert np.array_equal(got[f"i{K}"], i) and np.array_equal(got[f"j{K}"], j), (rank, K)
            assert np.array_equal(got[f"s{K}"].view(np.uint32), s.view(np.uint32))


def test_bench_self_launches_n_ranks():
    """`python bench.py --gpus N` with no launcher must start N ranks itself (the driver's multi-GPU command goes
    through torchrun, a user's may not): the launch + rendezvous path alone, over gloo, no GPU touched."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--launch-check"], cwd=root,
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1, out.stdout                       # rank 0 alone prints
    rep = json.loads(line[0])
    assert rep["n_gpus"] == 3 and rep["process_group"]["ranks_answered"] == 3
    assert rep["process_group"]["backend"] == "gloo" and rep["process_group"]["launcher"] == "bench.py self-launch"
    assert sorted(rep["process_group"]["devices"]) == [0, 1, 2]


# ---------------------------------------------------------------- round 5: the sharded result is provably the reference's
def _tie_data(seed=0, nq=300, nr=400, dim=8):
    """Rows on a coarse grid (massive exact score ties), static "videos" (duplicate rows) and continuous rows."""
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((nq, dim))
    r = rng.standard_normal((nr, dim))
    q[: nq // 2] = np.round(q[: nq // 2] * 2) / 2
    r[: nr // 2] = np.round(r[: nr // 2] * 2) / 2
    q[40:52] = q[40]
    q[nq - 100: nq - 70] = q[nq - 100]
    r[100:120] = r[100]
    return q.astype(np.float32), r.astype(np.float32)


def _tie_classes(q, r, want=3):
    """K values by what happens on the cut: no tie / tie kept by the reference / tie dropped by the reference."""
    import oracle as orc

    S = np.sort(orc.scores(q, r).ravel())[::-1]
    cls = {"notie": [], "kept": [], "dropped": []}
    for K in list(range(40, 4000, 31)) + list(range(4000, 70000, 797)):
        tie = S[K - 1] == S[K]
        if all(len(v) >= want for v in cls.values()):
            break
        name = "notie"
        if tie:
            _, _, s, info = orc.global_threshold_search(q, r, K, return_info=True)
            name = "dropped" if np.float32(info["radius"]) == S[K - 1] else "kept"
        if len(cls[name]) < want:
            cls[name].append(K)
    assert all(len(v) >= 1 for v in cls.values()), {k: len(v) for k, v in cls.items()}
    return cls


def test_prefix_select_reports_tie_on_cut():
    sys.path[:0] = [ROOT]
    from vsc2022_amd.dist import distributed_prefix_select

    rng = np.random.default_rng(5)
    seen = set()
    for trial in range(300):
        n = int(rng.integers(1, 300))
        x = rng.normal(size=n).astype(np.float32)
        if trial % 2 == 0:
            x = np.round(x * 2) / 2
        x = np.sort(x)[::-1].copy()
        k = int(rng.integers(1, 320))
        n_take, tau, info = distributed_prefix_select(torch.from_numpy(x), k, return_info=True)
        want = k < n and x[k - 1] == x[k]
        assert info.tie_on_cut == want, (trial, n, k)
        assert info.total == n
        if k < n:
            assert info.n_above == int((x > x[k - 1]).sum())
        seen.add(want)
    assert seen == {True, False}


def _qshard_tie_worker(rank, world, port, Ks, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
        import oracle as orc
        from vsc2022_amd import dist as vdist

        q, r = _tie_data()
        lo, hi = vdist.shard_ranges(len(q), world)[rank]
        Q = q[lo:hi]
        res = {}
        for K in Ks:
            def local_search(k_local):
                i, j, s, info = orc.global_threshold_search(Q, r, k_local, return_info=True)
                return torch.from_numpy(i), torch.from_numpy(j), torch.from_numpy(s), info["radius"]

            def range_scores(r0, r1, radius):  # this rank's rows of the batch [r0, r1) of the GLOBAL row order
                a, b = max(r0, lo) - lo, min(r1, hi) - lo
                if b <= a:
                    return torch.zeros(0, dtype=torch.float32)
                return torch.from_numpy(orc.range_search(Q[a:b], r, radius)[1])

            hi_, hj_, hs_, tau, info = vdist.sharded_hits(local_search, Q.shape[0] * r.shape[0], K,
                                                          k_local_start=max(1, K // 5), return_info=True)
            keep, proven, dropped = vdist.resolve_tie_on_cut(
                hs_, tau, info, lambda: vdist.emulate_schedule_radius(range_scores, len(q), K))
            res[f"i{K}"], res[f"j{K}"], res[f"s{K}"] = (hi_[keep].numpy() + lo, hj_[keep].numpy(), hs_[keep].numpy())
            res[f"f{K}"] = np.array([info.tie_on_cut, proven, dropped])
        np.savez(f"{out_path}.{rank}.npz", **res)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_query_sharded_hits_are_the_references_with_ties_on_the_cut(tmp_path, world):
    """A tie sits exactly on the K cut (s_K == s_(K+1)): the sharded search must return what vsc/index.py:142-165
    returns -- all of the top K when the reference's final radius lies below the tie, NONE of the tied hits when the
    reference's schedule ends on that score -- and must say which case it proved."""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
    import oracle as orc

    q, r = _tie_data()
    cls = _tie_classes(q, r)
    Ks = sorted(set(sum(cls.values(), [])))
    out = str(tmp_path / "qs")
    mp.spawn(_qshard_tie_worker, args=(world, 29100 + os.getpid() % 800, Ks, out), nprocs=world, join=True)
    parts = [np.load(f"{out}.{k}.npz") for k in range(world)]
    for K in Ks:
        i, j, s = orc.global_threshold_search(q, r, K)
        gi = np.concatenate([p[f"i{K}"] for p in parts])
        gj = np.concatenate([p[f"j{K}"] for p in parts])
        gs = np.concatenate([p[f"s{K}"] for p in parts])
        order = np.lexsort((gj, gi, -gs.astype(np.float64)))
        assert np.array_equal(gi[order], i) and np.array_equal(gj[order], j), K
        assert np.array_equal(gs[order].view(np.uint32), s.view(np.uint32)), K
        flags = parts[0][f"f{K}"]
        assert all(np.array_equal(p[f"f{K}"], flags) for p in parts)
        assert bool(flags[1]), K  # proven
        assert bool(flags[0]) == (K not in cls["notie"]) and bool(flags[2]) == (K in cls["dropped"]), (K, flags)
        if K in cls["dropped"]:
            assert len(s) < K


def _emulate_worker(rank, world, port, Ks, out_path, by_cols=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
        import oracle as orc
        from vsc2022_amd import dist as vdist

        q, r = _tie_data()
        lo, hi = vdist.shard_ranges(len(q), world)[rank]
        Q = q[lo:hi]
        rng = np.random.default_rng(11 + rank)
        res = {}
        for K in Ks:
            # lists prepared "beforehand" for the batches behind row 96, at floors that are sometimes too high (the
            # emulation must then search the batch at the schedule's own radius), sometimes far too low
            prepared, stats = {}, dict(used=0, redone=0, demand=0)
            S_all = np.sort(orc.scores(Q, r).ravel())[::-1] if len(Q) else np.zeros(0, np.float32)
            for r0, r1 in vdist.exponential_batches(len(q)):
                a, b = max(r0, lo) - lo, min(r1, hi) - lo
                if r0 < 96 or b <= a:
                    continue
                dens = min(1.0, float(rng.choice([0.6, 2.3, 6.0])) * K / (r0 * len(r)))
                floor = float(S_all[min(len(S_all) - 1, int(dens * len(S_all)))])
                lims, D, I = orc.range_search(Q[a:b], r, floor)
                rows = np.repeat(np.arange(a, b), np.diff(lims.astype(np.int64))).astype(np.int32)
                prepared[(a, b)] = (floor, torch.from_numpy(rows), torch.from_numpy(I.astype(np.int32)), torch.from_numpy(D))

            c0, c1 = vdist.shard_ranges(len(r), world)[rank]

            def search_rows(r0, r1, radius):
                if by_cols and r0 < 96:
                    # the head of the query set split by reference COLUMNS: every rank searches all of the batch's rows
                    # against its slice (GLOBAL row numbers until the hand-over)
                    lims, D, I = orc.range_search(q[r0:r1], r[c0:c1], radius)
                    rows = np.repeat(np.arange(r0, r1), np.diff(lims.astype(np.int64))).astype(np.int32)
                    return torch.from_numpy(rows), torch.from_numpy((I + c0).astype(np.int32)), torch.from_numpy(D)
                a, b = max(r0, lo) - lo, min(r1, hi) - lo
                if b <= a:
                    return torch.zeros(0, dtype=torch.int32), torch.zeros(0, dtype=torch.int32), torch.zeros(0)
                got = prepared.get((a, b))
                if got is not None and got[0] <= radius:
                    stats["used"] += 1
                    m = got[3] > radius
                    return got[1][m], got[2][m], got[3][m]
                stats["redone" if got is not None else "demand"] += 1
                lims, D, I = orc.range_search(Q[a:b], r, radius)
                rows = np.repeat(np.arange(a, b), np.diff(lims.astype(np.int64))).astype(np.int32)
                return torch.from_numpy(rows), torch.from_numpy(I.astype(np.int32)), torch.from_numpy(D)

            def to_row_owners(i, j, sc):
                allh = vdist.all_gather_varlen(torch.stack([i.to(torch.int64), j.to(torch.int64),
                                                            sc.view(torch.int32).to(torch.int64)], dim=1))
                allh = allh[(allh[:, 0] >= lo) & (allh[:, 0] < hi)]
                return (allh[:, 0] - lo).to(torch.int32), allh[:, 1].to(torch.int32), allh[:, 2].to(torch.int32).view(torch.float32)

            trace = []
            head_end = min([a for a, _ in vdist.exponential_batches(len(q)) if a >= 96] + [len(q)])
            radius, hi_, hj_, hs_ = vdist.emulate_schedule(search_rows, len(q), K, trace=trace,
                                                           handover=(head_end, to_row_owners) if by_cols else None)
            if world == 2:
                # the same walk with the per-batch count all-reduce skipped wherever no event is possible (n_cols_total: the
                # engine's form of the call, round 6) and the phases accounted by a PhaseTimer: same radius, same hits
                timer = vdist.PhaseTimer(None)
                radius2, hi2, hj2, hs2 = vdist.emulate_schedule(search_rows, len(q), K, n_cols_total=len(r), timer=timer,
                                                                handover=(head_end, to_row_owners) if by_cols else None)
                assert np.float32(radius2) == np.float32(radius)
                o1 = np.lexsort((hj_.numpy(), hi_.numpy()))
                o2 = np.lexsort((hj2.numpy(), hi2.numpy()))
                assert np.array_equal(hi2.numpy()[o2], hi_.numpy()[o1]) and np.array_equal(hj2.numpy()[o2], hj_.numpy()[o1])
                assert np.array_equal(hs2.numpy()[o2].view(np.uint32), hs_.numpy()[o1].view(np.uint32))
                rep = timer.collect()
                n_batches = len(vdist.exponential_batches(len(q)))
                assert rep.get("count", {"calls": 0})["calls"] + timer.calls.get("count_skipped", 0) == n_batches
                res[f"k{K}"] = np.array([timer.calls.get("count_skipped", 0)])
            order = np.lexsort((hj_.numpy(), hi_.numpy(), -hs_.numpy().astype(np.float64)))
            hs_sorted = hs_[torch.from_numpy(order)]
            n_take, tau, info = vdist.distributed_prefix_select(hs_sorted, K, ties="rank", return_info=True)
            keep = order[:n_take]
            res[f"i{K}"], res[f"j{K}"], res[f"s{K}"] = hi_.numpy()[keep] + lo, hj_.numpy()[keep], hs_.numpy()[keep]
            res[f"r{K}"] = np.array([radius, sum(1 for t in trace if t[4])], dtype=np.float64)
            res[f"u{K}"] = np.array([stats["used"], stats["redone"], stats["demand"]])
        np.savez(f"{out_path}.{rank}.npz", **res)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,by_cols", [(2, False), (3, False), (2, True), (3, True)])
def test_emulated_schedule_over_query_shards_is_the_references(tmp_path, world, by_cols):
    """dist.emulate_schedule (round 5: what DeviceMatcher.match runs over query shards): the reference's batch schedule
    walked over lists the ranks hold -- prepared at floors below, or (deliberately) above, the schedule's radius -- must
    end on the reference's final radius and leave exactly the reference's hits, for K with no tie on the cut, with a tie
    the reference keeps and with a tie it drops.  by_cols: the batches at the head of the query set are searched under
    another partition (all of their rows against the rank's slice of the reference COLUMNS) and handed to the row owners
    before the first batch behind them, as engine.DeviceMatcher does with the doubling phase."""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
    import oracle as orc

    q, r = _tie_data()
    cls = _tie_classes(q, r)
    Ks = sorted(set(sum(cls.values(), [])))
    out = str(tmp_path / "em")
    mp.spawn(_emulate_worker, args=(world, 29300 + os.getpid() % 600, Ks, out, by_cols), nprocs=world, join=True)
    parts = [np.load(f"{out}.{k}.npz") for k in range(world)]
    used = redone = 0
    for K in Ks:
        i, j, s, info = orc.global_threshold_search(q, r, K, return_info=True)
        gi = np.concatenate([p[f"i{K}"] for p in parts])
        gj = np.concatenate([p[f"j{K}"] for p in parts])
        gs = np.concatenate([p[f"s{K}"] for p in parts])
        order = np.lexsort((gj, gi, -gs.astype(np.float64)))
        assert np.array_equal(gi[order], i) and np.array_equal(gj[order], j), K
        assert np.array_equal(gs[order].view(np.uint32), s.view(np.uint32)), K
        for p in parts:
            assert np.float32(p[f"r{K}"][0]) == np.float32(info["radius"]), (K, p[f"r{K}"], info)
            assert int(p[f"r{K}"][1]) == info["n_rethreshold"], (K, p[f"r{K}"], info)
        used += sum(int(p[f"u{K}"][0]) for p in parts)
        redone += sum(int(p[f"u{K}"][1]) for p in parts)
        if K in cls["dropped"]:
            assert len(s) < K
    assert used > 0 and redone > 0, (used, redone)   # both ways of answering a batch were exercised
    if world == 2:
        assert sum(int(p[f"k{K}"][0]) for p in parts for K in Ks) > 0   # (counts really were skipped somewhere)


def test_kth_best_unsorted_single_process():
    sys.path[:0] = [ROOT]
    from vsc2022_amd.dist import kth_best_unsorted

    rng = np.random.default_rng(8)
    for trial in range(200):
        n = int(rng.integers(0, 400))
        x = rng.normal(size=n).astype(np.float32)
        if trial % 2:
            x = np.round(x * 2) / 2
        if trial % 5 == 0 and n:
            x[rng.integers(0, n, 3)] = [0.0, -0.0, np.float32(1e-42)]
        k = n if trial % 7 == 3 and n else int(rng.integers(1, 420))
        tau, total = kth_best_unsorted(torch.from_numpy(x), k)
        assert total == n
        if k > n:
            assert tau == float("-inf")
        else:
            assert np.float32(tau) == np.sort(x)[::-1][k - 1], (trial, n, k)


class _OracleIndexRS(_OracleIndex):
    def range_scores(self, x, radius, k_hint, device_out=False):
        import oracle as orc

        return orc.range_search(np.ascontiguousarray(x, dtype=np.float32), self.rows, radius)[1]


def _refshard_tie_worker(rank, world, port, Ks, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
        from vsc2022_amd import dist as vdist
        from vsc2022_amd.refshard import RefShardedIndex

        q, r = _tie_data(seed=3, nq=150, nr=500)
        lo, hi = vdist.shard_ranges(len(r), world)[rank]
        idx = RefShardedIndex(_OracleIndexRS(r[lo:hi]), lo, len(r))
        res = {}
        for K in Ks:
            i, j, s, tau = idx.global_topk(q, K, k_local_start=K // 4 + 1)
            res[f"i{K}"], res[f"j{K}"], res[f"s{K}"] = i.numpy(), j.numpy(), s.numpy()
            res[f"f{K}"] = np.array([idx.last_select.tie_on_cut, idx.last_matches_reference, idx.last_ties_dropped])
        np.savez(f"{out_path}.{rank}.npz", **res)
    finally:
        dist.destroy_process_group()


def test_ref_sharded_topk_is_the_references_with_ties_on_the_cut(tmp_path):
    """Column shards, world 3: same statement as the query-sharded test; the reference's final radius comes from the
    schedule's batches run on all shards at once (dist.emulate_schedule_radius)."""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
    import oracle as orc

    q, r = _tie_data(seed=3, nq=150, nr=500)
    cls = _tie_classes(q, r)
    Ks = sorted(set(sum(cls.values(), [])))
    out = str(tmp_path / "rs")
    world = 3
    mp.spawn(_refshard_tie_worker, args=(world, 28100 + os.getpid() % 800, Ks, out), nprocs=world, join=True)
    for K in Ks:
        i, j, s = orc.global_threshold_search(q, r, K)
        for rank in range(world):
            got = np.load(f"{out}.{rank}.npz")
            assert np.array_equal(got[f"i{K}"], i) and np.array_equal(got[f"j{K}"], j), (rank, K)
            assert np.array_equal(got[f"s{K}"].view(np.uint32), s.view(np.uint32))
            flags = got[f"f{K}"]
            assert bool(flags[1])
            assert bool(flags[0]) == (K not in cls["notie"]) and bool(flags[2]) == (K in cls["dropped"]), (K, flags)


def test_emulated_schedule_radius_equals_the_oracles_single_process():
    """dist.emulate_schedule_radius in one process (no shards) against the final radius of the C oracle's schedule."""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
    import oracle as orc
    from vsc2022_amd import dist as vdist

    for seed in range(4):
        q, r = _tie_data(seed=seed, nq=220 + 40 * seed, nr=300)
        for K in (1, 7, 100, 1500, 9000, 40000, q.shape[0] * r.shape[0]):
            info = orc.global_threshold_search(q, r, K, return_info=True)[3]
            t = vdist.emulate_schedule_radius(
                lambda r0, r1, rad: torch.from_numpy(orc.range_search(q[r0:r1], r, rad)[1]), len(q), K)
            assert np.float32(t) == np.float32(info["radius"]), (seed, K, t, info)


def _owners_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        from vsc2022_amd import dist as vdist

        g = torch.Generator().manual_seed(100 + rank)
        n = [0, 1000, 37][rank % 3]                      # one rank has nothing to send
        rows = torch.randint(0, 1 << 20, (n, 3), generator=g, dtype=torch.int32)
        rows[:, 0] = rank                                 # (sender, payload, payload)
        dest = torch.randint(0, world, (n,), generator=g, dtype=torch.int64)
        if rank == 1:
            dest[dest == 2] = 0                           # ... and one link carries nothing
        got = vdist.send_to_owners(rows, dest, None)
        np.savez(os.path.join(out_dir, f"o{rank}.npz"), rows=rows.numpy(), dest=dest.numpy(), got=got.numpy())
        # rows of unequal length from every rank, rank order
        allv = vdist.all_gather_varlen(rows[:, 1].contiguous(), None)
        np.save(os.path.join(out_dir, f"g{rank}.npy"), allv.numpy())
    finally:
        dist.destroy_process_group()


def test_send_to_owners_and_varlen_gather_world3(tmp_path):
    """the two transfers of the column-sharded schedule (queries gathered once, kept hits to the ranks that own their rows):
    the same all-to-all / all-gather calls the RCCL run makes, here over gloo on CPU tensors"""
    world = 3
    mp.spawn(_owners_worker, args=(world, 29800 + os.getpid() % 150, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"o{r}.npz") for r in range(world)]
    for r in range(world):
        exp = np.concatenate([p["rows"][p["dest"] == r] for p in parts])
        got = parts[r]["got"]
        assert got.shape == exp.shape
        # any order between senders is fine; inside a sender the stable sort keeps the rows' order
        for sender in range(world):
            assert np.array_equal(got[got[:, 0] == sender], exp[exp[:, 0] == sender])
    allv = np.concatenate([p["rows"][:, 1] for p in parts])
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"g{r}.npy"), allv)


def _phase_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path[:0] = [ROOT]
        import json

        from vsc2022_amd import dist as vdist

        t = vdist.PhaseTimer(None)
        for _ in range(2 + rank):
            with t.phase("search"):
                pass
        if rank == 1:                       # a phase only ONE rank ever entered (row lists, a head without its rows)
            with t.phase("prepare_row_lists"):
                pass
            t.add_bytes("prepare_row_lists", 77)
        t.add_bytes("search", 10 * (rank + 1))
        rep = vdist.reduce_phase_report(t.collect(), torch.device("cpu"))
        with open(os.path.join(out_dir, f"phases{rank}.json"), "w") as f:
            json.dump(rep, f)
    finally:
        dist.destroy_process_group()


def test_phase_report_is_reduced_over_ranks_that_saw_different_phases(tmp_path):
    """`dist.reduce_phase_report` (bench.py's `sharded_search.phases_max_over_ranks`): max over ranks of every figure; a rank
    that never entered a phase must still take part in the collective with a tensor of the same shape."""
    import json

    mp.spawn(_phase_worker, args=(3, 29700 + os.getpid() % 200, str(tmp_path)), nprocs=3, join=True)
    reps = [json.load(open(tmp_path / f"phases{r}.json")) for r in range(3)]
    assert reps[0] == reps[1] == reps[2]
    assert set(reps[0]) == {"search", "prepare_row_lists"}
    assert reps[0]["search"]["calls"] == 4 and reps[0]["search"]["bytes"] == 30
    assert reps[0]["prepare_row_lists"]["calls"] == 1 and reps[0]["prepare_row_lists"]["bytes"] == 77
