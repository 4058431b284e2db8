"""GPU: the reference-sharded index (vsc2022_amd/refshard.py, BASELINE configs[4]) equals a single index bit
for bit.  The ranks share the one GPU of the test box, so the collectives run over gloo (staged through the
host); on a multi-GPU node the same code runs over RCCL.  Every shard runs the HIP kernels (shard-local
FlatIndex: vsc_index_knn / vsc_index_global_topk) on its own rows with its own row offset."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _data(seed):
    rng = np.random.default_rng(seed)
    nq, nr, d = {1: (300, 5000, 64), 2: (150, 70000, 128), 3: (64, 900, 32)}[seed]
    q = rng.standard_normal((nq, d)).astype(np.float32)
    r = rng.standard_normal((nr, d)).astype(np.float32)
    r[nr // 2 : nr // 2 + 40] = r[7:47]   # duplicate rows in different shards: exact ties across ranks
    q[3] = q[4]
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    r /= np.linalg.norm(r, axis=1, keepdims=True)
    return q, r


def _worker(rank, world, port, out_dir, seed, k, K):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        from vsc2022_amd import _lib
        from vsc2022_amd.refshard import RefShardedIndex

        q, r = _data(seed)
        idx = RefShardedIndex.build(r, r.shape[1], _lib.METRIC_INNER_PRODUCT, 0)
        D, I = idx.search(q, k)
        i, j, s, tau = idx.global_topk(q, K)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), D=D, I=I, i=i.cpu().numpy(), j=j.cpu().numpy(),
                 s=s.cpu().numpy(), row0=idx.row0, nloc=idx.local.ntotal)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("seed,world,k,K", [(1, 2, 5, 20000), (2, 3, 20, 60000), (3, 3, 1, 400)])
def test_ref_sharded_index_equals_single_index(gpu, tmp_path, seed, world, k, K):
    from vsc2022_amd import _lib
    from vsc2022_amd.vsc.index import FlatIndex

    q, r = _data(seed)
    single = FlatIndex(r.shape[1], _lib.METRIC_INNER_PRODUCT, 0)
    single.add(r)
    D, I = single.search(q, k)
    i, j, s, _ = single.global_topk(q, K)
    del single
    torch.cuda.empty_cache()
    mp.spawn(_worker, args=(world, 29750 + os.getpid() % 200, str(tmp_path), seed, k, K), nprocs=world, join=True)
    rows = 0
    for rank in range(world):
        got = np.load(tmp_path / f"rank{rank}.npz")
        assert int(got["row0"]) == rows
        rows += int(got["nloc"])
        assert np.array_equal(got["I"], I), rank
        assert np.array_equal(got["D"].view(np.uint32), D.view(np.uint32))
        assert np.array_equal(got["i"], i) and np.array_equal(got["j"], j.astype(np.int64)), rank
        assert np.array_equal(got["s"].view(np.uint32), s.view(np.uint32))
    assert rows == len(r)
