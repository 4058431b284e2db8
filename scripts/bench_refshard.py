#!/usr/bin/env python3
"""BASELINE configs[4]: reference rows sharded over the ranks, per-shard search on the GPU, merge over
torch.distributed (vsc2022_amd/refshard.py).

    torchrun --nproc-per-node 8 scripts/bench_refshard.py --refs-per-rank 2000000        # 8 GPUs, RCCL: 16M x 512
    python scripts/bench_refshard.py --spawn 2 --share-gpu --refs-per-rank 1000000       # one GPU, gloo (debugging)

Every rank generates ITS shard on the device (seeded by rank) and the same query rows; reports the k-NN and the
global top-K times and checks that all ranks hold the same merged result."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist


def run(rank, world, args):
    share = args.share_gpu
    dev = torch.device("cuda", 0 if share else int(os.environ.get("LOCAL_RANK", rank)))
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29931")
    if share:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from vsc2022_amd import _lib
    from vsc2022_amd.refshard import RefShardedIndex
    from vsc2022_amd.vsc.index import FlatIndex

    def unit(n, seed):
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        x = torch.randn((n, args.dim), generator=g, device=dev)
        return x / x.norm(dim=1, keepdim=True)

    shard = unit(args.refs_per_rank, 100 + rank)
    q = unit(args.queries, 7)
    local = FlatIndex(args.dim, _lib.METRIC_INNER_PRODUCT, dev.index)
    local.add(shard)
    del shard
    idx = RefShardedIndex(local, rank * args.refs_per_rank, world * args.refs_per_rank, None, dev)
    idx.search(q[:1024], args.k)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    D, I = idx.search(q, args.k)
    dist.barrier()
    t_knn = time.perf_counter() - t0
    t0 = time.perf_counter()
    i, j, s, tau = idx.global_topk(q, args.K)
    torch.cuda.synchronize()
    dist.barrier()
    t_top = time.perf_counter() - t0
    h = torch.tensor([int(I.sum()), int(D.view(np.uint32).astype(np.int64).sum()), int(j.sum()), int(i.to(torch.int64).sum())],
                     dtype=torch.int64)
    hs = [torch.zeros_like(h) for _ in range(world)]
    dist.all_gather(hs, h if share else h.to(dev))
    same = all(torch.equal(x.cpu(), hs[0].cpu()) for x in hs)
    if rank == 0:
        nr = world * args.refs_per_rank
        print(f"ref-sharded x{world} ({'gloo, shared GPU' if share else 'RCCL'}): {args.queries} queries x {nr} refs x {args.dim}-d | "
              f"k-NN k={args.k}: {t_knn * 1e3:.1f} ms ({args.queries / t_knn:.0f} rows/s) | global top-{args.K}: {t_top * 1e3:.1f} ms, "
              f"tau {tau:.6f}, {int(s.numel())} hits | all ranks hold the same result: {same}")
    assert same
    dist.destroy_process_group()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--refs-per-rank", type=int, default=2000000)
    ap.add_argument("--queries", type=int, default=32768)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--K", type=int, default=1000000)
    ap.add_argument("--spawn", type=int, default=0, help="spawn this many ranks from one process (else: torchrun env)")
    ap.add_argument("--share-gpu", action="store_true")
    args = ap.parse_args()
    if args.spawn:
        import torch.multiprocessing as mp

        mp.spawn(run, args=(args.spawn, args), nprocs=args.spawn, join=True)
    else:
        run(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), args)
