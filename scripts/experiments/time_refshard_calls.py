import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from vsc2022_amd import _lib
from vsc2022_amd.vsc.index import FlatIndex
dev = torch.device("cuda", 0)
def unit(n, d, seed):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    x = torch.randn((n, d), generator=g, device=dev); return x / x.norm(dim=1, keepdim=True)
shard = unit(4_000_000, 512, 100); q = unit(4096, 512, 7)
idx = FlatIndex(512, _lib.METRIC_INNER_PRODUCT, 0); idx.add(shard)
def t(f, name):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); print(name, round(time.perf_counter() - t0, 3), flush=True); return r
for rep in range(2):
    t(lambda: idx.search(q, 20), "knn20")
    for K in (62501, 200001):
        r = t(lambda: idx.global_topk(q, K, device_out=True), f"topk {K}")
        print("  n", r[2].numel(), "radius", r[3])
    t(lambda: idx.range_scores(q[:2048], 0.2, 400000, device_out=True), "range_scores")
print("--- the first two batches of emulate_schedule_radius on this shard")
t(lambda: idx.range_scores(q[:32], -1e10, 400000, device_out=True), "range_scores batch 0-32 at -1e10")
t(lambda: idx.range_scores(q[32:96], 0.14785, 400000, device_out=True), "range_scores batch 32-96 at 0.14785")
t(lambda: idx.range_scores(q[32:96], 0.14785, 400000, device_out=True), "again")
t(lambda: idx.global_topk(q[32:96], 400000, device_out=True, seed_radius=0.14785), "seeded topk 400000")
print("--- emulate_schedule_radius on this shard alone (what a tie on the K cut costs a reference-sharded top-K), one process")
from vsc2022_amd import dist as vdist
marks = []
def rs(r0, r1, rad):
    torch.cuda.synchronize(); a = time.perf_counter()
    s = idx.range_scores(q[r0:r1], rad, 400000, device_out=True)
    torch.cuda.synchronize(); marks.append((r0, r1, rad, int(s.numel()), time.perf_counter() - a, time.perf_counter()))
    return s
torch.cuda.synchronize(); t0 = time.perf_counter()
rad = vdist.emulate_schedule_radius(rs, 4096, 200000, None, dev)
torch.cuda.synchronize(); print("emulate_schedule_radius", round(time.perf_counter() - t0, 3), "s, radius", rad)
prev = t0
for r0, r1, rr, n, dt, t_end in marks:
    print(f"  batch {r0}-{r1} radius {rr:.5f}: {n} scores, search {dt:.3f} s, since previous batch ended {t_end - dt - prev:.3f} s")
    prev = t_end
