import sys, time, os
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from vsc2022_amd import _lib
from vsc2022_amd.vsc.index import FlatIndex
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn((32768, 513), generator=g, device=dev); q /= q.norm(dim=1, keepdim=True)
for nr in (512, 65536, 250000):
    r = torch.randn((nr, 513), generator=g, device=dev); r /= r.norm(dim=1, keepdim=True)
    idx = FlatIndex(513, _lib.METRIC_INNER_PRODUCT, 0); idx.use_torch_stream(); idx.set_option("sort_hits", 0); idx.add(r)
    for rows in (32768, 4096):
        for _ in range(3): idx.global_topk(q[:rows], 1 << 22, device_out=True, seed_radius=0.9)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): idx.global_topk(q[:rows], 1 << 22, device_out=True, seed_radius=0.9)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        algo = 2.0 * rows * nr * 512 / 2.9e15
        print(f"nr={nr} rows={rows}: {dt*1e3:.3f} ms per seeded call with no hits (int8 kernel alone ~{algo*1e3:.3f} ms)")
