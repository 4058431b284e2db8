#!/usr/bin/env python3
"""Cost of the SHARDED code path itself (world_size 1, so no communication and the same work as the
single-process path): DeviceMatcher.match() through vsc2022_amd.dist vs the plain path, same inputs."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

from vsc2022_amd.engine import DeviceMatcher

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group(os.environ.get("BACKEND", "gloo"), rank=0, world_size=1)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(1)
n_qv, qf, n_rv, rf, dim = 8000, 25, 40000, 50, 512
refs = torch.randn((n_rv * rf, dim), generator=g, device=dev)
refs /= refs.norm(dim=1, keepdim=True)
q = torch.randn((n_qv * qf, dim), generator=g, device=dev)
q /= q.norm(dim=1, keepdim=True)
m = DeviceMatcher(refs, np.arange(n_rv + 1, dtype=np.int64) * rf, 0)
m.set_queries(q, np.arange(n_qv + 1, dtype=np.int64) * qf)
for forced in ("0", "1", "0", "1"):
    os.environ["VSC_FORCE_SHARDED"] = forced
    m.match(n_qvid_global=n_qv)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        res = m.match(n_qvid_global=n_qv)
    torch.cuda.synchronize()
    print(f"sharded path={forced}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms/step, hits {res.n_hits}, "
          f"candidates {res.n_candidates}, matches {res.n_matches}")
