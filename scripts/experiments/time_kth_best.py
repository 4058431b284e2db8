"""How long the sharded events' order statistic takes on a WHOLE score matrix slice (the first event of a schedule: 1.28e8 scores of
all signs and magnitudes), single process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vsc2022_amd.dist import kth_best_unsorted, _RadixState
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
for n in (1_000_000, 16_000_000, 128_000_000):
    x = torch.randn(n, generator=g, device=dev) * 0.044
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tau, tot = kth_best_unsorted(x, 200001)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        st = _RadixState(200001, dev)
        per = []
        for shift in (24, 16, 8, 0):
            torch.cuda.synchronize(); a = time.perf_counter()
            h = st.hist(x, shift); torch.cuda.synchronize(); b = time.perf_counter()
            st.pick(h, shift); torch.cuda.synchronize()
            per.append(round(1e3 * (b - a), 2))
        m = x[x > tau]; torch.cuda.synchronize(); t2 = time.perf_counter()
        print(n, "kth", round(t1 - t0, 4), "s tau", tau, "levels ms", per, "kept", m.numel(), flush=True)
