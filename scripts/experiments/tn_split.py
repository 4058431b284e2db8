"""Where tn_pair_kernel's time goes on score-normalised descriptors (bias 0.5: nearly every top-k entry is a node):
the same 40 k pairs with max_path 10 (reference default), 3, 1 and 0, and with bias 0 (sparse graphs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bench import plant_copies, synth_on_device
from vsc2022_amd.engine import DeviceMatcher, DeviceScoreNormalizer

dev = torch.device("cuda", 0)
n_qv, qf, n_rv, rf, dim = 8000, 25, 40000, 50, 512
refs = synth_on_device(torch, dev, 1, n_rv, rf, dim)
q = synth_on_device(torch, dev, 2, n_qv, qf, dim)
plant_copies(torch, dev, 3, q, n_qv, qf, refs, n_rv, rf)
noise = synth_on_device(torch, dev, 77, 400000, 1, dim, static_frac=0.0)
norm = DeviceScoreNormalizer(noise, beta=1.2)
m = DeviceMatcher(norm.refs(refs), np.arange(n_rv + 1, dtype=np.int64) * rf, 0)
m.set_queries(norm.queries(q), np.arange(n_qv + 1, dtype=np.int64) * qf)
res = m.match(bias=0.5)
pq, pr = res.cand_q[:40000].contiguous(), res.cand_r[:40000].contiguous()
for bias, kw in ((0.5, {}), (0.5, dict(max_path=3)), (0.5, dict(max_path=1)), (0.5, dict(max_path=0)), (0.0, {})):
    args = dict(tn_max_step=5, min_length=4); args.update(kw)
    m.localize(pq, pr, bias, **args)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): nbox, boxes, bmax = m.localize(pq, pr, bias, **args)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"bias {bias} {kw}: {dt*1e3:.2f} ms per 40 k pairs = {dt/40000*1e9:.0f} ns/pair, boxes {int(nbox.sum())}")
# phase cycles (only with a -DVSC_TN_PROFILE build of tn.hip: VSCMI_LIB=build/libvscmi_tnprof.so)
import ctypes
from vsc2022_amd import _lib
L = _lib.lib()
if hasattr(L, "vsc_tn_prof_read"):
    buf = (ctypes.c_ulonglong * 8)()
    L.vsc_tn_prof_read(buf, 1)
    m.localize(pq, pr, 0.5, tn_max_step=5, min_length=4)
    torch.cuda.synchronize()
    L.vsc_tn_prof_read(buf, 1)
    names = ["sims tile", "top-k + ranges", "DP rows", "DP sink", "end node", "back-track (lane 0)", "MaxSim + box out"]
    tot = sum(buf[:7])
    for n, v in zip(names, buf[:7]):
        print(f"{n:22s} {v / 40000:10.0f} cycles/pair  {100.0 * v / tot:5.1f} %")
