"""Debug aid: configs[3] at full size -- which planted (query video, ref video) pairs miss the candidate table, and why."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import plant_copies, synth_on_device
from vsc2022_amd.engine import DeviceMatcher, DeviceScoreNormalizer

dev = torch.device("cuda", 0)
n_qv, qf, n_rv, rf, dim = int(sys.argv[1]) if len(sys.argv) > 1 else 40000, 25, 40000, 50, 512
nr = n_rv * rf
refs = synth_on_device(torch, dev, 31, n_rv, rf, dim)
queries = synth_on_device(torch, dev, 32, n_qv, qf, dim)
gt = plant_copies(torch, dev, 33, queries, n_qv, qf, refs, n_rv, rf)
noise = synth_on_device(torch, dev, 34, nr, 1, dim, static_frac=0.0)
norm = DeviceScoreNormalizer(noise, beta=1.2)
del noise
qn = norm.queries(queries)
rn = norm.refs(refs)
m = DeviceMatcher(rn, np.arange(n_rv + 1, dtype=np.int64) * rf, 0)
m.set_queries(qn, np.arange(n_qv + 1, dtype=np.int64) * qf)
K = 1200 * n_qv
hi, hj, hs, radius = m.search(K)
print("hits", hs.numel(), "radius", radius, "top", float(hs[0]), "last", float(hs[-1]))
res = m.match(bias=0.5)
cq, cr, cs = res.cand_q.cpu().numpy(), res.cand_r.cpu().numpy(), res.cand_score.cpu().numpy()
print("candidates", len(cs), "score range", cs[0], cs[-1])
cand = set(zip(cq.tolist(), cr.tolist()))
miss = [p for p in gt if p not in cand]
print("planted", len(gt), "missing", len(miss))
hq = (hi // qf).cpu().numpy(); hr = (hj // rf).cpu().numpy(); hsn = hs.cpu().numpy()
pairkey = hq.astype(np.int64) * n_rv + hr
for p in miss[:12]:
    k = p[0] * n_rv + p[1]
    sel = pairkey == k
    a = qn[p[0] * qf:(p[0] + 1) * qf]; b = rn[p[1] * rf:(p[1] + 1) * rf]
    best = float((a @ b.T).max())
    print(p, "hits of the pair", int(sel.sum()), "best hit", hsn[sel].max() if sel.any() else None, "true best frame score", best,
          "bias col", float(a[:, -1].min()), float(a[:, -1].max()))
