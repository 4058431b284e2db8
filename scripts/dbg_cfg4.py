import faulthandler, sys, time
faulthandler.dump_traceback_later(120, repeat=True, file=sys.stderr)
sys.argv=['bench.py']; sys.path.insert(0, '.')
import numpy as np
import bench, torch
args=bench.parse()
dev=torch.device('cuda',0)
torch.cuda.set_device(dev)
from vsc2022_amd.engine import DeviceMatcher, DeviceScoreNormalizer
n_qv, qf, n_rv, rf, dim = int(sys.argv[1]) if len(sys.argv)>1 else 40000, 25, 40000, 50, 512
static = float(sys.argv[2]) if len(sys.argv)>2 else 0.01
refs = bench.synth_on_device(torch, dev, 301, n_rv, rf, dim, static_frac=static)
queries = bench.synth_on_device(torch, dev, 302, n_qv, qf, dim, static_frac=static)
bench.plant_copies(torch, dev, 303, queries, n_qv, qf, refs, n_rv, rf)
noise = bench.synth_on_device(torch, dev, 310, n_rv*rf, 1, dim, static_frac=0.0)
norm = DeviceScoreNormalizer(noise, beta=1.2)
del noise
qn = norm.queries(queries)
print("qn finite", bool(torch.isfinite(qn).all()), "bias min/max", qn[:,511].min().item(), qn[:,511].max().item(), "absmax", qn.abs().max().item(), flush=True)
rn = norm.refs(refs)
m = DeviceMatcher(rn, np.arange(n_rv + 1, dtype=np.int64) * rf, 0)
m.set_queries(qn, np.arange(n_qv + 1, dtype=np.int64) * qf)
m.index.profile(True)
for K in (1200*n_qv//40, 1200*n_qv):
    torch.cuda.synchronize(); t0=time.time()
    hi,hj,hs,rad = m.search(K)
    torch.cuda.synchronize()
    p = m.index.profile_read(reset=True)
    print("K", K, "search s", time.time()-t0, "hits", hs.numel(), "radius", rad, {k:round(v,1) for k,v in p.items() if 'ms' in k or 'launch' in k or k=='candidates'}, flush=True)
