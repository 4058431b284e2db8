"""Debug aid: which (row, ref) pairs does the forced-int8 route lose against the all-fp32 route?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vsc2022_amd.vsc.index import FlatIndex

def mk(d, **kv):
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update(kv)
    try:
        return FlatIndex(d)
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v

rng = np.random.default_rng(1)
nq, nr, d, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
q = rng.standard_normal((nq, d)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
r = rng.standard_normal((nr, d)).astype(np.float32); r /= np.linalg.norm(r, axis=1, keepdims=True)
a = mk(d, VSC_PREFILTER="2", VSC_I8="2"); a.add(r)
b = mk(d, VSC_PREFILTER="0"); b.add(r)
ia, ja, sa, _ = a.global_topk(q, K)
ib, jb, sb, _ = b.global_topk(q, K)
A = set(zip(ia.tolist(), ja.tolist())); B = set(zip(ib.tolist(), jb.tolist()))
miss = sorted(B - A); extra = sorted(A - B)
print("hits", len(A), len(B), "missing", len(miss), "extra", len(extra))
m = np.array(miss) if miss else np.zeros((0, 2), int)
if len(m):
    print("missing rows mod 16 hist", np.bincount(m[:, 0] % 16, minlength=16))
    print("missing rows //16 %8 hist", np.bincount((m[:, 0] // 16) % 8, minlength=8))
    print("missing cols mod 16 hist", np.bincount(m[:, 1] % 16, minlength=16))
    print("missing cols //16 %4 hist", np.bincount((m[:, 1] // 16) % 4, minlength=4))
    print("first", miss[:10])
pb = {(i, j) for i, j in B}
mb = np.array(sorted(B))
print("all rows mod 16 hist ", np.bincount(mb[:, 0] % 16, minlength=16))
