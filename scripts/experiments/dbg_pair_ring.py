"""Repro of the nondeterministic misses of the paired int8 shape with a ring of 4 (fuzz seed 602)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
os.environ["VSC_I8"] = "2"; os.environ["VSC_I8P_PAIR"] = "2"
from vsc2022_amd.vsc.index import FlatIndex

def make(mode, d):
    os.environ["VSC_PREFILTER"] = mode
    return FlatIndex(d)

d, nq, nr, K = 512, 1834, 17772, 42789
rng = np.random.default_rng(5)
q = rng.standard_normal((nq, d)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
r = rng.standard_normal((nr, d)).astype(np.float32); r /= np.linalg.norm(r, axis=1, keepdims=True)
a, b = make("2", d), make("0", d)
a.add(r); b.add(r)
rb = b.global_topk(q, K)
ref = set(zip(rb[0].tolist(), rb[1].tolist()))
bad = 0
for t in range(int(os.environ.get("TRIALS", "40"))):
    ra = a.global_topk(q, K)
    got = set(zip(ra[0].tolist(), ra[1].tolist()))
    if got != ref or ra[3] != rb[3]:
        bad += 1
        miss = sorted(ref - got); extra = sorted(got - ref)
        print(f"trial {t}: radius {ra[3]} vs {rb[3]}; missing {len(miss)} extra {len(extra)}")
        print("   missing (i, j, i%256, i//16%16, j%256, j%16):", [(i, j, i % 256, (i // 16) % 16, j % 256, j % 16) for i, j in miss[:12]])
print("bad trials", bad)
