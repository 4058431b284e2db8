import sys, traceback
sys.path[:0] = ["/root/repo", "/root/repo/oracle"]
import numpy as np, oracle as orc
from vsc2022_amd.vsc.baseline.score_normalization import normalize
from vsc2022_amd.vsc.index import FlatIndex
rng = np.random.default_rng(2)
x = rng.standard_normal((777, 511)).astype(np.float32)
out = normalize(x); ref = orc.row_normalize(x)
d = out.view(np.uint32).astype(np.int64) - ref.view(np.uint32).astype(np.int64)
print("normalize: ndiff", (d != 0).sum(), "of", d.size, "max ulp", np.abs(d).max(), "rows with diff", (np.abs(d).max(axis=1) > 0).sum())
# is the norm different? recompute norm from ratio
r0 = np.nonzero(np.abs(d).max(axis=1) > 0)[0][:3]
for r in r0:
    print(r, np.nonzero(d[r])[0][:5], d[r][np.nonzero(d[r])[0][:5]])
try:
    q = rng.standard_normal((300, 512)).astype(np.float32); r = rng.standard_normal((5000, 512)).astype(np.float32)
    idx = FlatIndex(512); idx.add(r)
    D, I = idx.search(q, 20)
    oD, oI = orc.knn(q, r, 20)
    print("knn eq", np.array_equal(I, oI), np.array_equal(D.view(np.uint32), oD.view(np.uint32)))
except Exception:
    traceback.print_exc()
