import numpy as np, os, sys
sys.path.insert(0, os.getcwd())
from vsc2022_amd.vsc.index import FlatIndex
def unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32); return x / np.linalg.norm(x, axis=1, keepdims=True)
rng = np.random.default_rng(101)
q, r = unit(rng, 1000, 64), unit(rng, 1000, 64)
idx = FlatIndex(64); idx.set_option("topk_shortcut", 2); idx.set_option("topk_sample", 64); idx.set_option("debug_i8", 1)
idx.add(r)
i, j, s, rad = idx.global_topk(q, 60000)
print(len(s), rad, idx.get_option("last_topk_route"))
