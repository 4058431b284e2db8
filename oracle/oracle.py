"""ctypes front-end of the CPU oracle (oracle/libvscoracle.so).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under vsc2022_amd/ imports this module.

Every function here restates a piece of the reference hot path; see the header of
vsc_oracle.c for the file:line map into /root/reference.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvscoracle.so")

METRIC_INNER_PRODUCT = 0
METRIC_L2 = 1

_lib = None


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (no-op if it is already built and up to date)."""
    srcs = [os.path.join(_HERE, f) for f in ("vsc_oracle.c", "vsc_oracle_tn.c")]
    if (
        not force
        and os.path.exists(_LIB_PATH)
        and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)
    ):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libvscoracle.so"])
    return _LIB_PATH


def _usable_cpus() -> int:
    """CPUs this process may really use: the affinity mask, cut by the cgroup CPU quota (a pod on a large host sees
    every core in omp_get_max_threads(); a team that size on a small quota spends its time being throttled)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                quota = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0 and period > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, n)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        c_f = ctypes.POINTER(ctypes.c_float)
        c_i64 = ctypes.POINTER(ctypes.c_int64)
        c_i32 = ctypes.POINTER(ctypes.c_int32)
        i64 = ctypes.c_int64
        L.orc_scores.argtypes = [c_f, i64, c_f, i64, i64, ctypes.c_int, c_f]
        L.orc_scores.restype = None
        L.orc_range_search.argtypes = [c_f, i64, c_f, i64, i64, ctypes.c_int, ctypes.c_float,
                                       c_i64, c_f, c_i64, i64]
        L.orc_range_search.restype = i64
        L.orc_global_threshold_search.argtypes = [c_f, i64, c_f, i64, i64, ctypes.c_int, i64,
                                                  c_i64, c_i64, c_f, i64, c_i64, c_f, c_i64]
        L.orc_global_threshold_search.restype = ctypes.c_int
        L.orc_knn.argtypes = [c_f, i64, c_f, i64, i64, ctypes.c_int, i64, c_f, c_i64]
        L.orc_knn.restype = None
        L.orc_pair_max.argtypes = [c_i64, c_i64, c_f, i64, c_i32, c_i32, c_i32, c_i32, c_f, c_i64]
        L.orc_pair_max.restype = i64
        L.orc_row_normalize.argtypes = [c_f, i64, i64, c_f]
        L.orc_row_normalize.restype = None
        L.orc_num_threads.restype = ctypes.c_int
        L.orc_set_num_threads.argtypes = [ctypes.c_int]
        L.orc_set_num_threads(min(L.orc_num_threads(), _usable_cpus()))
        if hasattr(L, "orc_tn"):
            L.orc_tn.argtypes = [c_f, i64, i64, ctypes.POINTER(TNParams), c_i32, i64]
            L.orc_tn.restype = i64
        if hasattr(L, "orc_pair_sims"):
            L.orc_pair_sims.argtypes = [c_f, i64, c_f, i64, i64, ctypes.c_float, c_f]
            L.orc_pair_sims.restype = None
        _lib = L
    return _lib


class TNParams(ctypes.Structure):
    """Mirror of orc_tn_params (vsc_oracle_tn.c); VCSL `tn` keyword arguments."""

    _fields_ = [
        ("tn_max_step", ctypes.c_int32),
        ("tn_top_k", ctypes.c_int32),
        ("max_path", ctypes.c_int32),
        ("min_length", ctypes.c_int32),
        ("min_sim", ctypes.c_float),
        ("max_iou", ctypes.c_float),
    ]


def tn_params(tn_max_step=10, tn_top_k=5, max_path=10, min_sim=0.2, min_length=5, max_iou=0.3,
              **_ignored) -> TNParams:
    return TNParams(int(tn_max_step), int(tn_top_k), int(max_path), int(min_length),
                    float(min_sim), float(max_iou))


def _f32(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def scores(q, r, metric=METRIC_INNER_PRODUCT):
    """[nq, nr] fp32 similarity in the defined ascending-k fma-chain order."""
    q, r = _f32(q), _f32(r)
    assert q.ndim == 2 and r.ndim == 2 and q.shape[1] == r.shape[1]
    out = np.empty((q.shape[0], r.shape[0]), dtype=np.float32)
    lib().orc_scores(_p(q, ctypes.c_float), q.shape[0], _p(r, ctypes.c_float), r.shape[0],
                     q.shape[1], metric, _p(out, ctypes.c_float))
    return out


def range_search(q, r, radius, metric=METRIC_INNER_PRODUCT):
    """faiss IndexFlat.range_search: (lims uint64[nq+1], D float32, I int64)."""
    q, r = _f32(q), _f32(r)
    nq = q.shape[0]
    lims = np.zeros(nq + 1, dtype=np.int64)
    L = lib()
    n = L.orc_range_search(_p(q, ctypes.c_float), nq, _p(r, ctypes.c_float), r.shape[0], q.shape[1],
                           metric, float(np.float32(radius)), _p(lims, ctypes.c_int64), None, None, 0)
    if n < 0:
        raise MemoryError("orc_range_search failed")
    D = np.empty(n, dtype=np.float32)
    I = np.empty(n, dtype=np.int64)
    n2 = L.orc_range_search(_p(q, ctypes.c_float), nq, _p(r, ctypes.c_float), r.shape[0], q.shape[1],
                            metric, float(np.float32(radius)), _p(lims, ctypes.c_int64),
                            _p(D, ctypes.c_float), _p(I, ctypes.c_int64), n)
    assert n2 == n
    return lims.astype(np.uint64), D, I


def global_threshold_search(q, r, K, metric=METRIC_INNER_PRODUCT, return_info=False):
    """VideoIndex._global_threshold_knn_search: (i int64[n], j int64[n], s float32[n]), n <= K."""
    q, r = _f32(q), _f32(r)
    K = int(K)
    cap = max(K, 1)
    oi = np.empty(cap, dtype=np.int64)
    oj = np.empty(cap, dtype=np.int64)
    os_ = np.empty(cap, dtype=np.float32)
    n = ctypes.c_int64(0)
    rad = ctypes.c_float(0)
    nre = ctypes.c_int64(0)
    rc = lib().orc_global_threshold_search(
        _p(q, ctypes.c_float), q.shape[0], _p(r, ctypes.c_float), r.shape[0], q.shape[1], metric, K,
        _p(oi, ctypes.c_int64), _p(oj, ctypes.c_int64), _p(os_, ctypes.c_float), cap,
        ctypes.byref(n), ctypes.byref(rad), ctypes.byref(nre))
    if rc != 0:
        raise RuntimeError(f"orc_global_threshold_search rc={rc}")
    res = (oi[: n.value].copy(), oj[: n.value].copy(), os_[: n.value].copy())
    if return_info:
        return res + ({"radius": rad.value, "n_rethreshold": nre.value},)
    return res


def knn(q, r, k, metric=METRIC_INNER_PRODUCT):
    """faiss index.search(q, k): (D float32[nq,k], I int64[nq,k])."""
    q, r = _f32(q), _f32(r)
    D = np.empty((q.shape[0], k), dtype=np.float32)
    I = np.empty((q.shape[0], k), dtype=np.int64)
    lib().orc_knn(_p(q, ctypes.c_float), q.shape[0], _p(r, ctypes.c_float), r.shape[0], q.shape[1],
                  metric, k, _p(D, ctypes.c_float), _p(I, ctypes.c_int64))
    return D, I


def pair_max(hi, hj, hs, row2q, row2r):
    """index.py:121-140 regroup + candidates.py:24-40: (q_vid int32, r_vid int32, score, first_hit)."""
    hi = np.ascontiguousarray(hi, dtype=np.int64)
    hj = np.ascontiguousarray(hj, dtype=np.int64)
    hs = _f32(hs)
    row2q = np.ascontiguousarray(row2q, dtype=np.int32)
    row2r = np.ascontiguousarray(row2r, dtype=np.int32)
    n = hi.shape[0]
    oq = np.empty(max(n, 1), dtype=np.int32)
    orr = np.empty(max(n, 1), dtype=np.int32)
    os_ = np.empty(max(n, 1), dtype=np.float32)
    of = np.empty(max(n, 1), dtype=np.int64)
    m = lib().orc_pair_max(_p(hi, ctypes.c_int64), _p(hj, ctypes.c_int64), _p(hs, ctypes.c_float), n,
                           _p(row2q, ctypes.c_int32), _p(row2r, ctypes.c_int32),
                           _p(oq, ctypes.c_int32), _p(orr, ctypes.c_int32), _p(os_, ctypes.c_float),
                           _p(of, ctypes.c_int64))
    if m < 0:
        raise MemoryError("orc_pair_max failed")
    return oq[:m].copy(), orr[:m].copy(), os_[:m].copy(), of[:m].copy()


def row_normalize(x):
    x = _f32(x)
    out = np.empty_like(x)
    lib().orc_row_normalize(_p(x, ctypes.c_float), x.shape[0], x.shape[1], _p(out, ctypes.c_float))
    return out


def pair_sims(qf, rf, bias=0.0):
    """localization.py:36,52-54: q.feature @ r.feature.T (+ bias), fma-chain order, fp32."""
    qf, rf = _f32(qf), _f32(rf)
    out = np.empty((qf.shape[0], rf.shape[0]), dtype=np.float32)
    lib().orc_pair_sims(_p(qf, ctypes.c_float), qf.shape[0], _p(rf, ctypes.c_float), rf.shape[0],
                        qf.shape[1], float(bias), _p(out, ctypes.c_float))
    return out


MAX_BOXES = 16


def tn(sims, **kwargs):
    """vcsl.vta `tn` on one fp32 similarity matrix -> list of [q_min, r_min, q_max, r_max]."""
    sims = _f32(sims)
    prm = tn_params(**kwargs)
    boxes = np.zeros((MAX_BOXES, 4), dtype=np.int32)
    n = lib().orc_tn(_p(sims, ctypes.c_float), sims.shape[0], sims.shape[1], ctypes.byref(prm),
                     _p(boxes, ctypes.c_int32), MAX_BOXES)
    if n < 0:
        raise RuntimeError(f"orc_tn rc={n}")
    return [[int(v) for v in b] for b in boxes[:n]]


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))
