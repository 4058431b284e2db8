"""Minimal numpy stand-in for the 8 `faiss` symbols the reference touches.

ORACLE TOOLING ONLY (never imported by vsc2022_amd).  It exists so that the reference's own
Python (`/root/reference/vsc/*.py`) imports UNMODIFIED in the development container, where
faiss is not installed, in order to generate tests/golden/* (oracle/gen_golden.py) and to run
the reference's unit tests as a pin.  FAISS is a third-party dependency of the reference
(docs/installation.md:9-11, conda `faiss-gpu`, ~1.7.x) and is not vendored; the semantics
restated here are those used at vsc/index.py:11-13,79-82,145-154,169-174 and
vsc/baseline/score_normalization.py:10,88-89.

Scores come from oracle.scores (ascending-k fp32 fma chain) so that goldens are reproducible
bit for bit by the HIP path; FAISS itself (BLAS sgemm) pins no summation order.  `SCORE_MODE = "blas"`
switches the inner-product scores to `x @ xb.T` (numpy -> the host BLAS sgemm, i.e. what a real FAISS
flat index computes, in ITS summation order): oracle/eps_band.py runs the reference both ways and counts
what moves (hard part 2).
"""
import os
import sys

import numpy as np

_ORACLE_DIR = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ORACLE_DIR not in sys.path:
    sys.path.insert(0, _ORACLE_DIR)
import oracle as _orc  # noqa: E402

METRIC_INNER_PRODUCT = 0
METRIC_L2 = 1
SCORE_MODE = "fma"  # "fma": the oracle's ascending-k fp32 fma chain; "blas": numpy sgemm (inner product only)


class IndexFlat:
    def __init__(self, d, metric=METRIC_INNER_PRODUCT):
        self.d = int(d)
        self.metric_type = metric
        self._xb = np.zeros((0, self.d), dtype=np.float32)

    @property
    def ntotal(self):
        return self._xb.shape[0]

    def add(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        assert x.ndim == 2 and x.shape[1] == self.d
        self._xb = np.concatenate([self._xb, x], axis=0)

    def _scores(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        assert x.ndim == 2 and x.shape[1] == self.d
        if SCORE_MODE == "blas" and self.metric_type == METRIC_INNER_PRODUCT:
            return np.ascontiguousarray(x @ self._xb.T, dtype=np.float32)
        m = _orc.METRIC_INNER_PRODUCT if self.metric_type == METRIC_INNER_PRODUCT else _orc.METRIC_L2
        return _orc.scores(x, self._xb, m)

    def search(self, x, k):
        s = self._scores(x)
        nq, nb = s.shape
        D = np.full((nq, k), -np.finfo(np.float32).max if self.metric_type == METRIC_INNER_PRODUCT
                    else np.finfo(np.float32).max, dtype=np.float32)
        I = np.full((nq, k), -1, dtype=np.int64)
        key = -s if self.metric_type == METRIC_INNER_PRODUCT else s
        order = np.argsort(key, axis=1, kind="stable")[:, :k]
        kk = order.shape[1]
        D[:, :kk] = np.take_along_axis(s, order, axis=1)
        I[:, :kk] = order
        return D, I

    def range_search(self, x, radius):
        s = self._scores(x)
        radius = np.float32(radius)
        mask = (s > radius) if self.metric_type == METRIC_INNER_PRODUCT else (s < radius)
        lims = np.zeros(s.shape[0] + 1, dtype=np.uint64)
        lims[1:] = np.cumsum(mask.sum(axis=1))
        rows, cols = np.nonzero(mask)  # row-major: rows ascending, cols ascending within a row
        return lims, s[rows, cols].astype(np.float32), cols.astype(np.int64)


def index_factory(d, description="Flat", metric=METRIC_INNER_PRODUCT):
    if description != "Flat":
        raise NotImplementedError("oracle faiss shim only provides the flat index")
    return IndexFlat(d, metric)


def get_num_gpus():
    return 0


def index_cpu_to_all_gpus(index, co=None, ngpu=-1):
    return index
