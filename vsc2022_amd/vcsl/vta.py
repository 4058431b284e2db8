"""`vcsl.vta` surface used by the reference (vsc/baseline/localization.py:44-46,58;
tests/test_localization.py:17): `build_vta_model(model_type, **kwargs)` returning an object with
`forward_sim([(name, sims), ...]) -> [(name, [[q_lo, r_lo, q_hi, r_hi], ...]), ...]`.

"TN" -- the only model the reference ever requests (sscd_baseline.py:121,131; dns_baseline.py:202) -- runs on the GPU
(libvscmi vsc_tn_forward_sim, one candidate pair per workgroup).  "DTW", "DP" and "HV" (vsc2022_amd/vcsl/aligners.py) are
host code over the similarity matrices, as VCSL's own are: the reference's route, off the hot path.  "SPD" is a trained
detector network whose weights ship with VCSL only: not buildable here (aligners.py, module docstring).  The third-party
VCSL source is not part of the reference checkout; see DESIGN.md ("TN: parity unpinned" -- the same holds for these).
"""
import ctypes
from typing import List, Optional, Sequence, Tuple

import numpy as np

from vsc2022_amd import _lib

# VCSL `tn` defaults
TN_DEFAULTS = dict(tn_max_step=10, tn_top_k=5, max_path=10, min_sim=0.2, min_length=5, max_iou=0.3)


def tn_params(**config) -> _lib.TNParams:
    unknown = set(config) - set(TN_DEFAULTS)
    if unknown:
        raise TypeError(f"unexpected TN arguments: {sorted(unknown)}")
    cfg = dict(TN_DEFAULTS, **config)
    return _lib.TNParams(int(cfg["tn_max_step"]), int(cfg["tn_top_k"]), int(cfg["max_path"]),
                         int(cfg["min_length"]), float(cfg["min_sim"]), float(cfg["max_iou"]))


def unpack_boxes(nbox: np.ndarray, boxes: np.ndarray) -> List[List[List[int]]]:
    return [[[int(v) for v in boxes[p, b]] for b in range(int(nbox[p]))] for p in range(len(nbox))]


class TN:
    """Temporal Network aligner.  `concurrency` (a multiprocessing pool size in VCSL) is accepted
    and ignored: every pair of a call is aligned concurrently on the GPU."""

    def __init__(self, concurrency: int = 1, device: Optional[int] = None, **config):
        self.config = dict(config)
        self.params = tn_params(**config)
        self.device = device

    def forward_sim(self, data: Sequence[Tuple[str, np.ndarray]]) -> List[Tuple[str, List[List[int]]]]:
        names = [name for name, _ in data]
        mats = [_lib.f32c(sim) for _, sim in data]
        n = len(mats)
        if n == 0:
            return []
        for m in mats:
            if m.ndim != 2:
                raise ValueError("forward_sim expects 2-D similarity matrices")
        lq = np.array([m.shape[0] for m in mats], dtype=np.int32)
        lr = np.array([m.shape[1] for m in mats], dtype=np.int32)
        off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lq.astype(np.int64) * lr, out=off[1:])
        flat = np.concatenate([m.reshape(-1) for m in mats]) if off[-1] else np.zeros(1, np.float32)
        nbox = np.zeros(n, dtype=np.int32)
        boxes = np.zeros((n, _lib.TN_MAX_BOXES, 4), dtype=np.int32)
        bmax = np.zeros((n, _lib.TN_MAX_BOXES), dtype=np.float32)
        device = _lib.default_device() if self.device is None else self.device
        _lib.check(_lib.lib().vsc_tn_forward_sim(
            flat.ctypes.data, off.ctypes.data, lq.ctypes.data, lr.ctypes.data, n,
            ctypes.byref(self.params), nbox.ctypes.data, boxes.ctypes.data, bmax.ctypes.data, device))
        return list(zip(names, unpack_boxes(nbox, boxes)))


_REGISTRY = {}


def register_vta_model(name: str, factory):
    """Plug another aligner in behind `build_vta_model(name, **kwargs)`: `factory(concurrency=..., **kwargs)` must
    return an object with `forward_sim([(name, sims), ...]) -> [(name, [[q_lo, r_lo, q_hi, r_hi], ...]), ...]`
    (VCSL's DTW / DP / HV / SPD have that shape).  Such models run on the reference's route of
    `VCSLLocalization.localize_all`: similarity matrices from the GPU, alignment by the model, `score()` per box."""
    _REGISTRY[str(name)] = factory


def build_vta_model(method="TN", concurrency: int = 1, **config):
    if not isinstance(method, str):
        if not hasattr(method, "forward_sim"):
            raise TypeError("an alignment model object must provide forward_sim(data)")
        return method  # a ready-made aligner
    if method == "TN":
        return TN(concurrency=concurrency, **config)
    if method in ("DTW", "DP", "HV") and method not in _REGISTRY:
        from vsc2022_amd.vcsl import aligners

        return getattr(aligners, method)(concurrency=concurrency, **config)
    if method in _REGISTRY:
        return _REGISTRY[method](concurrency=concurrency, **config)
    raise NotImplementedError(
        f"alignment model {method!r}: 'TN' (GPU), 'DTW', 'DP' and 'HV' (host) ship with this package; VCSL's 'SPD' is a "
        "TRAINED detector network over the similarity matrix -- its weights ship with VCSL, which is not part of the "
        "reference checkout, and the reference never requests it; register a model that has them with "
        "vsc2022_amd.vcsl.vta.register_vta_model"
    )
