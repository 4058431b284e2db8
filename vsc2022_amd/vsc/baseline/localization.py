"""Temporal localisation of candidate pairs: mirror of the reference's
`vsc/baseline/localization.py` (same class names, arguments and hooks; paths relative to
/root/reference).

`VCSLLocalization` keeps the query and reference descriptors resident in HBM (a libvscmi TN
context) and, for the Temporal-Network model, runs similarity + alignment + box score fused on
the GPU -- one candidate pair per workgroup -- instead of computing each matrix on the host and
pickling it to a process pool.  Subclasses that override `score` with something the fused path
cannot know still work: they receive the (downloaded) similarity matrix exactly as in the
reference.
"""
import abc
import ctypes
from typing import List

import numpy as np

from vsc2022_amd import _lib
from vsc2022_amd.vsc.index import VideoFeature, VideoLayout
from vsc2022_amd.vsc.metrics import CandidatePair, Match


class Localization(abc.ABC):
    """vsc/baseline/localization.py:16-25"""

    @abc.abstractmethod
    def localize(self, candidate: CandidatePair) -> List[Match]:
        pass

    def localize_all(self, candidates: List[CandidatePair]) -> List[Match]:
        matches = []
        for candidate in candidates:
            matches.extend(self.localize(candidate))
        return matches


class LocalizationWithMetadata(Localization):
    """vsc/baseline/localization.py:28-36; descriptors live in HBM."""

    def __init__(self, queries: List[VideoFeature], refs: List[VideoFeature], device=None):
        self.queries = {m.video_id: m for m in queries}
        self.refs = {m.video_id: m for m in refs}
        # dict semantics of the reference: a repeated id keeps its last VideoFeature
        self._q_list = list(self.queries.values())
        self._r_list = list(self.refs.values())
        self._q_ord = {v.video_id: k for k, v in enumerate(self._q_list)}
        self._r_ord = {v.video_id: k for k, v in enumerate(self._r_list)}
        self.device = _lib.default_device() if device is None else int(device)
        ql, rl = VideoLayout(self._q_list), VideoLayout(self._r_list)
        qf, rf = VideoLayout.features(self._q_list), VideoLayout.features(self._r_list)
        dim = qf.shape[1] if qf.size else (rf.shape[1] if rf.size else 1)
        if qf.size and rf.size and qf.shape[1] != rf.shape[1]:
            raise ValueError("query and reference descriptors differ in dimension")
        self._ctx = ctypes.c_void_p()
        _lib.check(_lib.lib().vsc_tn_create(
            qf.ctypes.data if qf.size else None, ql.offsets.ctypes.data, len(self._q_list),
            rf.ctypes.data if rf.size else None, rl.offsets.ctypes.data, len(self._r_list),
            int(dim), _lib.MEM_HOST, self.device, ctypes.byref(self._ctx)))

    def __del__(self):
        ctx = getattr(self, "_ctx", None)
        if ctx is not None and ctx.value:
            try:
                _lib.lib().vsc_tn_destroy(ctx)
            except Exception:
                pass
            self._ctx = None

    def _pair_sims(self, candidate: CandidatePair, bias: float) -> np.ndarray:
        q = self._q_ord[candidate.query_id]
        r = self._r_ord[candidate.ref_id]
        lq, lr = len(self._q_list[q]), len(self._r_list[r])
        out = np.empty((lq, lr), dtype=np.float32)
        a, b = ctypes.c_int32(0), ctypes.c_int32(0)
        _lib.check(_lib.lib().vsc_tn_similarity(self._ctx, q, r, float(bias), out.ctypes.data,
                                                out.size, ctypes.byref(a), ctypes.byref(b)))
        return out

    def similarity(self, candidate: CandidatePair):
        return self._pair_sims(candidate, 0.0)


class VCSLLocalization(LocalizationWithMetadata):
    """vsc/baseline/localization.py:39-85"""

    def __init__(self, queries, refs, model_type, similarity_bias=0.0, device=None, **kwargs):
        super().__init__(queries, refs, device=device)
        # Late import, as in the reference (localization.py:43-46)
        from vsc2022_amd.vcsl.vta import build_vta_model

        self.model = build_vta_model(model_type, **kwargs)
        self.similarity_bias = similarity_bias

    def similarity(self, candidate: CandidatePair):
        """Add an optional similarity bias (localization.py:48-54)."""
        return self._pair_sims(candidate, self.similarity_bias)

    def _fused(self, candidates):
        n = len(candidates)
        pq = np.fromiter((self._q_ord[c.query_id] for c in candidates), dtype=np.int32, count=n)
        pr = np.fromiter((self._r_ord[c.ref_id] for c in candidates), dtype=np.int32, count=n)
        nbox = np.zeros(n, dtype=np.int32)
        boxes = np.zeros((n, _lib.TN_MAX_BOXES, 4), dtype=np.int32)
        bmax = np.zeros((n, _lib.TN_MAX_BOXES), dtype=np.float32)
        _lib.check(_lib.lib().vsc_tn_localize(
            self._ctx, pq.ctypes.data, pr.ctypes.data, n, _lib.MEM_HOST, ctypes.byref(self.model.params),
            float(self.similarity_bias), nbox.ctypes.data, boxes.ctypes.data, bmax.ctypes.data,
            _lib.MEM_HOST))
        return nbox, boxes, bmax

    def localize_all(self, candidates: List[CandidatePair]) -> List[Match]:
        candidates = list(candidates)
        if not candidates:
            return []
        from vsc2022_amd.vcsl.vta import TN

        known_score = type(self).score in (
            VCSLLocalization.score, VCSLLocalizationMaxSim.score, VCSLLocalizationCandidateScore.score)
        if type(self.model) is TN and known_score and type(self).similarity is VCSLLocalization.similarity:
            nbox, boxes, bmax = self._fused(candidates)
            sims = None
        else:  # generic route of the reference (localization.py:57-59)
            sims = [(f"{c.query_id}-{c.ref_id}", self.similarity(c)) for c in candidates]
            results = self.model.forward_sim(sims)
            assert len(results) == len(candidates)
        matches = []
        for k, candidate in enumerate(candidates):
            query: VideoFeature = self.queries[candidate.query_id]
            ref: VideoFeature = self.refs[candidate.ref_id]
            if sims is None:
                pair_boxes = boxes[k, : nbox[k]]
            else:
                assert sims[k][0] == results[k][0]
                pair_boxes = results[k][1]
            for b, box in enumerate(pair_boxes):
                (x1, y1, x2, y2) = (int(v) for v in box)
                match = Match(
                    query_id=candidate.query_id,
                    ref_id=candidate.ref_id,
                    query_start=query.get_timestamps(x1)[0],
                    query_end=query.get_timestamps(x2)[1],
                    ref_start=ref.get_timestamps(y1)[0],
                    ref_end=ref.get_timestamps(y2)[1],
                    score=0.0,
                )
                if sims is None:
                    score = self._fused_score(candidate, bmax[k, b])
                else:
                    score = self.score(candidate, match, (x1, y1, x2, y2), sims[k][1])
                matches.append(match._replace(score=score))
        return matches

    def _fused_score(self, candidate: CandidatePair, box_max) -> float:
        return 1.0

    def localize(self, candidate: CandidatePair) -> List[Match]:
        return self.localize_all([candidate])

    def score(self, candidate: CandidatePair, match: Match, box, similarity) -> float:
        return 1.0


class VCSLLocalizationMaxSim(VCSLLocalization):
    """vsc/baseline/localization.py:88-91 (half-open slice: the box's last row/column is excluded)."""

    def score(self, candidate: CandidatePair, match: Match, box, similarity) -> float:
        x1, y1, x2, y2 = box
        return similarity[x1:x2, y1:y2].max() - self.similarity_bias

    def _fused_score(self, candidate: CandidatePair, box_max) -> float:
        return box_max  # the kernel already returns max(sims[x1:x2, y1:y2]) - bias


class VCSLLocalizationCandidateScore(VCSLLocalization):
    """vsc/baseline/localization.py:94-96"""

    def score(self, candidate: CandidatePair, match: Match, box, similarity) -> float:
        return candidate.score

    def _fused_score(self, candidate: CandidatePair, box_max) -> float:
        return candidate.score
