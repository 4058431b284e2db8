"""Temporal localisation of candidate pairs on the MI355X engine.

Class surface of the reference module (`Localization`, `LocalizationWithMetadata`,
`VCSLLocalization`, `VCSLLocalizationMaxSim`, `VCSLLocalizationCandidateScore`, with their
`localize` / `localize_all` / `similarity` / `score` methods;
/root/reference/vsc/baseline/localization.py:16-96).

How it runs here: the constructor uploads both descriptor sets once into a libvscmi TN context
(HBM resident).  `localize_all` then sends only the (query ordinal, reference ordinal) list of the
batch; one workgroup per pair computes the frame x frame similarity on the matrix cores, runs the
Temporal-Network alignment and scores the boxes, and the host merely converts box corners to
timestamps.  A subclass that overrides `score` (or `similarity`) in a way the fused kernel cannot
know is still honoured: it falls back to the reference's route -- similarity matrices on the host,
`self.model.forward_sim(...)`, `self.score(...)` per box.
"""
import abc
import ctypes
from typing import Dict, List, Sequence

import numpy as np

from vsc2022_amd import _lib
from vsc2022_amd.vsc.index import VideoFeature, VideoLayout
from vsc2022_amd.vsc.metrics import CandidatePair, Match


class Localization(abc.ABC):
    @abc.abstractmethod
    def localize(self, candidate: CandidatePair) -> List[Match]:
        ...

    def localize_all(self, candidates: List[CandidatePair]) -> List[Match]:
        out: List[Match] = []
        for candidate in candidates:
            out += self.localize(candidate)
        return out


class LocalizationWithMetadata(Localization):
    """Keeps the descriptors of both sides (by video id) -- here: resident in HBM."""

    def __init__(self, queries: List[VideoFeature], refs: List[VideoFeature], device=None):
        # id -> VideoFeature; a repeated id keeps its last occurrence, as a dict does in the reference
        self.queries: Dict[object, VideoFeature] = {v.video_id: v for v in queries}
        self.refs: Dict[object, VideoFeature] = {v.video_id: v for v in refs}
        self._q_videos, self._r_videos = list(self.queries.values()), list(self.refs.values())
        self._q_ordinal = {v.video_id: n for n, v in enumerate(self._q_videos)}
        self._r_ordinal = {v.video_id: n for n, v in enumerate(self._r_videos)}
        self.device = _lib.default_device() if device is None else int(device)
        self._ctx = ctypes.c_void_p()
        self._upload()

    def _upload(self):
        q_layout, r_layout = VideoLayout(self._q_videos), VideoLayout(self._r_videos)
        q_rows, r_rows = VideoLayout.features(self._q_videos), VideoLayout.features(self._r_videos)
        if q_rows.size and r_rows.size and q_rows.shape[1] != r_rows.shape[1]:
            raise ValueError("query and reference descriptors differ in dimension")
        dim = q_rows.shape[1] if q_rows.size else (r_rows.shape[1] if r_rows.size else 1)
        _lib.check(_lib.lib().vsc_tn_create(
            q_rows.ctypes.data if q_rows.size else None, q_layout.offsets.ctypes.data, len(self._q_videos),
            r_rows.ctypes.data if r_rows.size else None, r_layout.offsets.ctypes.data, len(self._r_videos),
            int(dim), _lib.MEM_HOST, self.device, ctypes.byref(self._ctx)))

    def __del__(self):
        ctx = getattr(self, "_ctx", None)
        if ctx is not None and ctx.value:
            try:
                _lib.lib().vsc_tn_destroy(ctx)
            except Exception:  # interpreter shutdown
                pass
            self._ctx = None

    def _device_similarity(self, candidate: CandidatePair, bias: float) -> np.ndarray:
        q, r = self._q_ordinal[candidate.query_id], self._r_ordinal[candidate.ref_id]
        out = np.empty((len(self._q_videos[q]), len(self._r_videos[r])), dtype=np.float32)
        n_q, n_r = ctypes.c_int32(0), ctypes.c_int32(0)
        _lib.check(_lib.lib().vsc_tn_similarity(self._ctx, q, r, float(bias), out.ctypes.data, out.size,
                                                ctypes.byref(n_q), ctypes.byref(n_r)))
        return out

    def similarity(self, candidate: CandidatePair):
        """Frame x frame inner products of the pair (fp32, computed on the GPU)."""
        return self._device_similarity(candidate, 0.0)


class VCSLLocalization(LocalizationWithMetadata):
    """Alignment with a VCSL model (only "TN" exists here, as in every call of the reference)."""

    def __init__(self, queries, refs, model_type, similarity_bias=0.0, device=None, **kwargs):
        super().__init__(queries, refs, device=device)
        from vsc2022_amd.vcsl.vta import build_vta_model  # late import, as in the reference

        self.model = build_vta_model(model_type, **kwargs)
        self.similarity_bias = similarity_bias

    def similarity(self, candidate: CandidatePair):
        """Similarity plus the optional bias (some aligners dislike negative values)."""
        return self._device_similarity(candidate, self.similarity_bias)

    # -- the two ways of getting boxes ------------------------------------------------------
    def _can_fuse(self) -> bool:
        from vsc2022_amd.vcsl.vta import TN

        stock_scores = (VCSLLocalization.score, VCSLLocalizationMaxSim.score, VCSLLocalizationCandidateScore.score)
        return (type(self.model) is TN and type(self).score in stock_scores
                and type(self).similarity is VCSLLocalization.similarity)

    def _fused_boxes(self, candidates: Sequence[CandidatePair]):
        n = len(candidates)
        pair_q = np.fromiter((self._q_ordinal[c.query_id] for c in candidates), dtype=np.int32, count=n)
        pair_r = np.fromiter((self._r_ordinal[c.ref_id] for c in candidates), dtype=np.int32, count=n)
        n_boxes = np.zeros(n, dtype=np.int32)
        boxes = np.zeros((n, _lib.TN_MAX_BOXES, 4), dtype=np.int32)
        box_max = np.zeros((n, _lib.TN_MAX_BOXES), dtype=np.float32)
        _lib.check(_lib.lib().vsc_tn_localize(
            self._ctx, pair_q.ctypes.data, pair_r.ctypes.data, n, _lib.MEM_HOST, ctypes.byref(self.model.params),
            float(self.similarity_bias), n_boxes.ctypes.data, boxes.ctypes.data, box_max.ctypes.data, _lib.MEM_HOST))
        return n_boxes, boxes, box_max

    def _to_match(self, candidate: CandidatePair, box) -> Match:
        """Box corners are frame indices with inclusive ends: the segment runs from the start of the
        first frame to the end of the last one."""
        x1, y1, x2, y2 = box
        query, ref = self.queries[candidate.query_id], self.refs[candidate.ref_id]
        return Match(query_id=candidate.query_id, ref_id=candidate.ref_id, score=0.0,
                     query_start=query.get_timestamps(x1)[0], query_end=query.get_timestamps(x2)[1],
                     ref_start=ref.get_timestamps(y1)[0], ref_end=ref.get_timestamps(y2)[1])

    def localize_all(self, candidates: List[CandidatePair]) -> List[Match]:
        candidates = list(candidates)
        if not candidates:
            return []
        matches: List[Match] = []
        if self._can_fuse():
            n_boxes, boxes, box_max = self._fused_boxes(candidates)
            for k, candidate in enumerate(candidates):
                for b in range(int(n_boxes[k])):
                    match = self._to_match(candidate, [int(v) for v in boxes[k, b]])
                    matches.append(match._replace(score=self._fused_score(candidate, box_max[k, b])))
            return matches
        # the reference's route: matrices on the host, any model, any score hook
        named = [(f"{c.query_id}-{c.ref_id}", self.similarity(c)) for c in candidates]
        results = self.model.forward_sim(named)
        assert len(results) == len(candidates)
        for candidate, (key, sim), (name, pair_boxes) in zip(candidates, named, results):
            assert key == name
            for box in pair_boxes:
                box = tuple(int(v) for v in box)
                match = self._to_match(candidate, box)
                matches.append(match._replace(score=self.score(candidate, match, box, sim)))
        return matches

    def localize(self, candidate: CandidatePair) -> List[Match]:
        return self.localize_all([candidate])

    def score(self, candidate: CandidatePair, match: Match, box, similarity) -> float:
        return 1.0

    def _fused_score(self, candidate: CandidatePair, box_max) -> float:
        return 1.0


class VCSLLocalizationMaxSim(VCSLLocalization):
    """Box score = best frame similarity inside the box (half-open slice: the last row and column of
    the box are left out, exactly like the reference's `similarity[x1:x2, y1:y2]`), bias removed."""

    def score(self, candidate: CandidatePair, match: Match, box, similarity) -> float:
        x1, y1, x2, y2 = box
        return similarity[x1:x2, y1:y2].max() - self.similarity_bias

    def _fused_score(self, candidate: CandidatePair, box_max) -> float:
        return box_max  # computed by the kernel as max(sims[x1:x2, y1:y2]) - bias


class VCSLLocalizationCandidateScore(VCSLLocalization):
    """Box score = the retrieval score of the candidate pair."""

    def score(self, candidate: CandidatePair, match: Match, box, similarity) -> float:
        return candidate.score

    def _fused_score(self, candidate: CandidatePair, box_max) -> float:
        return candidate.score
