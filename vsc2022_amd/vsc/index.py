"""Video-level flat index on MI355X: host-side mirror of the reference's `vsc/index.py`.

Same public names, arguments and error behaviour (cited per symbol; paths relative to
/root/reference), but the data model is flat arrays + offset tables instead of per-row Python
lists, the reference set lives in HBM, and every numeric step is a call through the C ABI of
libvscmi.so (include/vscmi.h).  There is no CPU fallback.
"""
import logging
from dataclasses import dataclass
from typing import Iterable, Iterator, List, NamedTuple, Optional, Sequence, Tuple

import ctypes
import numpy as np

from vsc2022_amd import _lib

# numeric values of faiss.METRIC_INNER_PRODUCT / faiss.METRIC_L2 (vsc/index.py:78; tests/test_index.py:43)
METRIC_INNER_PRODUCT = _lib.METRIC_INNER_PRODUCT
METRIC_L2 = _lib.METRIC_L2

SearchIndices = Tuple[int, int, float]


@dataclass
class VideoMetadata:
    """vsc/index.py:18-30"""

    video_id: str
    timestamps: np.ndarray  # either Nx2 (start and end timestamps) or N

    def __len__(self):
        return self.timestamps.shape[0]

    def get_timestamps(self, idx: int) -> Tuple[float, float]:
        t = self.timestamps[idx]
        if len(self.timestamps.shape) == 1:
            return (t, t)
        return (t[0], t[1])


@dataclass
class VideoFeature(VideoMetadata):
    """vsc/index.py:33-46"""

    feature: np.ndarray

    def __post_init__(self):
        assert self.feature.shape[0] == len(self.timestamps), "Mismatched timestamps / feature size"

    def metadata(self):
        return VideoMetadata(video_id=self.video_id, timestamps=self.timestamps)

    def dimensions(self):
        return self.feature.shape[1]


class PairMatch(NamedTuple):
    """vsc/index.py:49-52"""

    query_timestamps: Tuple[float, float]
    ref_timestamps: Tuple[float, float]
    score: float


class _MatchView(Sequence):
    """The frame-level hits of one (query video, ref video) pair, materialised on access."""

    __slots__ = ("_q_meta", "_r_meta", "_q_idx", "_r_idx", "_scores")

    def __init__(self, q_meta, r_meta, q_idx, r_idx, scores):
        self._q_meta, self._r_meta = q_meta, r_meta
        self._q_idx, self._r_idx, self._scores = q_idx, r_idx, scores

    def __len__(self):
        return len(self._scores)

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[x] for x in range(*k.indices(len(self)))]
        return PairMatch(
            query_timestamps=self._q_meta.get_timestamps(self._q_idx[k]),
            ref_timestamps=self._r_meta.get_timestamps(self._r_idx[k]),
            score=self._scores[k],
        )

    def __eq__(self, other):
        return list(self) == list(other)

    def __repr__(self):
        return repr(list(self))

    @property
    def scores(self) -> np.ndarray:
        return self._scores


@dataclass
class PairMatches:
    """vsc/index.py:55-71"""

    query_id: str
    ref_id: str
    matches: List[PairMatch]

    def records(self):
        for match in self.matches:
            yield {
                "query_id": self.query_id,
                "ref_id": self.ref_id,
                "query_start": match.query_timestamps[0],
                "query_end": match.query_timestamps[1],
                "ref_start": match.ref_timestamps[0],
                "ref_end": match.ref_timestamps[1],
                "score": match.score,
            }


class FlatIndex:
    """HBM-resident flat index that quacks like the faiss object the reference reaches into
    (`VideoIndex.index`; vsc/baseline/score_normalization.py:87-96 calls `.search` on it):
    `.d`, `.metric_type`, `.ntotal`, `.add(x)`, `.search(x, k) -> (D, I)`,
    `.range_search(x, radius) -> (lims, D, I)`.
    """

    def __init__(self, d: int, metric: int = METRIC_INNER_PRODUCT, device: Optional[int] = None,
                 options: Optional[dict] = None):
        """options: {name: value} applied with `set_option` while the index is still empty -- the explicit form of the
        VSC_* environment switches (include/vscmi.h), including the ones that decide which images of the reference
        rows are kept ("prefilter", "i8", "f16_kernel", "i8_exclude") and can only be set before the first `add`."""
        if metric not in (METRIC_INNER_PRODUCT, METRIC_L2):
            raise ValueError(f"unsupported metric {metric}")
        self.d = int(d)
        self.metric_type = metric
        self.device = _lib.default_device() if device is None else int(device)
        self._h = ctypes.c_void_p()
        self._stream = None  # None: the handle's own stream; else the hipStream_t value it was bound to (0 = default stream)
        _lib.check(_lib.lib().vsc_index_create(self.d, metric, self.device, ctypes.byref(self._h)))
        for name, value in (options or {}).items():
            self.set_option(name, value)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                _lib.lib().vsc_index_destroy(h)
            except Exception:  # interpreter shutdown
                pass
            self._h = None

    @property
    def handle(self):
        return self._h

    @property
    def ntotal(self) -> int:
        return int(_lib.lib().vsc_index_ntotal(self._h))

    # ---- options and stream (include/vscmi.h: vsc_index_set_option / _get_option / _set_stream)
    def set_option(self, name: str, value: float):
        """Programmatic form of the VSC_* switches (name = the variable's without the prefix, lower case), e.g.
        `index.set_option("i8_density", 3e-4)`; ValueError for unknown names / values / options fixed once rows exist."""
        _lib.check(_lib.lib().vsc_index_set_option(self._h, name.encode(), float(value)))

    def get_option(self, name: str) -> float:
        v = ctypes.c_double(0.0)
        _lib.check(_lib.lib().vsc_index_get_option(self._h, name.encode(), ctypes.byref(v)))
        return v.value

    def use_stream(self, hip_stream: Optional[int]):
        """Run this handle's work on the caller's HIP stream (an integer hipStream_t, e.g.
        `torch.cuda.current_stream().cuda_stream`; 0 = the default stream); None: back to the handle's own stream."""
        if hip_stream is None:
            _lib.check(_lib.lib().vsc_index_set_stream(self._h, None, 1))
        else:
            _lib.check(_lib.lib().vsc_index_set_stream(self._h, ctypes.c_void_p(int(hip_stream)), 0))
        self._stream = None if hip_stream is None else int(hip_stream)

    def use_torch_stream(self):
        """Bind the handle to torch's current stream on its device: tensors torch produced on that stream are then
        ordered before the library's reads without a device-wide synchronisation."""
        import torch

        self.use_stream(torch.cuda.current_stream(torch.device("cuda", self.device)).cuda_stream)

    def _after_torch(self, dev=None):
        """Make torch's queued work visible to the handle's stream: nothing to do when the handle runs ON torch's current
        stream, a device synchronisation otherwise (the handle's own stream is non-blocking)."""
        import torch

        dev = torch.device("cuda", self.device) if dev is None else dev
        if self._stream is not None and torch.cuda.current_stream(dev).cuda_stream == self._stream:
            return
        torch.cuda.synchronize(dev)

    def _rows(self, x):
        """fp32 C-contiguous [n, d] as (pointer, mem kind, n, keep-alive)."""
        if isinstance(x, np.ndarray) or not hasattr(x, "data_ptr"):
            x = _lib.f32c(x)
        else:  # torch tensor (host or HBM)
            import torch

            x = x.to(torch.float32).contiguous()
            if x.is_cuda:
                # whatever torch kernel produced (or converted) these rows must be ordered before the library's reads
                self._after_torch(x.device)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise ValueError(f"expected [n, {self.d}] features, got {tuple(x.shape)}")
        p, mem = _lib.ptr(x)
        return p, mem, int(x.shape[0]), x

    def add(self, x):
        p, mem, n, keep = self._rows(x)
        _lib.check(_lib.lib().vsc_index_add(self._h, p, n, mem))

    def search(self, x, k: int, device_out: bool = False):
        """faiss index.search: (D float32 [n, k], I int64 [n, k]); with device_out torch tensors that never leave
        the HBM (score normalisation and the reference-sharded merge consume them there)."""
        p, mem, n, keep = self._rows(x)
        if device_out:
            import torch

            dev = torch.device("cuda", self.device)
            D = torch.empty((n, k), dtype=torch.float32, device=dev)
            I = torch.empty((n, k), dtype=torch.int64, device=dev)
            self._after_torch(dev)
            _lib.check(_lib.lib().vsc_index_knn(self._h, p, n, mem, int(k), D.data_ptr(), I.data_ptr(),
                                                _lib.MEM_DEVICE))
            return D, I
        D = np.empty((n, k), dtype=np.float32)
        I = np.empty((n, k), dtype=np.int64)
        _lib.check(_lib.lib().vsc_index_knn(self._h, p, n, mem, int(k), D.ctypes.data, I.ctypes.data,
                                            _lib.MEM_HOST))
        return D, I

    def range_search(self, x, radius: float):
        """faiss index.range_search: (lims uint64 [n+1], D float32, I int64), strict comparison."""
        p, mem, n, keep = self._rows(x)
        lims = np.zeros(n + 1, dtype=np.int64)
        need = ctypes.c_int64(0)
        L = _lib.lib()
        _lib.check(L.vsc_index_range_search(self._h, p, n, mem, float(radius), lims.ctypes.data, None,
                                            None, 0, ctypes.byref(need)))
        D = np.empty(need.value, dtype=np.float32)
        I = np.empty(need.value, dtype=np.int64)
        if need.value:
            _lib.check(L.vsc_index_range_search(self._h, p, n, mem, float(radius), lims.ctypes.data,
                                                D.ctypes.data, I.ctypes.data, need.value,
                                                ctypes.byref(need)))
        return lims.astype(np.uint64), D, I

    def global_topk(self, x, K: int, device_out: bool = False, seed_radius=None):
        """The adaptive global-threshold search of vsc/index.py:142-165 in one call.

        Returns (i int32, j int32, s float32, radius): at most K hits ordered by
        (score desc, row asc, ref asc).  With device_out the arrays are torch tensors in HBM.

        seed_radius (extension, `vsc_index_global_topk_seeded`): a radius the caller knows to lie below the K-th best
        score -- the rows then run as steady batches from it instead of replaying the reference's doubling schedule
        from -/+1e10.  Every hit strictly beyond the returned radius is listed (cut at K); fewer than K hits come back
        when the seed was too high.  The query-sharded pipeline seeds its ranks' searches this way (engine.py).
        """
        p, mem, n, keep = self._rows(x)
        K = int(K)
        n_out = ctypes.c_int64(0)
        radius = ctypes.c_float(0.0)
        cap = max(min(K, n * max(self.ntotal, 1)), 1)

        def call(pi, pj, ps, out_mem):
            if seed_radius is None:
                _lib.check(_lib.lib().vsc_index_global_topk(self._h, p, n, mem, K, pi, pj, ps, cap, out_mem,
                                                            ctypes.byref(n_out), ctypes.byref(radius)))
            else:
                _lib.check(_lib.lib().vsc_index_global_topk_seeded(self._h, p, n, mem, K, float(seed_radius), pi, pj, ps, cap,
                                                                   out_mem, ctypes.byref(n_out), ctypes.byref(radius)))

        if device_out:
            import torch

            dev = torch.device("cuda", self.device)
            oi = torch.empty(cap, dtype=torch.int32, device=dev)
            oj = torch.empty(cap, dtype=torch.int32, device=dev)
            os_ = torch.empty(cap, dtype=torch.float32, device=dev)
            self._after_torch(dev)
            call(oi.data_ptr(), oj.data_ptr(), os_.data_ptr(), _lib.MEM_DEVICE)
            m = n_out.value
            return oi[:m], oj[:m], os_[:m], radius.value
        oi = np.empty(cap, dtype=np.int32)
        oj = np.empty(cap, dtype=np.int32)
        os_ = np.empty(cap, dtype=np.float32)
        call(oi.ctypes.data, oj.ctypes.data, os_.ctypes.data, _lib.MEM_HOST)
        m = n_out.value
        return oi[:m], oj[:m], os_[:m], radius.value

    def range_scores(self, x, radius: float, k_hint: int, device_out: bool = False):
        """The scores STRICTLY beyond `radius` of x against the index (inner product), no ids: what one batch of
        range_search_max_results keeps (vsc/index.py:147-154) when the caller drives the schedule itself
        (dist.emulate_schedule_radius over shards).  Runs `vsc_index_global_topk_seeded` with a budget that is raised
        until the list is provably complete: its radius never moved and it is shorter than the budget."""
        n = int(x.shape[0])
        total = n * max(self.ntotal, 1)
        kk = max(int(k_hint), 1024)
        while True:
            k_try = min(kk, total)
            _, _, s, rad = self.global_topk(x, k_try, device_out=device_out, seed_radius=float(radius))
            if k_try >= total or (np.float32(rad) == np.float32(radius) and len(s) < k_try):
                return s
            kk *= 4

    def set_hit_capacity(self, cap: int):
        _lib.check(_lib.lib().vsc_index_set_hit_capacity(self._h, int(cap)))

    def profile(self, enable: bool):
        _lib.check(_lib.lib().vsc_index_profile(self._h, 1 if enable else 0))

    def profile_read(self, reset: bool = True):
        ms, n, fl = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_double(0)
        _lib.check(_lib.lib().vsc_index_profile_read(self._h, ctypes.byref(ms), ctypes.byref(n),
                                                     ctypes.byref(fl), 1 if reset else 0))
        out = {"sim_ms": ms.value, "sim_launches": n.value, "sim_flops": fl.value}
        # per kernel class: fp16 pre-filter GEMM and exact re-scoring of its candidates
        # (5: the int8 pre-filter kernel of the sparse batches, csrc/sim_i8p.hip; 6: its launches' preamble -- row
        # sorts and the quantisation of the query panels)
        for cls, name in ((1, "f16"), (2, "rescore"), (3, "select"), (4, "sort"), (5, "i8"), (6, "i8_prep")):
            _lib.check(_lib.lib().vsc_index_profile_read_class(self._h, cls, ctypes.byref(ms), ctypes.byref(n),
                                                               ctypes.byref(fl), 1 if reset else 0))
            out.update({f"{name}_ms": ms.value, f"{name}_launches": n.value, f"{name}_flops": fl.value})
        cand, hits = ctypes.c_int64(0), ctypes.c_int64(0)
        _lib.check(_lib.lib().vsc_index_search_stats(self._h, ctypes.byref(cand), ctypes.byref(hits)))
        out["candidates"] = cand.value
        return out


class _RowToVideoId(Sequence):
    """`VideoIndex.video_clip_to_video_ids` (vsc/index.py:84,90-92) without one Python object per row."""

    def __init__(self, row2vid: np.ndarray, video_ids: list):
        self._row2vid, self._ids = row2vid, video_ids

    def __len__(self):
        return len(self._row2vid)

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self._ids[v] for v in self._row2vid[k]]
        return self._ids[self._row2vid[k]]


class VideoLayout:
    """Flat description of a list of VideoFeature: CSR-like row tables."""

    def __init__(self, videos: Sequence[VideoFeature]):
        self.video_ids = [v.video_id for v in videos]
        lens = np.fromiter((len(v) for v in videos), dtype=np.int64, count=len(videos))
        self.offsets = np.zeros(len(videos) + 1, dtype=np.int64)
        np.cumsum(lens, out=self.offsets[1:])
        self.n_rows = int(self.offsets[-1])
        self.row2vid = np.repeat(np.arange(len(videos), dtype=np.int32), lens)
        self.row2frame = (np.arange(self.n_rows, dtype=np.int64) - np.repeat(self.offsets[:-1], lens)).astype(np.int32)
        self.metadata = [v.metadata() for v in videos]

    @staticmethod
    def features(videos: Sequence[VideoFeature]) -> np.ndarray:
        if len(videos) == 0:
            return np.zeros((0, 0), dtype=np.float32)
        return np.concatenate([_lib.f32c(v.feature) for v in videos], axis=0)


class SearchHits:
    """Frame-level hits of one search in array form (the payload of VideoIndex.search)."""

    def __init__(self, i, j, s, q_layout: VideoLayout, index: "VideoIndex"):
        self.i, self.j, self.s = i, j, s
        self.q_layout, self.index = q_layout, index

    def __len__(self):
        return len(self.s)


class VideoIndex:
    """vsc/index.py:74-177"""

    def __init__(self, dim: int, codec_str: str = "Flat", metric: int = METRIC_INNER_PRODUCT,
                 device: Optional[int] = None):
        if codec_str != "Flat":
            raise NotImplementedError(
                f"codec {codec_str!r}: the MI355X engine implements the flat (exhaustive) index only"
            )
        self.dim = dim
        self.index = FlatIndex(dim, metric, device)
        self.video_metadata = {}
        self._video_ids: list = []
        self._vid_ordinal = {}
        self._row2vid = np.zeros(0, dtype=np.int32)
        self._row2frame = np.zeros(0, dtype=np.int32)

    # -- attributes of the reference object (vsc/index.py:83-85)
    @property
    def video_clip_idx(self) -> np.ndarray:
        return self._row2frame

    @property
    def video_clip_to_video_ids(self) -> Sequence:
        return _RowToVideoId(self._row2vid, self._video_ids)

    def add(self, db: List[VideoFeature]):
        """vsc/index.py:87-94 (incremental; may be called repeatedly)."""
        if len(db) == 0:
            return
        layout = VideoLayout(db)
        base = len(self._video_ids)
        ordinals = np.empty(len(db), dtype=np.int32)
        for k, vf in enumerate(db):
            # a video id seen before keeps its ordinal (the reference overwrites its metadata)
            o = self._vid_ordinal.get(vf.video_id)
            if o is None:
                o = len(self._video_ids)
                self._vid_ordinal[vf.video_id] = o
                self._video_ids.append(vf.video_id)
            ordinals[k] = o
            self.video_metadata[vf.video_id] = layout.metadata[k]
        del base
        self._row2vid = np.concatenate([self._row2vid, ordinals[layout.row2vid]])
        self._row2frame = np.concatenate([self._row2frame, layout.row2frame])
        self.index.add(VideoLayout.features(db))

    # -- engine-level search returning arrays
    def search_hits(self, queries: List[VideoFeature], global_k: int, device_out: bool = False):
        layout = VideoLayout(queries)
        feats = VideoLayout.features(queries)
        if global_k < 0:
            k = -global_k
            logging.warning(
                "Using local k for KNN search. Warning: this is against the "
                "VSC rules, since predictions for a query-ref pair are not "
                "independent of other references. KNN search is provided for "
                "comparison."
            )
            D, I = self.index.search(feats, k)  # vsc/index.py:167-177: (row, rank) order
            i = np.repeat(np.arange(D.shape[0], dtype=np.int32), k)
            return SearchHits(i, I.reshape(-1).astype(np.int32), D.reshape(-1), layout, self)
        i, j, s, _ = self.index.global_topk(feats, global_k, device_out=device_out)
        return SearchHits(i, j, s, layout, self)

    def search(self, queries: List[VideoFeature], global_k: int) -> List[PairMatches]:
        """vsc/index.py:96-140: hits regrouped per (query video, ref video) in first-appearance
        order of the hit list."""
        hits = self.search_hits(queries, global_k)
        if len(hits) == 0:
            return []
        q_vid = hits.q_layout.row2vid[hits.i].astype(np.int64)
        r_vid = self._row2vid[hits.j].astype(np.int64)
        key = q_vid * max(len(self._video_ids), 1) + r_vid
        _, first, inverse = np.unique(key, return_index=True, return_inverse=True)
        pair_rank = np.empty(len(first), dtype=np.int64)
        pair_rank[np.argsort(first, kind="stable")] = np.arange(len(first))
        order = np.argsort(pair_rank[inverse], kind="stable")  # hits grouped by pair, list order kept
        bounds = np.r_[0, np.cumsum(np.bincount(pair_rank[inverse], minlength=len(first)))]
        q_frame = hits.q_layout.row2frame[hits.i][order]
        r_frame = self._row2frame[hits.j][order]
        scores = hits.s[order]
        q_vid, r_vid = q_vid[order], r_vid[order]
        out = []
        for p in range(len(first)):
            a, b = bounds[p], bounds[p + 1]
            qv, rv = int(q_vid[a]), int(r_vid[a])
            q_id, r_id = hits.q_layout.video_ids[qv], self._video_ids[rv]
            out.append(
                PairMatches(
                    q_id,
                    r_id,
                    _MatchView(hits.q_layout.metadata[qv], self.video_metadata[r_id], q_frame[a:b],
                               r_frame[a:b], scores[a:b]),
                )
            )
        return out

    # -- private helpers of the reference, kept for callers that reach them (vsc/index.py:142-177)
    def _global_threshold_knn_search(self, query_features: np.ndarray, global_k: int) -> Iterable[SearchIndices]:
        i, j, s, _ = self.index.global_topk(query_features, global_k)
        return list(zip(i.tolist(), j.tolist(), s))

    def _knn_search(self, query_features: np.ndarray, k) -> Iterator[SearchIndices]:
        D, I = self.index.search(query_features, k)
        for a in range(I.shape[0]):
            for b in range(I.shape[1]):
                yield (a, I[a, b], D[a, b])
