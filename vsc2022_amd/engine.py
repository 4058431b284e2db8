"""HBM-resident matching engine: search -> candidates -> Temporal-Network localisation with every
intermediate kept on the device (torch tensors are only the memory/stream plumbing; all compute
is libvscmi through the C ABI).

This is the array-level API underneath the `vsc.index` / `vsc.candidates` / `vcsl.vta` mirrors; it
is what bench.py times ("inputs already resident in HBM") and what the multi-GPU driver shards.
The constants are the reference's (vsc/descriptor_eval_lib.py:23-24, vsc/baseline/sscd_baseline.py:
93-94,111,121-124): 1200 hits, 25 candidates and 5 localised pairs per query video; TN with
tn_max_step=5, min_length=4.
"""
import ctypes
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from vsc2022_amd import _lib, dist as vdist
from vsc2022_amd.vcsl.vta import tn_params
from vsc2022_amd.vsc.index import FlatIndex

RETRIEVE_PER_QUERY = 1200
CANDIDATES_PER_QUERY = 25
LOCALIZE_PER_QUERY = 5
REFERENCE_TN = dict(tn_max_step=5, min_length=4)


@dataclass
class MatchResult:
    n_hits: int
    n_pairs: int
    n_candidates: int
    n_localized: int
    n_matches: int
    cand_q: torch.Tensor      # [n_candidates] int32 query video ordinal (global)
    cand_r: torch.Tensor      # [n_candidates] int32 ref video ordinal
    cand_score: torch.Tensor  # [n_candidates] fp32
    loc_index: torch.Tensor   # [n_localized_here] indices into the candidate table localised by this rank
    nbox: torch.Tensor        # [n_localized_here] int32
    boxes: torch.Tensor       # [n_localized_here, 16, 4] int32
    box_score: torch.Tensor   # [n_localized_here, 16] fp32 (MaxSim - bias)
    radius: float
    # Sharded runs (vsc2022_amd/dist.py, module docstring) emulate the reference's schedule itself: the result is the
    # reference's by construction, `radius` the schedule's final radius.
    matches_reference: bool = True
    tie_on_cut: bool = False         # s_K == s_(K+1) over the whole score matrix
    ties_dropped: bool = False       # ... and the reference's schedule ends on that very score: hits tied with it dropped


def _dev_ptr(t: torch.Tensor) -> int:
    assert t.is_cuda and t.is_contiguous()
    return t.data_ptr()


# The entry points that own no handle (vsc_row_normalize, vsc_pair_max) run on a per-device stream of the library; bound to
# torch's current stream (vsc_set_aux_stream) they need no device-wide synchronisation before they read torch's tensors.
_AUX_STREAM = {}


def bind_aux_stream(device: torch.device):
    st = torch.cuda.current_stream(device).cuda_stream
    if _AUX_STREAM.get(device.index) != st:
        _lib.check(_lib.lib().vsc_set_aux_stream(device.index, ctypes.c_void_p(st), 0))
        _AUX_STREAM[device.index] = st


def _after_torch(device: torch.device):
    if os.environ.get("VSC_TORCH_STREAM", "1") != "0":
        bind_aux_stream(device)
        return
    torch.cuda.synchronize(device)


def row_normalize_device(x: torch.Tensor) -> torch.Tensor:
    """Rows scaled to unit L2 norm by libvscmi (zero rows stay zero), HBM in, HBM out."""
    x = x.to(torch.float32).contiguous()
    out = torch.empty_like(x)
    if x.shape[0]:
        _after_torch(x.device)
        _lib.check(_lib.lib().vsc_row_normalize(_dev_ptr(x), x.shape[0], x.shape[1], _lib.MEM_DEVICE, _dev_ptr(out),
                                                _lib.MEM_DEVICE, x.device.index))
    return out


def sort_hits_device(hi: torch.Tensor, hj: torch.Tensor, hs: torch.Tensor, max_row: int = 0, max_ref: int = 0):
    """A hit list in HBM ordered by (score desc, row asc, ref asc) -- vsc/index.py:158-165 -- by libvscmi's radix sorts
    (`vsc_sort_hits`); max_row / max_ref: exclusive bounds of the row / reference numbers (they save sort passes)."""
    n = int(hs.numel())
    hi, hj, hs = hi.to(torch.int32).contiguous(), hj.to(torch.int32).contiguous(), hs.to(torch.float32).contiguous()
    oi, oj, os_ = torch.empty_like(hi), torch.empty_like(hj), torch.empty_like(hs)
    if n:
        _after_torch(hs.device)
        _lib.check(_lib.lib().vsc_sort_hits(_dev_ptr(hi), _dev_ptr(hj), _dev_ptr(hs), n, _lib.MEM_DEVICE, int(max_row),
                                            int(max_ref), _dev_ptr(oi), _dev_ptr(oj), _dev_ptr(os_), _lib.MEM_DEVICE,
                                            hs.device.index))
    return oi, oj, os_


class DeviceScoreNormalizer:
    """`score_normalize` (vsc/baseline/score_normalization.py:31-105) on frame-row tensors that stay in HBM,
    with the noise set resident: built once, then applied to any number of query / reference batches.

    Same algebra as the list-of-VideoFeature mirror in vsc/baseline/score_normalization.py: drop the
    coordinate with the lowest variance over the noise set, row-normalise, append -beta * (best inner
    product with the noise set) to every query row and 1 to every reference row.  The 1-NN runs through
    `vsc_index_knn` (fp16 pre-filter + exact stage: bit-identical to the exact kernel).  The variance is
    taken on the device in float64 (the list mirror keeps numpy's float32 `var` so that it picks the
    reference's column even on near-ties).
    """

    def __init__(self, noise: torch.Tensor, beta: float = 1.0, l2_normalize: bool = True, replace_dim: bool = True):
        self.beta, self.l2_normalize = float(beta), bool(l2_normalize)
        dev = noise.device
        self.sel = None
        if replace_dim:
            weakest = int(noise.to(torch.float64).var(dim=0, unbiased=False).argmin().item())
            keep = [c for c in range(noise.shape[1]) if c != weakest]
            self.sel = torch.tensor(keep, dtype=torch.int64, device=dev)
        noise = self._prepare(noise)
        self.noise_index = FlatIndex(int(noise.shape[1]), _lib.METRIC_INNER_PRODUCT, dev.index)
        if os.environ.get("VSC_TORCH_STREAM", "1") != "0":
            self.noise_index.use_torch_stream()
        self.noise_index.add(noise)

    def _prepare(self, x: torch.Tensor) -> torch.Tensor:
        if self.sel is not None:
            x = x.index_select(1, self.sel)
        return row_normalize_device(x) if self.l2_normalize else x.to(torch.float32).contiguous()

    def queries(self, q: torch.Tensor) -> torch.Tensor:
        q = self._prepare(q)
        best, _ = self.noise_index.search(q, 1, device_out=True)  # [n, 1] in HBM: no host round trip
        return torch.cat([q, best * (-self.beta)], dim=1).contiguous()

    def refs(self, r: torch.Tensor) -> torch.Tensor:
        r = self._prepare(r)
        return torch.cat([r, torch.ones((r.shape[0], 1), dtype=torch.float32, device=r.device)], dim=1).contiguous()


def score_normalize_device(queries: torch.Tensor, refs: torch.Tensor, noise: torch.Tensor, beta: float = 1.0,
                           l2_normalize: bool = True, replace_dim: bool = True):
    """One-shot form of DeviceScoreNormalizer: returns (queries', refs')."""
    norm = DeviceScoreNormalizer(noise, beta=beta, l2_normalize=l2_normalize, replace_dim=replace_dim)
    return norm.queries(queries), norm.refs(refs)


class DeviceMatcher:
    """One GPU's share of the matching pipeline.

    refs / queries: fp32 [rows, dim] torch tensors in HBM (or numpy arrays, uploaded once);
    *_off: int64 numpy row offsets per video.
    """

    def __init__(self, ref_feats, r_off: np.ndarray, device: Optional[int] = None, tn_ref_feats=None):
        """tn_ref_feats: reference rows the ALIGNER sees when they differ from the rows that are searched
        (vsc/baseline/sscd_baseline.py:128-135: without score normalisation the reference searches the descriptors as they
        are and localises on their L2-normalised copies)."""
        self.device = _lib.default_device() if device is None else int(device)
        self.tdev = torch.device("cuda", self.device)
        torch.cuda.set_device(self.tdev)
        self.r_off = np.ascontiguousarray(r_off, dtype=np.int64)
        self.n_rvid = len(self.r_off) - 1
        self.ref_feats = self._as_dev(ref_feats)
        self.tn_ref_feats = self.ref_feats if tn_ref_feats is None else self._as_dev(tn_ref_feats)
        self.dim = int(self.ref_feats.shape[1])
        self.index = FlatIndex(self.dim, _lib.METRIC_INNER_PRODUCT, self.device)
        # the library's handles run on torch's current stream (vsc_index_set_stream & co., round 5): no device-wide
        # synchronisation in front of every call (VSC_TORCH_STREAM=0: the handles' own streams + synchronisations)
        self.torch_stream = os.environ.get("VSC_TORCH_STREAM", "1") != "0"
        if self.torch_stream:
            self.index.use_torch_stream()
        self.index.add(self.ref_feats)
        lens = torch.from_numpy(np.diff(self.r_off)).to(self.tdev)
        self.row2r = torch.repeat_interleave(torch.arange(self.n_rvid, dtype=torch.int32, device=self.tdev), lens)
        self._tn = None
        self._bufs = {}

    def _as_dev(self, x):
        if isinstance(x, torch.Tensor):
            return x.to(self.tdev, torch.float32).contiguous()
        return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(self.tdev)

    def _buf(self, name, n, dtype, cols=None):
        shape = (n,) if cols is None else (n,) + tuple(cols)
        b = self._bufs.get(name)
        if b is None or b.shape[0] < n or b.dtype != dtype or b.shape[1:] != shape[1:]:
            b = torch.empty(shape, dtype=dtype, device=self.tdev)
            self._bufs[name] = b
        return b

    # ---- queries
    def set_queries(self, q_feats, q_off: np.ndarray, tn_q_feats=None):
        self.q_off = np.ascontiguousarray(q_off, dtype=np.int64)
        self.n_qvid = len(self.q_off) - 1
        self.q_feats = self._as_dev(q_feats)
        self.tn_q_feats = self.q_feats if tn_q_feats is None else self._as_dev(tn_q_feats)
        lens = torch.from_numpy(np.diff(self.q_off)).to(self.tdev)
        self.row2q = torch.repeat_interleave(torch.arange(self.n_qvid, dtype=torch.int32, device=self.tdev), lens)
        if self._tn is not None:
            # the references stay packed in the Temporal-Network context: only the query side is replaced
            self._order()
            _lib.check(_lib.lib().vsc_tn_set_queries(self._tn, _dev_ptr(self.tn_q_feats), self.q_off.ctypes.data,
                                                     self.n_qvid, _lib.MEM_DEVICE))
            return
        torch.cuda.synchronize(self.tdev)  # (once: a fresh context packs its rows on its own stream, bound below)
        ctx = ctypes.c_void_p()
        _lib.check(_lib.lib().vsc_tn_create(
            _dev_ptr(self.tn_q_feats), self.q_off.ctypes.data, self.n_qvid, _dev_ptr(self.tn_ref_feats),
            self.r_off.ctypes.data, self.n_rvid, int(self.tn_ref_feats.shape[1]), _lib.MEM_DEVICE, self.device,
            ctypes.byref(ctx)))
        self._tn = ctx
        self._tn_stream = None
        self._order()

    def _order(self):
        """torch's queued work before the library's next reads: (re)bind the handles to torch's current stream, or
        synchronise the device when the library keeps its own streams"""
        if not self.torch_stream:
            torch.cuda.synchronize(self.tdev)
            return
        st = torch.cuda.current_stream(self.tdev).cuda_stream
        if self.index._stream != st:
            self.index.use_stream(st)
        if self._tn is not None and getattr(self, "_tn_stream", None) != st:
            _lib.check(_lib.lib().vsc_tn_set_stream(self._tn, ctypes.c_void_p(st), 0))
            self._tn_stream = st
        bind_aux_stream(self.tdev)

    def __del__(self):
        tn = getattr(self, "_tn", None)
        if tn is not None:
            try:
                _lib.lib().vsc_tn_destroy(tn)
            except Exception:
                pass
            self._tn = None

    # ---- stages (all tensors stay in HBM)
    def search(self, K: int, seed_radius: Optional[float] = None, rows: Optional[torch.Tensor] = None,
               schedule: bool = False, index: Optional[FlatIndex] = None):
        """vsc/index.py:142-165 over the resident queries: (i, j, s) sorted hits + final radius.

        seed_radius: the rows run as steady batches from this radius (`vsc_index_global_topk_seeded`) instead of
        replaying the doubling schedule: every hit beyond it, cut at K (the sharded pipeline's row lists).
        rows: search these query rows instead (an [n, dim] tensor in HBM; row numbers are relative to it).
        schedule: replay the reference's batch schedule even where the library would take its proven top-K route
        (include/vscmi.h, "topk_shortcut"): the returned radius is then the SCHEDULE's final radius."""
        index = self.index if index is None else index
        if schedule and seed_radius is None and index.get_option("topk_shortcut") != 0:
            keep = index.get_option("topk_shortcut")
            index.set_option("topk_shortcut", 0)
            try:
                return self.search(K, None, rows, index=index)
            finally:
                index.set_option("topk_shortcut", keep)
        q = self.q_feats if rows is None else rows.to(self.tdev, torch.float32).contiguous()
        nq = int(q.shape[0])
        cap = int(max(1, min(K, nq * max(index.ntotal, 1))))
        tag = "hit" if rows is None else "seed_hit"
        oi = self._buf(tag + "_i", cap, torch.int32)
        oj = self._buf(tag + "_j", cap, torch.int32)
        os_ = self._buf(tag + "_s", cap, torch.float32)
        n_out, radius = ctypes.c_int64(0), ctypes.c_float(0.0)
        self._order()
        if index is not self.index and self.torch_stream:
            st = torch.cuda.current_stream(self.tdev).cuda_stream
            if index._stream != st:   # (vsc_index_set_stream synchronises the stream it leaves: only on a real change)
                index.use_stream(st)
        if seed_radius is None:
            _lib.check(_lib.lib().vsc_index_global_topk(
                index.handle, _dev_ptr(q), nq, _lib.MEM_DEVICE, int(K),
                _dev_ptr(oi), _dev_ptr(oj), _dev_ptr(os_), cap, _lib.MEM_DEVICE, ctypes.byref(n_out),
                ctypes.byref(radius)))
        else:
            _lib.check(_lib.lib().vsc_index_global_topk_seeded(
                index.handle, _dev_ptr(q), nq, _lib.MEM_DEVICE, int(K), float(seed_radius),
                _dev_ptr(oi), _dev_ptr(oj), _dev_ptr(os_), cap, _lib.MEM_DEVICE, ctypes.byref(n_out),
                ctypes.byref(radius)))
        m = n_out.value
        return oi[:m], oj[:m], os_[:m], radius.value

    def pair_max(self, hi, hj, hs):
        """vsc/index.py:121-140 + vsc/candidates.py:24-40 on device hits."""
        n = int(hs.numel())
        oq = self._buf("pair_q", max(n, 1), torch.int32)
        orr = self._buf("pair_r", max(n, 1), torch.int32)
        os_ = self._buf("pair_s", max(n, 1), torch.float32)
        of = self._buf("pair_f", max(n, 1), torch.int64)
        n_pairs = ctypes.c_int64(0)
        if n:
            hi, hj, hs = hi.contiguous(), hj.contiguous(), hs.contiguous()
            self._order()
            _lib.check(_lib.lib().vsc_pair_max(
                _dev_ptr(hi), _dev_ptr(hj), _dev_ptr(hs), n, _lib.MEM_DEVICE, _dev_ptr(self.row2q),
                int(self.row2q.numel()), _dev_ptr(self.row2r), int(self.row2r.numel()), _lib.MEM_DEVICE,
                _dev_ptr(oq), _dev_ptr(orr), _dev_ptr(os_), _dev_ptr(of), n, _lib.MEM_DEVICE,
                ctypes.byref(n_pairs), self.device))
        m = n_pairs.value
        return oq[:m], orr[:m], os_[:m], of[:m]

    def localize(self, pair_q: torch.Tensor, pair_r: torch.Tensor, bias: float = 0.0, **tn_kwargs):
        """vsc/baseline/localization.py:56-96 for device-resident pair lists (LOCAL query ordinals)."""
        n = int(pair_q.numel())
        nbox = self._buf("tn_nbox", max(n, 1), torch.int32)
        boxes = self._buf("tn_boxes", max(n, 1), torch.int32, (_lib.TN_MAX_BOXES, 4))
        bmax = self._buf("tn_bmax", max(n, 1), torch.float32, (_lib.TN_MAX_BOXES,))
        if n:
            prm = tn_params(**(tn_kwargs or REFERENCE_TN))
            pq, pr = pair_q.to(torch.int32).contiguous(), pair_r.to(torch.int32).contiguous()
            self._order()
            _lib.check(_lib.lib().vsc_tn_localize(
                self._tn, _dev_ptr(pq), _dev_ptr(pr), n, _lib.MEM_DEVICE, ctypes.byref(prm), float(bias),
                _dev_ptr(nbox), _dev_ptr(boxes), _dev_ptr(bmax), _lib.MEM_DEVICE))
        return nbox[:n], boxes[:n], bmax[:n]

    # ---- the whole hot path
    def match(self, n_qvid_global: Optional[int] = None, qvid_base: int = 0, row_base: int = 0, group=None,
              bias: float = 0.0, localize: bool = True) -> MatchResult:
        """search -> candidates -> localisation for the resident queries.

        Single process: n_qvid_global is None.  Sharded (one process per GPU): this rank owns the
        query videos [qvid_base, qvid_base + n_qvid) / rows [row_base, ...) of a global query set
        of n_qvid_global videos; the two global cuts are resolved with vsc2022_amd.dist.
        localize=False stops after the candidate table (vsc/descriptor_eval_lib.py:42-49: no aligner).
        """
        sharded = n_qvid_global is not None and torch.distributed.is_initialized() and \
            (torch.distributed.get_world_size(group) > 1 or os.environ.get("VSC_FORCE_SHARDED") == "1")
        nq_glob = n_qvid_global if n_qvid_global is not None else self.n_qvid
        K = int(RETRIEVE_PER_QUERY * nq_glob)
        n_cand_cut = int(CANDIDATES_PER_QUERY * nq_glob)
        n_loc_cut = int(LOCALIZE_PER_QUERY * nq_glob)
        if not sharded:
            hi, hj, hs, radius = self.search(K)
            self.last_hits = (hi, hj, hs)   # (views of the search's output buffers: valid until the next search)
            pq, pr, ps, pf = self.pair_max(hi, hj, hs)
            n_cand = min(int(ps.numel()), n_cand_cut)
            n_loc = min(n_cand, n_loc_cut) if localize else 0
            nbox, boxes, bmax = self.localize(pq[:n_loc], pr[:n_loc], bias)
            return MatchResult(int(hs.numel()), int(ps.numel()), n_cand, n_loc, int(nbox.sum().item()),
                               pq[:n_cand], pr[:n_cand], ps[:n_cand],
                               torch.arange(n_loc, device=self.tdev), nbox, boxes, bmax, radius)
        # The reference's schedule itself, over query shards (vsc2022_amd/dist.py, module docstring + emulate_schedule): a
        # proof about the K cut cannot replace it -- at BASELINE's sizes a tie on the cut is certain.
        hi, hj, hs, radius, info = self.sharded_schedule_search(K, nq_glob_rows=None, row_base=row_base, group=group)
        # (a list shorter than K although the matrix holds more: the schedule ended ON the K-th best score and dropped
        # everything tied with it, as the reference does)
        dropped = info.total < min(K, self.last_shard_stats["n_rows"] * self.index.ntotal)
        proven, tie = True, bool(info.tie_on_cut or dropped)
        n_take = int(hs.numel())
        self.last_hits = (hi, hj, hs)       # this rank's share of the K hits (rows LOCAL)
        pq, pr, ps, pf = self.pair_max(hi, hj, hs)
        first_i = hi[pf].to(torch.int64) + row_base if pf.numel() else pf
        first_j = hj[pf].to(torch.int64) if pf.numel() else pf
        cands = vdist.merge_candidates(pq + qvid_base, pr, ps, first_i, first_j, n_cand_cut, group)
        n_cand = len(cands)
        n_loc = min(n_cand, n_loc_cut) if localize else 0
        mine = (cands.q_vid[:n_loc] >= qvid_base) & (cands.q_vid[:n_loc] < qvid_base + self.n_qvid)
        loc_index = torch.nonzero(mine).flatten()
        nbox, boxes, bmax = self.localize(cands.q_vid[loc_index] - qvid_base, cands.r_vid[loc_index], bias)
        n_matches = vdist.all_reduce_sum_int(int(nbox.sum().item()), self.tdev, group)
        n_hits = vdist.all_reduce_sum_int(n_take, self.tdev, group)
        return MatchResult(n_hits, int(ps.numel()), n_cand, n_loc, n_matches,
                           cands.q_vid, cands.r_vid, cands.score, loc_index, nbox, boxes, bmax, radius,
                           matches_reference=proven, tie_on_cut=tie, ties_dropped=dropped)

    # ---- the query-sharded search
    def _rows_above(self, rows: torch.Tensor, radius: float, budget: int, index: Optional[FlatIndex] = None):
        """EVERY pair of the query rows `rows` with score > radius (strict): (i relative to rows, j, s), sorted by (score
        desc, row asc, ref asc).  Steady batches from `radius` (`vsc_index_global_topk_seeded`: pre-filters + exact stage)
        with a budget the list must stay below -- a full list may be a truncated one: the budget doubles and the rows run
        again.  index: another index than the matcher's (the rank's column slice of the references).

        A list SHORTER than the budget is complete only if the search's radius never moved: the seeded entry keeps the
        schedule's re-threshold rule (more than 2 x budget kept -> the radius becomes the (budget+1)-th best, strict
        filter), and with a group of equal scores at that position fewer than `budget` hits come back although every
        hit in (radius, new radius] is gone.  Same acceptance rule as `FlatIndex.range_scores`."""
        idx = self.index if index is None else index
        cap_all = int(rows.shape[0]) * max(idx.ntotal, 1)
        budget = max(1024, min(int(budget), cap_all))
        while True:
            i, j, s, rad = self.search(budget, seed_radius=float(radius), rows=rows, index=idx)
            if budget >= cap_all or (int(s.numel()) < budget and np.float32(rad) == np.float32(radius)):
                break
            self.rows_above_reruns = getattr(self, "rows_above_reruns", 0) + 1
            budget = min(cap_all, budget * 2)
        return i.clone(), j.clone(), s.clone()

    def sharded_schedule_search(self, K: int, nq_glob_rows: Optional[int], row_base: int, group=None):
        """vsc/index.py:142-165 for a query set that is sharded over ranks: this rank's share of the reference's K hits
        (sorted; rows LOCAL), the schedule's final radius and the selection's report.  Identical to what one process
        returns for the whole query set, ties included.

        The schedule (range_search_max_results over exponential_query_iterator) is a radius, a list of kept hits and
        GLOBAL row batches that must be walked in order -- the radius a batch is searched at is decided by the batches
        before it --; every decision it takes is a count or an order statistic of the kept list, i.e. a sum over any
        partition of that list.  The score matrix of a batch is split by reference COLUMNS (VSC_SHARD_MODE=cols, default):
          1. the (score-normalised) query rows are all-gathered once (2 GB at configs[3]) and every rank keeps an index of
             its 1/world slice of the reference rows next to the full set the localisation needs;
          2. `dist.emulate_schedule` walks the batches in lockstep: every rank searches ALL rows of the batch against its
             slice at exactly the schedule's radius (the library's steady-batch entry, hits unsorted), one all-reduce of the
             kept count per batch, the exact distributed (K+1)-th best at every event;
          3. the kept hits go to the ranks that own their query rows (one all-to-all), where the exact distributed
             selection cuts {s > final radius} at K and candidate generation / localisation carry on per query video.
        Every rank does 1/world of every launch of the single-process search -- the same candidates, the same exact-stage
        work, no prediction and nothing searched twice; measured on ranks sharing one GPU the searches of all ranks
        together take 1.30 s (2 ranks) / 1.38 s (4 ranks) against 1.26 s in one process.

        VSC_SHARD_MODE=rows is this round's first design, kept for comparison: only the doubling batches at the head are
        split by columns; for the steady batches every rank lists its OWN rows' hits beforehand above a floor predicted
        from a row sample (`dist.predict_schedule_density`) and the schedule filters those lists, searching a batch on
        demand where the floor was too high.  It needs no all-gather of the queries, but the rank that owns the first rows
        (low radii: three times the hits of the others) carries most of the exact-stage work, and every list is 1.25 x what
        the schedule needs.  VSC_SHARD_SPEC_START=<row> (default 65504): where its prepared batches begin."""
        dev = self.tdev
        nq_loc = int(self.q_feats.shape[0])
        nr = self.index.ntotal
        n_rows = vdist.all_reduce_sum_int(nq_loc, dev, group) if nq_glob_rows is None else int(nq_glob_rows)
        # VSC_SHARD_MODE=cols (default): EVERY batch of the schedule is split by reference columns (below); =rows: only the
        # doubling batches are, the steady ones come from lists the row owners prepared (steps 1-2 above)
        by_rows = os.environ.get("VSC_SHARD_MODE", "cols") == "rows"
        spec_start = max(32, int(os.environ.get("VSC_SHARD_SPEC_START", "65504"))) if by_rows else n_rows + 1
        spec_factor = float(os.environ.get("VSC_SHARD_SPEC_FACTOR", "1.25"))
        stats = dict(n_rows=n_rows, prepared=0, prepared_hits=0, on_demand=0, floor_too_high=0)
        debug = os.environ.get("VSC_SHARD_DEBUG") == "1"
        import sys
        import time as _time

        # phases accounted without synchronising (dist.PhaseTimer: host wall time + HIP-event device time per phase, bytes per
        # collective) -- always on; VSC_SHARD_DEBUG=1 additionally synchronises around the legacy t_* wall clocks below
        timer = vdist.PhaseTimer(dev)

        def clock():
            if debug:
                torch.cuda.synchronize()
            return _time.perf_counter()

        t_start = clock()

        def local(r0, r1):
            return max(r0, row_base) - row_base, min(r1, row_base + nq_loc) - row_base

        # (batch start, local rows, predicted hits per row above the schedule's radius there)
        pieces = [(r0,) + local(r0, r1) + (d,) for r0, r1, d in vdist.predict_schedule_density(n_rows, K, nr) if r0 >= spec_start]
        pieces = [(r0, a, b, d) for r0, a, b, d in pieces if a < b]
        lists = {}
        m = min(int(os.environ.get("VSC_SHARD_SAMPLE", "4096")), nq_loc // 4)
        if pieces and m >= 16 and nr > 0:
            stride = nq_loc // m
            sample = self.q_feats[::stride][:m]
            m = int(sample.shape[0])
            # the floor of a batch: the score below which `spec_factor` x the predicted hits of the sample's rows lie, never
            # more than 2.3 K m / r0 of them (the radius at row r0 is never below the (2K+1)-th best of the rows before it)
            need = [int(np.ceil(min(spec_factor * d, 2.3 * K / r0) * m)) for r0, _, _, d in pieces]
            k_s = int(min(m * nr, max(need) + max(need) // 16 + 1024))
            _, _, ss, _ = self.search(k_s, rows=sample, schedule=True)
            ss = ss.clone()
            stats["t_sample"] = clock() - t_start
            for (r0, a, b, _), c in zip(pieces, need):
                if c > int(ss.numel()):
                    continue  # (the sample does not reach that deep: the batch is searched when the schedule gets there)
                floor = float(np.nextafter(np.float32(ss[c - 1].item()), np.float32(-np.inf)))
                with timer.phase("prepare_row_lists"):
                    i, j, sc = self._rows_above(self.q_feats[a:b], floor, int(1.3 * (b - a) * c / m) + (1 << 16))
                lists[(a, b)] = (floor, i + a, j, sc)
                stats["prepared"] += 1
                stats["prepared_hits"] += int(sc.numel())
        empty = (torch.zeros(0, dtype=torch.int32, device=dev), torch.zeros(0, dtype=torch.int32, device=dev),
                 torch.zeros(0, dtype=torch.float32, device=dev))

        # The head of the query set (the doubling batches: every one of them hands ~K hits to the exact stage whatever its
        # size, none of it shrinks with the number of ranks if the rank that owns the rows searches them alone) is split by
        # reference COLUMNS instead: every rank gets the head's rows and searches them against its slice of the references.
        rank, world = vdist._world(group)
        head_end = min([r0 for r0, _ in vdist.exponential_batches(n_rows) if r0 >= spec_start] + [n_rows])
        by_cols = world > 1 and head_end > 0 and nr >= 64 * world and os.environ.get("VSC_SHARD_HEAD_COLS", "1") != "0"
        if by_cols:
            ha, hb = local(0, head_end)
            t0 = clock()
            with timer.phase("gather_queries"):
                head_q = vdist.all_gather_varlen(self.q_feats[ha:max(ha, hb)], group)   # rank order = row order
            timer.add_bytes("gather_queries", int(head_q.numel()) * 4)
            stats["t_gather"] = clock() - t0
            c0, c1 = [(x // 64) * 64 for x in vdist.shard_ranges(nr, world)[rank]]
            if rank == world - 1:
                c1 = nr
            col_index = getattr(self, "_col_index", None)
            if col_index is None or self._col_range != (c0, c1):
                col_index = FlatIndex(self.dim, _lib.METRIC_INNER_PRODUCT, self.device)
                if self.torch_stream:
                    col_index.use_torch_stream()
                col_index.set_option("sort_hits", 0)   # (a batch's hits join a list that is sorted once, at the end)
                col_index.add(self.ref_feats[c0:c1])
                self._col_index, self._col_range = col_index, (c0, c1)

        def head_budget(r0, n_here, share):
            return int(min(2.5 * K, 4.0 * K * n_here / max(r0, n_here)) * share) + (1 << 20)

        def search_rows(r0, r1, radius):
            if by_cols and r0 < head_end:
                t0 = clock()
                # the exact / fp16 / int8 rule of the single-process schedule for this batch (K / (rows so far x references));
                # the budget below is several times the hits expected and would send int8 batches to the fp16 kernel
                col_index.set_option("density_hint", min(1.0, K / (float(r0) * nr)) if r0 > 0 else 1.0)
                # the budget: 1.3 x an even share of what the whole batch can hold -- or 1.5 x the share THIS slice showed in the
                # batch before, whichever is larger (with copied / clustered reference videos a slice can hold several
                # times its even share, and an undersized budget costs a second search of the batch)
                share = max(1.3 / world, min(1.0, 1.5 * stats.get("slice_share", 0.0)))
                with timer.phase("search"):
                    i, j, sc = self._rows_above(head_q[r0:r1], radius, head_budget(r0, r1 - r0, share), index=col_index)
                stats["slice_share"] = float(sc.numel()) / max(head_budget(r0, r1 - r0, 1.0) - (1 << 20), 1)
                stats["slice_share_max"] = max(stats.get("slice_share_max", 0.0), stats["slice_share"])
                stats["on_demand"] += 1
                stats["t_on_demand"] = stats.get("t_on_demand", 0.0) + clock() - t0
                return i + r0, j + c0, sc          # GLOBAL rows until the hand-over
            a, b = local(r0, r1)
            if a >= b:
                return empty
            got = lists.get((a, b))
            if got is not None and got[0] <= radius:
                n = int((got[3] > radius).sum().item())   # (sorted by score: a prefix)
                return got[1][:n], got[2][:n], got[3][:n]
            stats["floor_too_high" if got is not None else "on_demand"] += 1
            if debug and got is not None:
                print(f"[shard {row_base}] batch at row {r0}: floor {got[0]:.6f} > radius {radius:.6f} ({int(got[3].numel())} listed)",
                      file=sys.stderr, flush=True)
            t0 = clock()
            with timer.phase("search"):
                i, j, sc = self._rows_above(self.q_feats[a:b], radius, head_budget(r0, b - a, 1.0))
            stats["t_on_demand"] = stats.get("t_on_demand", 0.0) + clock() - t0
            return i + a, j, sc

        def to_row_owners(i, j, sc):
            """the kept hits of the column-split head -> the ranks that own their query rows (local row numbers)"""
            t0 = clock()
            bases = torch.tensor(vdist._all_gather_scalar(row_base, dev, group), dtype=torch.int64, device=dev)
            owner = torch.searchsorted(bases, i.to(torch.int64), right=True) - 1
            with timer.phase("handover"):
                got = vdist.send_to_owners(torch.stack([i, j, sc.view(torch.int32)], dim=1), owner, group)
            timer.add_bytes("handover", int(i.numel()) * 12)
            stats["t_handover"] = clock() - t0
            return got[:, 0] - row_base, got[:, 1].contiguous(), got[:, 2].contiguous().view(torch.float32)

        stats["t_prepare"] = clock() - t_start
        t0 = clock()
        radius, hi, hj, hs = vdist.emulate_schedule(search_rows, n_rows, K, group, dev,
                                                    handover=(head_end, to_row_owners) if by_cols else None,
                                                    timer=timer, n_cols_total=nr)
        stats["t_emulate"] = clock() - t0
        t0 = clock()
        # (score desc, row asc, ref asc): the order the reference's stable sort leaves, then the global cut at K
        with timer.phase("final_sort_and_cut"):
            if hs.numel():
                hi, hj, hs = sort_hits_device(hi, hj, hs, max_row=nq_loc, max_ref=nr)
            n_take, tau, info = vdist.distributed_prefix_select(hs, K, group, ties="rank", return_info=True)
        stats["t_final"] = clock() - t0
        # (the selection's host sync is behind us: reading the events costs nothing more)
        stats["phases"] = timer.collect()
        stats["count_skipped"] = timer.calls.get("count_skipped", 0)
        stats["rows_above_reruns"] = getattr(self, "rows_above_reruns", 0)
        self.last_shard_stats = stats
        if debug:
            print(f"[shard {row_base}] " + " ".join(f"{k}={v:.3f}" if isinstance(v, float) else f"{k}={v}" for k, v in stats.items() if k != "phases"),
                  file=sys.stderr, flush=True)
        return hi[:n_take], hj[:n_take], hs[:n_take], radius, info

    def gather_boxes(self, res: MatchResult, group=None) -> torch.Tensor:
        """Every rank's localisation results in one table (the hand-over of vsc/baseline/sscd_baseline.py:139-152 when
        the pairs were localised by the ranks that own their query videos): int64 [n_boxes, 6] = (index into the
        candidate table, q_lo, r_lo, q_hi, r_hi, bits of the fp32 box score), ordered by (candidate, box) -- the
        order in which the single-process pipeline emits its Match rows.  Identical on every rank."""
        n_here = int(res.loc_index.numel())
        if n_here:
            slot = torch.arange(_lib.TN_MAX_BOXES, device=self.tdev).unsqueeze(0).expand(n_here, -1)
            valid = slot < res.nbox.to(torch.int64).unsqueeze(1)
            cand = res.loc_index.to(torch.int64).unsqueeze(1).expand(-1, _lib.TN_MAX_BOXES)
            rows = torch.cat([cand[valid].unsqueeze(1), slot[valid].unsqueeze(1).to(torch.int64),
                              res.boxes.to(torch.int64)[valid],
                              res.box_score.contiguous().view(torch.int32).to(torch.int64)[valid].unsqueeze(1)], dim=1)
        else:
            rows = torch.zeros((0, 7), dtype=torch.int64, device=self.tdev)
        allr = vdist.all_gather_varlen(rows, group)
        if allr.shape[0]:
            order = torch.sort(allr[:, 0] * _lib.TN_MAX_BOXES + allr[:, 1]).indices
            allr = allr[order]
        return torch.cat([allr[:, :1], allr[:, 2:]], dim=1)
