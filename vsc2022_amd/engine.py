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
    # Sharded runs (vsc2022_amd/dist.py, module docstring): the exact global top-K IS the reference's result unless a tie
    # sits on the K cut; then the reference's final radius decides whether the tied hits are dropped.
    matches_reference: bool = True   # proven identical to vsc/index.py:142-165 on this query set (False only when a tie
    #                                  sits on the cut and VSC_SHARD_TIE_RESOLVE=0 switched the resolution off)
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
    def search(self, K: int, seed_radius: Optional[float] = None, rows: Optional[torch.Tensor] = None):
        """vsc/index.py:142-165 over the resident queries: (i, j, s) sorted hits + final radius.

        seed_radius: a radius known to lie below the K-th best score (sharded pipeline): the rows run as steady
        batches from there (`vsc_index_global_topk_seeded`) instead of replaying the doubling schedule.
        rows: search these query rows instead (an [n, dim] tensor in HBM; row numbers are relative to it)."""
        q = self.q_feats if rows is None else rows.to(self.tdev, torch.float32).contiguous()
        nq = int(q.shape[0])
        cap = int(max(1, min(K, nq * max(self.index.ntotal, 1))))
        tag = "hit" if rows is None else "seed_hit"
        oi = self._buf(tag + "_i", cap, torch.int32)
        oj = self._buf(tag + "_j", cap, torch.int32)
        os_ = self._buf(tag + "_s", cap, torch.float32)
        n_out, radius = ctypes.c_int64(0), ctypes.c_float(0.0)
        self._order()
        if seed_radius is None:
            _lib.check(_lib.lib().vsc_index_global_topk(
                self.index.handle, _dev_ptr(q), nq, _lib.MEM_DEVICE, int(K),
                _dev_ptr(oi), _dev_ptr(oj), _dev_ptr(os_), cap, _lib.MEM_DEVICE, ctypes.byref(n_out),
                ctypes.byref(radius)))
        else:
            _lib.check(_lib.lib().vsc_index_global_topk_seeded(
                self.index.handle, _dev_ptr(q), nq, _lib.MEM_DEVICE, int(K), float(seed_radius),
                _dev_ptr(oi), _dev_ptr(oj), _dev_ptr(os_), cap, _lib.MEM_DEVICE, ctypes.byref(n_out),
                ctypes.byref(radius)))
        m = n_out.value
        return oi[:m], oj[:m], os_[:m], radius.value

    def seed_radius(self, K: int, group=None) -> Optional[float]:
        """A radius just below the K-th best score of the GLOBAL score matrix, estimated over a strided sample of every
        rank's query rows (same stride everywhere): each rank searches its sample with the reference's schedule, the
        1.25 * K * (sample share)-th best score over all ranks' sample hits is found with the exact distributed
        selection (two histogram all-reduces + one all-gather of tie counts).  None when the query set is too small for
        a sample to pay (then the ranks replay the schedule as before).  An estimate, never trusted: the caller checks
        the seeded result with the same exactness test as any other local search and falls back when it fails."""
        nq_loc = int(self.q_feats.shape[0])
        tot_rows = vdist.all_reduce_sum_int(nq_loc, self.tdev, group)
        target = int(os.environ.get("VSC_SHARD_SEED_ROWS", "4096"))  # sample rows over all ranks
        stride = tot_rows // max(target, 1)
        if stride < 4 or self.index.ntotal == 0:
            return None
        pick = torch.arange(0, nq_loc, stride, device=self.tdev)
        s_loc = int(pick.numel())
        s_tot = vdist.all_reduce_sum_int(s_loc, self.tdev, group)
        k_tot = int(np.ceil(1.25 * K * s_tot / tot_rows))
        if s_loc:
            k_s = int(min(s_loc * self.index.ntotal, np.ceil(2.0 * K * s_loc / tot_rows) + 1024))
            _, _, sc, _ = self.search(k_s, rows=self.q_feats.index_select(0, pick))
        else:
            sc = torch.zeros(0, dtype=torch.float32, device=self.tdev)
        _, tau = vdist.distributed_prefix_select(sc, k_tot, group)
        if not np.isfinite(tau):
            return None
        return float(np.nextafter(np.float32(tau), np.float32(-np.inf)))

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
            pq, pr, ps, pf = self.pair_max(hi, hj, hs)
            n_cand = min(int(ps.numel()), n_cand_cut)
            n_loc = min(n_cand, n_loc_cut) if localize else 0
            nbox, boxes, bmax = self.localize(pq[:n_loc], pr[:n_loc], bias)
            return MatchResult(int(hs.numel()), int(ps.numel()), n_cand, n_loc, int(nbox.sum().item()),
                               pq[:n_cand], pr[:n_cand], ps[:n_cand],
                               torch.arange(n_loc, device=self.tdev), nbox, boxes, bmax, radius)
        radius_box = [float("nan")]
        # The sharded search computes the exact global top-K (vsc2022_amd/dist.py) -- which is the reference's result
        # unless a tie sits on the K cut (resolved below) --, so a rank need not replay the 32, 64, ... doubling batches on
        # its own rows: a radius agreed over a row sample seeds every rank's FIRST local search; a retry (seed too high,
        # skewed shard) takes the unseeded search.  Budgets go up to K + 1: the selection must see whether the (K+1)-th
        # best ties with the K-th.
        seed = [self.seed_radius(K, group) if os.environ.get("VSC_SHARD_SEED", "1") != "0" else None]

        def local_search(k_local):
            if seed[0] is not None:
                try:
                    i, j, sc, rad = self.search(K + 1, seed_radius=seed[0])  # (full budget: see dist.sharded_hits)
                except _lib.VscError as e:
                    # a seed far too low (unrepresentative sample) can overflow the kept-hit buffer: the unseeded
                    # schedule bounds its own buffers
                    if e.code not in (_lib.VSC_ERR_OVERFLOW, _lib.VSC_ERR_CAPACITY, _lib.VSC_ERR_NOMEM):
                        raise
                    seed[0] = None
                    i, j, sc, rad = self.search(k_local)
                    radius_box[0] = rad
                    return i, j, sc, rad
                seed[0] = None
                radius_box[0] = rad
                return i, j, sc, rad, True
            i, j, sc, rad = self.search(k_local)
            radius_box[0] = rad
            return i, j, sc, rad

        hi, hj, hs, tau, info = vdist.sharded_hits(local_search, int(self.q_feats.shape[0]) * self.index.ntotal, K,
                                                   group, self.tdev, return_info=True)
        # s_K == s_(K+1): the reference drops every hit tied with the cut iff its schedule's final radius is that score
        # (dist.py module docstring).  Rare (duplicate frames of static videos put a few percent of the query sets here)
        # and decided exactly: rank 0 replays the reference's schedule on the gathered query rows.
        keep, proven, dropped = vdist.resolve_tie_on_cut(hs, tau, info, lambda: self.reference_radius(K, group),
                                                         os.environ.get("VSC_SHARD_TIE_RESOLVE", "1") != "0")
        if dropped:
            hi, hj, hs = hi[keep], hj[keep], hs[keep]
        radius = radius_box[0]
        n_take = int(hs.numel())
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
                           matches_reference=proven, tie_on_cut=info.tie_on_cut, ties_dropped=dropped)

    def reference_radius(self, K: int, group=None) -> float:
        """The final radius of the reference's schedule (vsc/index.py:147-154) over ALL ranks' query rows: the query
        rows are gathered (rank order = query order) and rank 0 runs the single-GPU search on them -- the code whose
        parity with the oracle the single-process suites pin --, then tells everybody.  Only the sharded pipeline's
        tie-on-the-cut case calls this."""
        allq = vdist.all_gather_varlen(self.q_feats, group)
        rank, world = vdist._world(group)
        rad = 0.0
        if rank == 0:
            _, _, _, rad = self.search(K, rows=allq)
        t = torch.tensor([float(rad) if rank == 0 else 0.0], dtype=torch.float64, device=self.tdev)
        vdist._all_reduce_sum(t, group)
        return float(t.item())

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
