"""Reference-sharded search (BASELINE configs[4]): every rank holds a contiguous slice
of the reference rows in its own HBM-resident flat index, every rank sees ALL queries, and the per-shard
results are merged over `torch.distributed` (RCCL on GPUs, gloo in the CPU tests).

This is what the reference gets from FAISS when the index does not fit one device
(`vsc/index.py:153` `ngpu=-1`, `vsc/index.py:171` / `vsc/baseline/score_normalization.py:88-89`
`faiss.index_cpu_to_all_gpus`, IndexShards): same results as one index over the concatenated rows.

  * `search(x, k)`        per-row k-NN: `vsc_index_knn` on the shard, local ids + row offset, one
                           all-gather of the [nq, k] (score, id) pairs, per-row merge by (score desc, id asc)
                           (`dist.ref_sharded_knn`).
  * `global_topk(x, K)`   the K best pairs of the whole score matrix (the search of
                           `vsc/index.py:142-165` when the columns are sharded): per-shard global-threshold
                           search with a local budget, exact distributed selection of the K-th best score
                           (histogram all-reduce, `dist.distributed_prefix_select`), all-gather of the
                           survivors, final order (score desc, query row asc, ref row asc).

Result contract: identical to a single index over all rows (bit for bit: scores are the same fp32 fma
chains), the reference's tie-dropping included: when the (K+1)-th best score equals the K-th the final radius
of the reference's schedule is computed over the shards (`dist.emulate_schedule_radius`: every batch of the
schedule on all shards at once) and the hits tied with the cut are dropped iff the reference drops them
(`vsc2022_amd/dist.py`, module docstring).
"""
import os
from typing import Optional, Tuple

import numpy as np
import torch

from vsc2022_amd import dist as vdist


class RefShardedIndex:
    """One rank's slice of a reference set + the collectives that make it look like one index.

    `local_index` quacks like `vsc2022_amd.vsc.index.FlatIndex` (`.search(x, k) -> (D, I)`,
    `.global_topk(x, K, device_out=...) -> (i, j, s, radius)`, `.ntotal`); the CPU tests plug in an
    oracle-backed stand-in, production uses FlatIndex.  `row0` = global id of the shard's first row,
    `n_total` = rows of the whole reference set.
    """

    def __init__(self, local_index, row0: int, n_total: int, group=None, device: Optional[torch.device] = None):
        self.local, self.row0, self.n_total, self.group = local_index, int(row0), int(n_total), group
        self.device = device if device is not None else torch.device("cpu")

    @classmethod
    def build(cls, ref_rows, dim: int, metric: int, device_index: int, group=None) -> "RefShardedIndex":
        """Shard `ref_rows` ([n, dim] array / tensor that every rank can see, e.g. an np.memmap of the
        descriptor file) by `dist.shard_ranges` and add this rank's rows to a FlatIndex on its GPU."""
        from vsc2022_amd.vsc.index import FlatIndex

        rank, world = vdist._world(group)
        n_total = int(ref_rows.shape[0])
        lo, hi = vdist.shard_ranges(n_total, world)[rank]
        idx = FlatIndex(dim, metric, device_index)
        if hi > lo:
            idx.add(ref_rows[lo:hi])
        return cls(idx, lo, n_total, group, torch.device("cuda", device_index))

    @property
    def ntotal(self) -> int:
        return self.n_total

    # ---- per-row k-NN (faiss index.search over IndexShards)
    def search(self, x, k: int) -> Tuple[np.ndarray, np.ndarray]:
        """(D float32 [n, k], I int64 [n, k]) with GLOBAL reference ids; identical on every rank."""
        n = int(x.shape[0])
        n_loc = int(self.local.ntotal)
        on_gpu = self.device.type == "cuda"
        # L2: distances ascend -- merge their negation (larger is better, ties by id ascending either way) and negate back
        l2 = getattr(self.local, "metric_type", 0) == 1
        # missing slots carry the single index's sentinel (-/+FLT_MAX and id -1, include/vscmi.h), not -/+inf
        D = torch.full((n, k), -float(np.finfo(np.float32).max), dtype=torch.float32, device=self.device)
        I = torch.full((n, k), -1, dtype=torch.int64, device=self.device)
        kk = min(k, n_loc)
        if kk > 0 and n > 0:
            if on_gpu:
                d, i = self.local.search(x, kk, device_out=True)  # stays in HBM until the all-gather
            else:
                d, i = (torch.from_numpy(np.ascontiguousarray(a)) for a in self.local.search(x, kk))
            D[:, :kk] = -d if l2 else d
            I[:, :kk] = torch.where(i >= 0, i + self.row0, torch.full_like(i, -1))
        gD, gI = vdist.ref_sharded_knn(D, I, k, self.group)
        if l2:
            gD = -gD
        return gD.cpu().numpy(), gI.cpu().numpy()

    def _range_scores(self, rows, radius: float, k_hint: int) -> torch.Tensor:
        """fp32 scores > radius (strict) of `rows` against this shard, on `self.device`."""
        n_loc = int(self.local.ntotal)
        if int(rows.shape[0]) == 0 or n_loc == 0:
            return torch.zeros(0, dtype=torch.float32, device=self.device)
        s = self.local.range_scores(rows, radius, k_hint, device_out=self.device.type == "cuda")
        if not isinstance(s, torch.Tensor):
            s = torch.from_numpy(np.ascontiguousarray(s))
        return s.to(self.device)

    # ---- global top-K of the whole score matrix
    def global_topk(self, x, K: int, k_local_start: Optional[int] = None):
        """(i int32, j int64 GLOBAL ref row, s float32, tau): the K best (query row, ref row) pairs over all
        shards, ordered (score desc, row asc, ref asc); torch tensors on `self.device`, identical on every rank."""
        n = int(x.shape[0])
        n_loc = int(self.local.ntotal)

        def local_search(k_local):
            if n == 0 or n_loc == 0:
                z = torch.zeros(0, dtype=torch.int32, device=self.device)
                return z, z, torch.zeros(0, dtype=torch.float32, device=self.device), float("-inf")
            i, j, s, rad = self.local.global_topk(x, k_local, device_out=self.device.type == "cuda")
            if not isinstance(s, torch.Tensor):
                i, j, s = (torch.from_numpy(np.ascontiguousarray(a)).to(self.device) for a in (i, j, s))
            return i, j, s, rad

        if getattr(self.local, "metric_type", 0) != 0:
            raise NotImplementedError("RefShardedIndex.global_topk: inner-product indexes only (the matching pipeline's metric)")
        hi, hj, hs, tau, info = vdist.sharded_hits(local_search, n * n_loc, int(K), self.group, self.device,
                                                   k_local_start=k_local_start, ties="all", return_info=True)
        self.last_select = info
        # s_K == s_(K+1): the reference keeps the tied hits unless its schedule's final radius is that very score
        keep, self.last_matches_reference, self.last_ties_dropped = vdist.resolve_tie_on_cut(
            hs, tau, info,
            lambda: vdist.emulate_schedule_radius(lambda r0, r1, rad: self._range_scores(x[r0:r1], rad, 2 * int(K)),
                                                  n, int(K), self.group, self.device),
            os.environ.get("VSC_SHARD_TIE_RESOLVE", "1") != "0")
        if self.last_ties_dropped:
            hi, hj, hs = hi[keep], hj[keep], hs[keep]
        packed = torch.stack([hi.to(torch.int64), hj.to(torch.int64) + self.row0,
                              hs.contiguous().view(torch.int32).to(torch.int64)], dim=1)
        allp = vdist.all_gather_varlen(packed, self.group)
        s = allp[:, 2].to(torch.int32).view(torch.float32)
        if s.is_cuda and int(allp[:, 1].max().item() if allp.shape[0] else 0) < (1 << 31):
            # libvscmi's radix sorts (`vsc_sort_hits`: score desc, row asc, ref asc -- vsc/index.py:158-165)
            from vsc2022_amd.engine import sort_hits_device

            oi, oj, os_ = sort_hits_device(allp[:, 0], allp[:, 1], s, max_row=n)
            return oi[: int(K)], oj[: int(K)].to(torch.int64), os_[: int(K)], tau
        # (score desc, row asc, ref asc): stable sorts from the least significant key up (CPU tensors: the gloo tests)
        o = torch.sort(allp[:, 1], stable=True).indices
        o = o[torch.sort(allp[o, 0], stable=True).indices]
        o = o[torch.sort(-s[o].to(torch.float64), stable=True).indices][: int(K)]
        return allp[o, 0].to(torch.int32), allp[o, 1], s[o], tau
