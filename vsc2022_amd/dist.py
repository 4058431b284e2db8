"""Query-sharded multi-GPU matching: one process per GPU, `torch.distributed` (backend "nccl" is
RCCL on ROCm; the CPU tests use "gloo").

Sharding: every rank holds the whole reference set and a contiguous range
of query videos.  Rows of the similarity matrix are independent and so are candidate pairs, so the
only exchange steps are the ones forced by the GLOBAL cuts of the reference pipeline:

  1. the K = 1200 * n_query_videos best frame hits over ALL queries (vsc/index.py:142-165):
     exact distributed selection over the per-rank score-sorted hit lists -- two all-reduces of a
     65536-bin histogram of the order-preserving fp32 key (512 KiB each) + one all-gather of tie
     counts;
  2. the 25 * n_query_videos best (query, ref) candidates (vsc/descriptor_eval_lib.py:45-49,
     vsc/baseline/sscd_baseline.py:101-102): the same selection over the per-rank candidate
     lists, then a variable-length all-gather of the surviving candidates (20 B each).

Messages are KBs to a few MBs, i.e. latency-bound on the direct xGMI links; no ring-sized
buckets are needed.  Everything here works on torch tensors of any device, so the same code runs
under gloo on CPU tensors in the tests (with the per-rank search results supplied by the oracle)
and under RCCL on HBM tensors in production.

Result contract: identical to the single-GPU result whenever the reference's own batch schedule
does not drop hits tied with a re-threshold radius (vsc/index.py semantics are schedule-dependent
only in that case); i.e. the sharded search returns the exact global top-K under the
total order (score desc, query row asc, ref row asc).
"""
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def score_keys(scores: torch.Tensor) -> torch.Tensor:
    """Order-preserving fp32 -> integer key (int64 holding a uint32): larger score, larger key."""
    bits = scores.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    neg = (bits & 0x80000000) != 0
    return torch.where(neg, (~bits) & 0xFFFFFFFF, bits | 0x80000000)


def key_to_score(key: int) -> float:
    """Inverse of score_keys for one key."""
    import numpy as np

    bits = (key & 0x7FFFFFFF) if (key & 0x80000000) else ((~key) & 0xFFFFFFFF)
    return float(np.array([bits], dtype=np.uint32).view(np.float32)[0])


def _world(group):
    if not dist.is_available() or not dist.is_initialized():
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def _via_host(t: torch.Tensor, group) -> bool:
    """gloo only implements a subset of the collectives for device tensors: stage through the host
    (used by the tests and by single-GPU multi-process debugging; RCCL takes device tensors as is)."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _all_reduce_sum(t: torch.Tensor, group) -> torch.Tensor:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if _via_host(t, group):
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def all_reduce_max_int(v: int, device, group=None) -> int:
    rank, world = _world(group)
    if world == 1:
        return int(v)
    t = torch.tensor([int(v)], dtype=torch.int64, device=device)
    if _via_host(t, group):
        t = t.cpu()
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


def all_reduce_sum_int(v: int, device, group=None) -> int:
    rank, world = _world(group)
    if world == 1:
        return int(v)
    t = torch.tensor([int(v)], dtype=torch.int64, device=device)
    if _via_host(t, group):
        t = t.cpu()
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t.item())


def _all_gather_scalar(v: int, device, group) -> List[int]:
    rank, world = _world(group)
    if world == 1:
        return [int(v)]
    mine = torch.tensor([int(v)], dtype=torch.int64, device=device)
    if _via_host(mine, group):
        mine = mine.cpu()
    out = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(out, mine, group=group)
    return [int(x.item()) for x in out]


def _all_gather_vec(v: torch.Tensor, group) -> torch.Tensor:
    """[world, len(v)] from every rank's small 1-D int64 vector (one collective, no host sync unless gloo)."""
    rank, world = _world(group)
    if world == 1:
        return v.unsqueeze(0)
    host = _via_host(v, group)
    send = v.cpu() if host else v
    out = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(out, send, group=group)
    return torch.stack(out).to(v.device)


def distributed_prefix_select(sorted_scores: torch.Tensor, k_total: int, group=None, ties: str = "rank"
                              ) -> Tuple[int, float]:
    """How many leading elements of this rank's score-DESCENDING list belong to the global top
    `k_total` of the union of all ranks' lists.  Returns (n_take, tau) with tau the k_total-th best
    score (-inf when the union is shorter than k_total).

    ties="rank": total order (score desc, rank asc, local position asc) -- exactly k_total elements are
                 taken over all ranks (query-sharded search: rank order IS query-row order).
    ties="all" : every element tied with tau is taken (reference-sharded search: the caller orders the
                 ties by (query row, ref row) after gathering them and cuts at k_total).

    Three collectives (all-reduce of a 65537-entry histogram that also carries the list lengths,
    all-reduce of the second-level histogram, all-gather of the tie counts) and ONE host sync at the end:
    everything in between stays on the device.
    """
    rank, world = _world(group)
    device = sorted_scores.device
    n_local = int(sorted_scores.numel())
    if k_total <= 0:
        return 0, float("inf")
    # The list is sorted, and a rank can contribute at most k_total elements: everything below is
    # O(65536 log n) searchsorted calls on the (ascending copy of the) leading k_total keys -- no pass
    # over the list besides the key conversion (a full histogram of 10 M scores cost 70 ms per call).
    lead = sorted_scores[: min(n_local, k_total)]
    n_lead = int(lead.numel())
    asc = score_keys(lead).flip(0).contiguous() if n_lead else torch.zeros(0, dtype=torch.int64, device=device)
    steps = torch.arange(65537, dtype=torch.int64, device=device)

    def level(base: torch.Tensor, shift: int) -> torch.Tensor:
        """65536-bin histogram of the keys in [base, base + 65536 << shift), bins of width 1 << shift"""
        ge = n_lead - torch.searchsorted(asc, base + (steps << shift))
        return (ge[:-1] - ge[1:]).to(torch.int64)

    def pick(hist: torch.Tensor, need: torch.Tensor):
        """walk from the top bin: (bin, elements above it) of the first bin whose cumulative count reaches `need`"""
        cum = torch.cumsum(hist.flip(0), 0)
        pos = torch.searchsorted(cum, need.reshape(1)).clamp(max=65535)[0]
        above = torch.where(pos > 0, cum[(pos - 1).clamp(min=0)], torch.zeros_like(pos))
        return 65535 - pos, above

    zero = torch.zeros((), dtype=torch.int64, device=device)
    k_t = torch.tensor(k_total, dtype=torch.int64, device=device)
    h1 = _all_reduce_sum(torch.cat([level(zero, 16), torch.tensor([n_local], dtype=torch.int64, device=device)]), group)
    total = h1[65536]
    b1, above1 = pick(h1[:65536], k_t)
    h2 = _all_reduce_sum(level(b1 << 16, 0), group)
    b2, above2 = pick(h2, k_t - above1)
    tau_key = (b1 << 16) | b2
    edge = n_lead - torch.searchsorted(asc, torch.stack([tau_key, tau_key + 1]))
    n_gt, n_eq = edge[1], edge[0] - edge[1]
    eq_all = _all_gather_vec(n_eq.reshape(1), group)[:, 0]
    before = eq_all[:rank].sum()
    m = k_t - (above1 + above2)  # elements tied with tau that still fit
    take_ties = n_eq if ties == "all" else torch.minimum(n_eq, (m - before).clamp(min=0))
    res = torch.stack([n_gt + take_ties, tau_key, total]).cpu()  # the one host sync
    n_take, tau_key_h, total_h = (int(x) for x in res)
    if total_h <= k_total:
        return n_local, float("-inf")
    return n_take, float(key_to_score(tau_key_h))


def all_gather_varlen(t: torch.Tensor, group=None) -> torch.Tensor:
    """Concatenate every rank's 1-D/2-D tensor (different lengths along dim 0), rank order."""
    rank, world = _world(group)
    if world == 1:
        return t
    lens = _all_gather_scalar(int(t.shape[0]), t.device, group)
    mx = max(lens)
    pad_shape = (mx,) + tuple(t.shape[1:])
    host = _via_host(t, group)
    padded = torch.zeros(pad_shape, dtype=t.dtype, device="cpu" if host else t.device)
    padded[: t.shape[0]] = t
    outs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(outs, padded, group=group)
    return torch.cat([o[:n] for o, n in zip(outs, lens)], dim=0).to(t.device)


class ShardedCandidates:
    """Globally ordered candidate table (every rank holds the same copy)."""

    def __init__(self, q_vid, r_vid, score, first_i, first_j):
        self.q_vid, self.r_vid, self.score = q_vid, r_vid, score
        self.first_i, self.first_j = first_i, first_j

    def __len__(self):
        return int(self.score.numel())


def merge_hits(local_scores_sorted: torch.Tensor, k_global: int, complete_above: float, group=None,
               ties: str = "rank") -> Tuple[int, float, bool]:
    """Step 1: prefix of the local hit list that survives the global K cut.

    `complete_above`: every local hit with score > complete_above is present in the local list
    (the local search's own cut).  Returns (n_take, tau, exact): exact is False on ranks whose
    local cut is not strictly below the global one -- the caller must then rerun that rank's
    local search with a larger local K (all ranks call merge_hits again).
    """
    n_take, tau = distributed_prefix_select(local_scores_sorted, k_global, group, ties)
    exact = (complete_above == float("-inf")) or (tau > complete_above)
    return n_take, tau, exact


def sharded_hits(local_search, local_matrix_size: int, k_global: int, group=None, device=None,
                 k_local_start: Optional[int] = None, ties: str = "rank"):
    """Exact global top-`k_global` hits from per-rank searches.

    local_search(k_local) -> (i, j, s, radius): this rank's global-threshold search with budget
    k_local (hits sorted by (score desc, row asc, ref asc); radius = the search's final radius).
    local_matrix_size = rows x columns of this rank's part of the score matrix.  Starts from k_local =
    1.25*K/world and doubles the budget of any rank whose own cut is not strictly below the global cut
    (skewed shards), until the result is exact everywhere.  A rank that already is exact keeps its
    result across the retries (its list stays complete above a cut that can only rise) and only takes
    part in the collectives; a rank whose budget has reached K is exact by construction (nothing beyond
    its K best can be among the global K best).  Every rank leaves the loop on the same, all-reduced
    flag.  Returns (i, j, s, tau) = this rank's share of the global top-K (ties="all": including every
    hit tied with tau, see distributed_prefix_select).

    local_search may return a fifth element, True for a SEEDED search (engine.DeviceMatcher.seed_radius: the rank
    searched with the full budget k_global from a radius agreed over a row sample instead of replaying the
    reference's doubling schedule; k_local is ignored).  Its list holds every local hit beyond the returned radius,
    cut at k_global: exact when it is full (k_global hits: nothing beyond a rank's K best can be among the global K
    best) or when the global cut lies strictly beyond that radius; otherwise -- the seed was too high -- the rank
    retries, and local_search is expected to answer the retry with the unseeded search.
    """
    rank, world = _world(group)
    # A rank's share of the global top-K is K/world up to sampling noise when the shards are alike; the
    # local search costs more the larger its budget (re-scoring, selection and sorting scale with it), so
    # start 25 % above the even share and let the doubling below handle skewed shards.
    k_local = max(1, min(k_global, (5 * k_global) // (4 * max(world, 1)) + 1))
    if k_local_start is not None:
        k_local = max(1, min(k_global, int(k_local_start)))
    cached = None
    while True:
        if cached is None:
            res = local_search(k_local)
            hi, hj, hs, radius = res[:4]
            seeded = len(res) > 4 and bool(res[4])
            n = int(hs.numel())
            full = False
            if n >= local_matrix_size:
                complete_above = float("-inf")      # the whole local matrix was kept
            elif seeded:
                complete_above = float(radius)      # every local hit beyond the radius is listed (cut at k_global)
                full = n >= k_global
            elif n >= k_local:
                complete_above = float(hs[-1].item())  # truncated at k_local: ties with the last may be missing
            else:
                complete_above = float(radius)      # hits <= radius were dropped by the schedule
            cached = (hi, hj, hs, complete_above, seeded, full)
        hi, hj, hs, complete_above, seeded, full = cached
        n_take, tau, exact = merge_hits(hs, k_global, complete_above, group, ties)
        exact = exact or full or (k_local >= k_global and not seeded)
        all_exact = all_reduce_max_int(0 if exact else 1, hs.device if device is None else device, group) == 0
        if all_exact:
            return hi[:n_take], hj[:n_take], hs[:n_take], tau
        if not exact:
            k_local = min(k_global, k_local * 2)
            cached = None


def merge_candidates(q_vid: torch.Tensor, r_vid: torch.Tensor, score: torch.Tensor, first_i: torch.Tensor,
                     first_j: torch.Tensor, m_global: int, group=None) -> ShardedCandidates:
    """Step 2: global top-`m_global` candidates, ordered as the single-process pipeline orders them:
    score descending, ties by the (global query row, ref row) of the pair's first hit.

    Inputs are this rank's candidates in local order (score desc, first-appearance); q_vid and
    first_i must already be GLOBAL ordinals/rows.  Ranks own ascending, disjoint query ranges, so
    local order + rank order is the global tie order.
    """
    n_take, _ = distributed_prefix_select(score, m_global, group)
    packed = torch.stack(
        [q_vid[:n_take].to(torch.int64), r_vid[:n_take].to(torch.int64),
         score[:n_take].contiguous().view(torch.int32).to(torch.int64),
         first_i[:n_take].to(torch.int64), first_j[:n_take].to(torch.int64)], dim=1)
    allp = all_gather_varlen(packed, group)
    s = allp[:, 2].to(torch.int32).view(torch.float32)
    # stable sort by score desc over (rank-ordered) concatenation == (score desc, first_i, first_j)
    order = torch.sort(-s.to(torch.float64), stable=True).indices
    allp = allp[order]
    return ShardedCandidates(allp[:, 0].to(torch.int32), allp[:, 1].to(torch.int32),
                             allp[:, 2].to(torch.int32).view(torch.float32), allp[:, 3], allp[:, 4])


def shard_ranges(n_items: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, balanced [begin, end) ranges (rank order)."""
    return [((n_items * r) // world, (n_items * (r + 1)) // world) for r in range(world)]


def ref_sharded_knn(local_scores: torch.Tensor, local_ids: torch.Tensor, k: int, group=None
                    ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Reference-sharded k-NN (BASELINE config 5): every rank holds a slice of
    the reference set and has searched ALL queries against it.

    local_scores [nq, k] fp32 and local_ids [nq, k] int64 are this rank's per-row top-k with GLOBAL
    reference ids (local id + the shard's row offset; -1 marks an empty slot).  One all-gather of the
    [nq, k] pairs (12 B per entry: 48 MB per rank at nq = 200k, k = 20, i.e. ~0.3 ms per xGMI link),
    then every rank merges the world*k candidates of each row by (score desc, id asc) -- the order a
    single index over the concatenated reference set produces.
    """
    rank, world = _world(group)
    if world == 1:
        return local_scores, local_ids
    nq = local_scores.shape[0]
    packed = torch.stack([local_scores.contiguous().view(torch.int32).to(torch.int64), local_ids.to(torch.int64)],
                         dim=2)  # [nq, k, 2]
    host = _via_host(packed, group)
    send = packed.cpu() if host else packed
    outs = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(outs, send, group=group)
    allp = torch.cat(outs, dim=1).to(local_scores.device)  # [nq, world*k, 2]
    s = allp[..., 0].to(torch.int32).view(torch.float32)
    ids = allp[..., 1]
    empty = ids < 0
    # sort by (score desc, id asc): stable sort by id first, then stable by score
    big = torch.iinfo(torch.int64).max
    o1 = torch.sort(torch.where(empty, torch.full_like(ids, big), ids), dim=1, stable=True).indices
    s1 = torch.gather(s, 1, o1)
    i1 = torch.gather(ids, 1, o1)
    key = torch.where(i1 < 0, torch.full_like(s1, float("-inf")), s1).to(torch.float64)
    o2 = torch.sort(-key, dim=1, stable=True).indices[:, :k]
    return torch.gather(s1, 1, o2), torch.gather(i1, 1, o2)
