"""The matching pipeline of `vsc.baseline.sscd_baseline` / `vsc.descriptor_eval_lib` over several GPUs of one node.

The reference gets every visible GPU without asking (`faiss.index_cpu_to_all_gpus`, /root/reference/vsc/index.py:153,171,
/root/reference/vsc/baseline/score_normalization.py:88-89); here the same entry points shard when they are started as N
ranks --

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        -m vsc2022_amd.vsc.baseline.sscd_baseline --query_features q.npz --ref_features r.npz --output_path out/ ...
    python -m torch.distributed.run ... -m vsc2022_amd.cli.descriptor_eval --query_features q.npz ...

-- one process per GPU (RCCL; gloo when the ranks share a GPU or `VSC_DIST_BACKEND=gloo`).  Every rank reads the
descriptor files, keeps ALL references and a contiguous range of the query videos (`dist.shard_ranges`), runs
`engine.DeviceMatcher.match` on its range -- the two global cuts of the pipeline and the tie-on-the-cut case are resolved in
`vsc2022_amd/dist.py`, so the candidate table IS the single-process one --, localises the candidate pairs whose query
video it owns, and the box table is gathered (`DeviceMatcher.gather_boxes`: one variable-length all-gather of 6 int64 per
box).  Rank 0 turns ordinals and frame indices into ids and timestamps and writes `candidates.csv` / `matches.csv` -- the
same bytes as the single-process run (`tests/test_gpu_config1.py::test_sharded_cli_writes_the_same_files`).
"""
import logging
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from vsc2022_amd.vsc import metrics as M
from vsc2022_amd.vsc.candidates import CandidateList
from vsc2022_amd.vsc.index import VideoFeature, VideoLayout

logger = logging.getLogger("sharded.py")
logger.setLevel(logging.INFO)


def requested() -> bool:
    """Was this process started as one of several ranks (torchrun / torch.distributed.run sets WORLD_SIZE)?"""
    return int(os.environ.get("WORLD_SIZE", "1")) > 1


def init() -> Tuple[int, int, int]:
    """Join the process group the launcher described; returns (rank, world, device index)."""
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        raise RuntimeError("vsc2022_amd: no MI355X (gfx950) device is visible -- there is no CPU fallback")
    device = local % n_dev
    torch.cuda.set_device(device)
    if not dist.is_initialized():
        backend = os.environ.get("VSC_DIST_BACKEND") or ("nccl" if n_dev >= int(os.environ.get("LOCAL_WORLD_SIZE", world)) else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
        logger.info("rank %d / %d on device %d, backend %s", rank, world, device, backend)
    return rank, world, device


def is_main() -> bool:
    import torch.distributed as dist

    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


def barrier():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def shard_of(videos: Sequence[VideoFeature]) -> Tuple[int, int]:
    """[begin, end) of the query videos this rank owns"""
    import torch.distributed as dist

    from vsc2022_amd import dist as vdist

    return vdist.shard_ranges(len(videos), dist.get_world_size())[dist.get_rank()]


def gather_videos(mine: List[VideoFeature], template: Sequence[VideoFeature]) -> List[VideoFeature]:
    """Every rank's adapted videos (e.g. score-normalised query descriptors) back as ONE list in the order of
    `template` (ranks own contiguous ranges): feature rows travel as one variable-length all-gather."""
    import dataclasses

    import torch

    from vsc2022_amd import dist as vdist

    import torch.distributed as dist

    # RCCL moves device memory only; gloo takes host tensors as they are
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    rows = VideoLayout.features(mine) if mine else np.zeros((0, template[0].feature.shape[1] if template else 1), np.float32)
    dim = torch.tensor([rows.shape[1] if len(mine) else 0], dtype=torch.int64, device=dev)
    # (ranks without a query video do not know the adapted dimension: take the largest over the ranks)
    dist.all_reduce(dim, op=dist.ReduceOp.MAX)
    d = int(dim.item())
    if rows.shape[1] != d:
        rows = np.zeros((0, d), dtype=np.float32)
    allrows = vdist.all_gather_varlen(torch.from_numpy(np.ascontiguousarray(rows, dtype=np.float32)).to(dev)).cpu().numpy()
    edges = np.cumsum([0] + [len(v) for v in template])
    assert edges[-1] == allrows.shape[0], (edges[-1], allrows.shape)
    return [dataclasses.replace(v, feature=allrows[a:b]) for v, a, b in zip(template, edges[:-1], edges[1:])]


def match_sharded(queries: List[VideoFeature], refs: List[VideoFeature], score_normalization: bool,
                  retrieve_per_query: float, candidates_per_query: float, localize_per_query: float,
                  tn_max_step: int, tn_min_length: int, bias: float, localize: bool = True
                  ) -> Tuple[CandidateList, Optional[List[M.Match]], object]:
    """(candidate pairs best first, Match rows or None, engine.MatchResult) -- identical on every rank.

    queries: ALL query videos (every rank slices its own range); what is searched is what the reference searches
    (vsc/baseline/sscd_baseline.py:90-104), what is localised follows its two variants (:118-135): with score
    normalisation the adapted descriptors themselves with `similarity_bias`, box score = MaxSim; without it the
    L2-normalised copies, box score = the candidate's score."""
    import torch
    import torch.distributed as dist

    from vsc2022_amd import dist as vdist
    from vsc2022_amd import engine
    from vsc2022_amd.vsc.baseline.score_normalization import normalize

    rank, world = dist.get_rank(), dist.get_world_size()
    if (engine.RETRIEVE_PER_QUERY, engine.CANDIDATES_PER_QUERY, engine.LOCALIZE_PER_QUERY) != (
            retrieve_per_query, candidates_per_query, localize_per_query) or \
            engine.REFERENCE_TN != dict(tn_max_step=tn_max_step, min_length=tn_min_length):
        raise ValueError("the sharded pipeline runs with the reference's constants (1200 / 25 / 5 per query video, TN(5, 4))")
    if len(queries) < world:
        raise ValueError(f"{len(queries)} query videos cannot be sharded over {world} ranks: start fewer ranks")
    device = torch.cuda.current_device()
    lo, hi = vdist.shard_ranges(len(queries), world)[rank]
    mine = list(queries[lo:hi])
    r_layout, q_layout_all = VideoLayout(refs), VideoLayout(queries)
    r_rows = VideoLayout.features(refs)
    dim = r_rows.shape[1]
    q_rows = VideoLayout.features(mine) if mine else np.zeros((0, dim), dtype=np.float32)
    q_off = np.r_[0, np.cumsum([len(v) for v in mine])].astype(np.int64)
    row_base = int(q_layout_all.offsets[lo])
    tn_r = tn_q = None
    if localize and not score_normalization:
        tn_r = normalize(r_rows, device)
        tn_q = normalize(q_rows, device) if len(q_rows) else q_rows
    m = engine.DeviceMatcher(r_rows, r_layout.offsets, device, tn_ref_feats=tn_r)
    m.set_queries(q_rows, q_off, tn_q_feats=tn_q)
    res = m.match(n_qvid_global=len(queries), qvid_base=lo, row_base=row_base, bias=bias if score_normalization else 0.0,
                  localize=localize)
    cq, cr = res.cand_q.cpu().numpy(), res.cand_r.cpu().numpy()
    cs = res.cand_score.cpu().numpy()
    pairs = CandidateList(cq.astype(np.int32), cr.astype(np.int32), cs, q_layout_all.video_ids, r_layout.video_ids)
    if not localize:
        return pairs, None, res
    boxes = m.gather_boxes(res).cpu().numpy()
    matches: List[M.Match] = []
    for k, x1, y1, x2, y2, bits in boxes:
        q, r = queries[cq[k]], refs[cr[k]]
        score = np.array([bits], dtype=np.int64).astype(np.int32).view(np.float32)[0] if score_normalization else cs[k]
        matches.append(M.Match(query_id=q.video_id, ref_id=r.video_id, score=score,
                               query_start=q.get_timestamps(int(x1))[0], query_end=q.get_timestamps(int(x2))[1],
                               ref_start=r.get_timestamps(int(y1))[0], ref_end=r.get_timestamps(int(y2))[1]))
    return pairs, matches, res
