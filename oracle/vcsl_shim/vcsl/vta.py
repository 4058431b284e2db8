"""networkx restatement of alipay/VCSL `vcsl/vta.py` Temporal-Network (TN) alignment.

ORACLE TOOLING ONLY (never imported by vsc2022_amd).

PARITY UNPINNED: the reference reaches this code through a dangling symlink
(/root/reference/vcsl/vta.py -> ../vcsl_module/vcsl/vta.py, empty un-pinned submodule,
.gitmodules:1-3); the real source cannot be consulted in this environment.  This file restates
the published algorithm (Tan et al., ACM MM'09, as packaged by VCSL) following SURVEY.md
Appendix B, and uses networkx 3.4.2 `dag_longest_path` -- the library the real code calls -- as
the DP.  It is the de-facto oracle for TN; the C restatement (vsc_oracle_tn.c) and the HIP kernel
are checked against it.  Call sites pinned by the reference: vsc/baseline/localization.py:44-46,58
(`build_vta_model(model_type, **kwargs).forward_sim([(name, sims)]) -> [(name, boxes)]`),
vsc/baseline/sscd_baseline.py:118-135 (kwargs), tests/test_localization.py:46-66 (properties).

Choices where Appendix B is ambiguous (kept identical in vsc_oracle_tn.c and tn.hip):
  * per-row top-k uses a STABLE argsort of -sims (ties -> lower ref index first);
  * constraint C3 is "every intermediate ref index is < r_src, or every one is > r_dst";
  * arithmetic follows numpy-2 promotion: everything stays in the dtype of `sims`.
"""
from typing import List, Sequence, Tuple

import networkx as nx
import numpy as np


def _iou(box: np.ndarray, boxes: np.ndarray) -> np.ndarray:
    """IoU of one (q1, r1, q2, r2) rectangle against accepted ones; area = dq * dr (no +1)."""
    lt = np.maximum(box[None, :2], boxes[:, :2])
    rb = np.minimum(box[None, 2:], boxes[:, 2:])
    wh = np.maximum(rb - lt, 0)
    inter = wh[:, 0] * wh[:, 1]
    area_a = (box[2] - box[0]) * (box[3] - box[1])
    area_b = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    return inter / (area_a + area_b - inter)


def tn(
    views: np.ndarray,
    tn_max_step: int = 10,
    tn_top_k: int = 5,
    max_path: int = 10,
    min_sim: float = 0.2,
    min_length: int = 5,
    max_iou: float = 0.3,
) -> List[List[int]]:
    views = np.asarray(views)
    n_q, n_r = views.shape
    top = min(tn_top_k, n_r)
    topk_idx = np.argsort(-views, axis=1, kind="stable")[:, :top]
    topk_sim = np.take_along_axis(views, topk_idx, axis=1)

    graph = nx.DiGraph()
    graph.add_node(0)  # source (-1, -1)
    id2pair = {0: (-1, -1)}
    pair2id = {(-1, -1): 0}
    node_num = 1
    for q in range(n_q):
        for k in range(top):
            pair = (q, int(topk_idx[q, k]))
            id2pair[node_num] = pair
            pair2id[pair] = node_num
            graph.add_node(node_num)
            node_num += 1

    for q_i in range(n_q):
        r_i = topk_idx[q_i]
        inter = np.empty((0,), dtype=np.int64)
        for q_j in range(q_i + 1, min(n_q, q_i + tn_max_step)):  # C1
            r_j = topk_idx[q_j]
            r_diff = r_j[:, None] - r_i[None, :]  # [dst rank, src rank]
            c2 = (r_diff > 0) & (r_diff < tn_max_step)
            if len(inter) == 0:
                c3 = np.ones(c2.shape, dtype=bool)
            else:
                src_ok = np.all(inter[None, :] < r_i[:, None], axis=1)[None, :]
                dst_ok = np.all(inter[None, :] > r_j[:, None], axis=1)[:, None]
                c3 = src_ok | dst_ok
            c4 = (topk_sim[q_j] >= min_sim)[:, None]
            rows, cols = np.where(c2 & c3 & c4)
            for b, a in zip(rows, cols):
                graph.add_edge(
                    pair2id[(q_i, int(r_i[a]))],
                    pair2id[(q_j, int(r_j[b]))],
                    weight=topk_sim[q_j, b],
                )
            inter = np.unique(np.concatenate([inter, r_j[rows].astype(np.int64)]))

    sink = node_num - 1
    for i in range(0, node_num - 1):
        p_i, p_j = id2pair[i], id2pair[sink]
        if (
            p_j[0] > p_i[0]
            and p_j[1] > p_i[1]
            and p_j[0] - p_i[0] <= tn_max_step
            and p_j[1] - p_i[1] <= tn_max_step
        ):
            graph.add_edge(i, sink, weight=0)

    boxes: List[List[int]] = []
    path = 0
    while True:
        if path > max_path:
            break
        longest = nx.dag_longest_path(graph)
        for i in range(1, len(longest)):
            graph.add_edge(longest[i - 1], longest[i], weight=0.0)
        if 0 in longest:
            longest.remove(0)
        if sink in longest:
            longest.remove(sink)
        path_q = [id2pair[n][0] for n in longest]
        path_r = [id2pair[n][1] for n in longest]
        if len(path_q) == 0:
            break
        score = 0.0
        for q, r in zip(path_q, path_r):
            score += views[q][r]
        if score > 0:
            q_min, q_max = min(path_q), max(path_q)
            r_min, r_max = min(path_r), max(path_r)
        else:
            q_min = q_max = r_min = r_max = 0
        ave_length = (r_max - r_min + q_max - q_min) / 2
        box = np.array([q_min, r_min, q_max, r_max], dtype=np.float32)
        ok = (
            ave_length != 0
            and score / ave_length > min_sim
            and min(r_max - r_min, q_max - q_min) > min_length
        )
        if ok and boxes:
            ious = _iou(box, np.array(boxes, dtype=np.float32))
            ok = bool(np.max(ious) < max_iou)
        if ok:
            boxes.append([int(q_min), int(r_min), int(q_max), int(r_max)])
        path += 1
    return boxes


class TN:
    def __init__(self, concurrency: int = 1, **config):
        self.config = config

    def forward_sim(
        self, data: Sequence[Tuple[str, np.ndarray]]
    ) -> List[Tuple[str, List[List[int]]]]:
        return [(name, tn(sims, **self.config)) for name, sims in data]


def build_vta_model(method: str = "TN", concurrency: int = 1, **config):
    if method != "TN":
        raise NotImplementedError(
            "oracle vcsl shim only restates TN (the only model the reference calls: "
            "sscd_baseline.py:121,131; dns_baseline.py:202)"
        )
    return TN(concurrency=concurrency, **config)
