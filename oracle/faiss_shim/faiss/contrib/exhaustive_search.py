"""numpy restatement of the two faiss.contrib.exhaustive_search helpers that
vsc/index.py:147-154 calls (third-party faiss ~1.7.x; semantics per SURVEY.md Appendix A).
ORACLE TOOLING ONLY."""
import numpy as np

from faiss import METRIC_INNER_PRODUCT


def exponential_query_iterator(xq, start_bs=32, max_bs=20000):
    nq = len(xq)
    bs = start_bs
    i = 0
    while i < nq:
        xqi = xq[i : i + bs]
        yield xqi
        if bs < max_bs:
            bs *= 2
        i += len(xqi)


def _threshold(nres, dis, ids, thresh, keep_max):
    mask = dis > thresh if keep_max else dis < thresh
    new_nres = np.zeros_like(nres)
    o = 0
    for i, nr in enumerate(nres):
        nr = int(nr)
        new_nres[i] = mask[o : o + nr].sum()
        o += nr
    return new_nres, dis[mask], ids[mask]


def range_search_max_results(index, query_iterator, radius, max_results=None, min_results=None,
                             shard=False, ngpu=0, clip_to_min=False):
    if min_results is None:
        min_results = int(0.8 * max_results)
    if max_results is None:
        max_results = int(min_results * 1.5)
    keep_max = index.metric_type == METRIC_INNER_PRODUCT
    batches = []
    totres = 0
    for xqi in query_iterator:
        lims_i, Di, Ii = index.range_search(xqi, radius)
        nres_i = (lims_i[1:] - lims_i[:-1]).astype(np.int64)
        batches.append((nres_i, Di, Ii))
        totres += len(Di)
        if max_results is not None and totres > max_results:
            alldis = np.hstack([d for _, d, _ in batches])
            if keep_max:
                alldis.partition(len(alldis) - min_results - 1)
                radius = float(alldis[-1 - min_results])
            else:
                alldis.partition(min_results)
                radius = float(alldis[min_results])
            totres = 0
            for b, (nres, dis, ids) in enumerate(batches):
                nres, dis, ids = _threshold(nres, dis, ids, radius, keep_max)
                totres += len(dis)
                batches[b] = (nres, dis, ids)
    nres = np.hstack([b[0] for b in batches]) if batches else np.zeros(0, dtype=np.int64)
    lims = np.zeros(len(nres) + 1, dtype="uint64")
    lims[1:] = np.cumsum(nres)
    D = np.hstack([b[1] for b in batches]) if batches else np.zeros(0, dtype=np.float32)
    I = np.hstack([b[2] for b in batches]) if batches else np.zeros(0, dtype=np.int64)
    return radius, lims, D, I
