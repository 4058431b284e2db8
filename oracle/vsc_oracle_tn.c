/*
 * vsc_oracle_tn.c -- CPU ORACLE (test infrastructure, NOT product code): Temporal-Network
 * localisation of one candidate pair, plus the per-pair similarity tile.
 *
 * Reference call sites (paths relative to /root/reference):
 *   vsc/baseline/localization.py:36,52-54  sims = q.feature @ r.feature.T + similarity_bias
 *   vsc/baseline/localization.py:58        model.forward_sim(sims)  -> vcsl.vta `tn`
 *   vsc/baseline/localization.py:66-73     box = (q_lo, r_lo, q_hi, r_hi), frame indices, inclusive
 *
 * PARITY UNPINNED for TN: vcsl/vta.py is a dangling symlink into an empty, un-pinned submodule
 * (alipay/VCSL; .gitmodules:1-3), so the third-party source cannot be read here.  This file
 * restates SURVEY.md Appendix B in an "implicit graph" form (no edge lists) that the HIP kernel
 * mirrors; it is validated against the networkx restatement oracle/vcsl_shim/vcsl/vta.py, which
 * runs the real networkx 3.4.2 dag_longest_path (the DP the third-party code calls).
 *
 * networkx semantics reproduced exactly (networkx/algorithms/dag.py, v3.4.2):
 *   - topological_sort = Kahn generations; zero in-degree nodes in node insertion order; children
 *     discovered in adjacency insertion order;
 *   - dist[v] = FIRST maximal (dist[u] + w(u,v)) over G.pred[v] in insertion order; no preds or a
 *     negative best -> (0, v);
 *   - end node = FIRST node in topological order with maximal dist; back-track to a self-parent.
 * All arithmetic is fp32 (numpy-2 promotion keeps np.float32 through the whole computation).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t tn_max_step;
    int32_t tn_top_k;
    int32_t max_path;
    int32_t min_length;
    float min_sim;
    float max_iou;
} orc_tn_params;

/* sims[lq*lr] = fma chain over k of qf[q][k]*rf[r][k], then + bias (one fp32 add). */
void orc_pair_sims(const float *qf, int64_t lq, const float *rf, int64_t lr, int64_t d, float bias,
                   float *out) {
    for (int64_t q = 0; q < lq; ++q)
        for (int64_t r = 0; r < lr; ++r) {
            float acc = 0.0f;
            for (int64_t k = 0; k < d; ++k) acc = fmaf(qf[q * d + k], rf[r * d + k], acc);
            out[q * lr + r] = acc + bias;
        }
}

typedef struct {
    int lq, lr, top, ms, n_nodes, sink;
    const int32_t *tidx; /* [lq][top] ref index of the k-th best cell of row q */
    const float *tsim;   /* [lq][top] */
    const int32_t *ilo;  /* [lq][ms] min of intermediates of q_i before step d (index d) */
    const int32_t *ihi;  /* [lq][ms] max, or ilo > ihi when the set is empty */
    float min_sim;
    int sink_q, sink_r;
} tn_graph;

static inline int node_q(const tn_graph *g, int v) { return v == 0 ? -1 : (v - 1) / g->top; }
static inline int node_k(const tn_graph *g, int v) { return (v - 1) % g->top; }
static inline int node_r(const tn_graph *g, int v) {
    return v == 0 ? -1 : g->tidx[(v - 1)];
}

/* regular edge (q_i, a) -> (q_i + d, b)?  1 <= d < ms, q_i + d < lq */
static inline int edge_ok(const tn_graph *g, int qi, int a, int d, int b) {
    const int qj = qi + d;
    const int ra = g->tidx[qi * g->top + a], rb = g->tidx[qj * g->top + b];
    const int rd = rb - ra;
    if (!(rd > 0 && rd < g->ms)) return 0; /* C2 */
    const int lo = g->ilo[qi * g->ms + d], hi = g->ihi[qi * g->ms + d];
    if (lo <= hi) { /* non-empty intermediate set: C3 */
        if (!(hi < ra || lo > rb)) return 0;
    }
    return g->tsim[qj * g->top + b] >= g->min_sim; /* C4 */
}

/* node u (any, incl. source) -> sink by the sink rule? */
static inline int sink_ok(const tn_graph *g, int u) {
    if (u == g->sink) return 0;
    const int qu = node_q(g, u), ru = node_r(g, u);
    return g->sink_q > qu && g->sink_r > ru && g->sink_q - qu <= g->ms && g->sink_r - ru <= g->ms;
}

/* bit index of the "weight zeroed" flag of regular edge (qi,a) -> (qi+d,b) */
static inline int64_t edge_bit(const tn_graph *g, int qi, int a, int d, int b) {
    return (((int64_t)(qi + d) * g->top + b) * g->ms + d) * g->top + a;
}

/*
 * vcsl.vta.tn on one similarity matrix.  boxes: [max_boxes][4] = q_min, r_min, q_max, r_max.
 * Returns the number of accepted boxes (>= 0) or < 0 on allocation failure / too many boxes.
 */
int64_t orc_tn(const float *sims, int64_t lq64, int64_t lr64, const orc_tn_params *p, int32_t *boxes,
               int64_t max_boxes) {
    const int lq = (int)lq64, lr = (int)lr64;
    const int ms = p->tn_max_step;
    const int top = p->tn_top_k < lr ? p->tn_top_k : lr;
    if (lq <= 0 || top <= 0) return 0;
    const int n_nodes = 1 + lq * top;
    const int msz = ms > 1 ? ms : 1;

    int32_t *tidx = (int32_t *)malloc(sizeof(int32_t) * (size_t)lq * top);
    float *tsim = (float *)malloc(sizeof(float) * (size_t)lq * top);
    int32_t *ilo = (int32_t *)malloc(sizeof(int32_t) * (size_t)lq * msz);
    int32_t *ihi = (int32_t *)malloc(sizeof(int32_t) * (size_t)lq * msz);
    const int64_t nbits = (int64_t)lq * top * msz * top;
    uint8_t *zero = (uint8_t *)calloc((size_t)(nbits / 8 + 1), 1);
    int32_t *indeg = (int32_t *)calloc((size_t)n_nodes, sizeof(int32_t));
    int32_t *order = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_nodes); /* topo order */
    float *dist = (float *)malloc(sizeof(float) * (size_t)n_nodes);
    int32_t *par = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_nodes);
    int32_t *path = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_nodes);
    int64_t nbox = 0;
    if (!tidx || !tsim || !ilo || !ihi || !zero || !indeg || !order || !dist || !par || !path) {
        nbox = -1;
        goto done;
    }

    /* 1. per-row top-k by (sim desc, ref asc)  == stable argsort(-sims)[:, :top] */
    for (int q = 0; q < lq; ++q) {
        const float *row = sims + (int64_t)q * lr;
        int m = 0;
        for (int r = 0; r < lr; ++r) {
            const float s = row[r];
            if (m == top && !(s > tsim[q * top + top - 1])) continue;
            int pos = m < top ? m : top - 1;
            while (pos > 0 && s > tsim[q * top + pos - 1]) {
                tsim[q * top + pos] = tsim[q * top + pos - 1];
                tidx[q * top + pos] = tidx[q * top + pos - 1];
                --pos;
            }
            tsim[q * top + pos] = s;
            tidx[q * top + pos] = r;
            if (m < top) ++m;
        }
    }

    tn_graph g;
    g.lq = lq; g.lr = lr; g.top = top; g.ms = msz; g.n_nodes = n_nodes; g.sink = n_nodes - 1;
    g.tidx = tidx; g.tsim = tsim; g.ilo = ilo; g.ihi = ihi; g.min_sim = p->min_sim;
    g.sink_q = lq - 1;
    g.sink_r = tidx[(lq - 1) * top + top - 1];

    /* 2. intermediate sets: before step d (q_j = q_i + d) the set holds every dst ref that got an
     * edge from q_i at a nearer step; only its min and max matter for C3. */
    for (int qi = 0; qi < lq; ++qi) {
        int lo = 1, hi = 0; /* empty */
        for (int d = 1; d < ms; ++d) {
            ilo[qi * msz + d] = lo;
            ihi[qi * msz + d] = hi;
            if (qi + d >= lq) continue;
            int nlo = lo, nhi = hi;
            for (int b = 0; b < top; ++b) {
                int any = 0;
                for (int a = 0; a < top && !any; ++a) any = edge_ok(&g, qi, a, d, b);
                if (any) {
                    const int rb = tidx[(qi + d) * top + b];
                    if (nlo > nhi) nlo = nhi = rb;
                    else {
                        if (rb < nlo) nlo = rb;
                        if (rb > nhi) nhi = rb;
                    }
                }
            }
            lo = nlo;
            hi = nhi;
        }
    }

    /* 3. Kahn generations.  Neighbours of u in adjacency insertion order: regular edges by
     * (d asc, b asc), then the sink edge unless the sink is already a regular neighbour. */
    for (int pass = 0; pass < 2; ++pass) {
        int head = 0, tail = 0;
        if (pass == 1)
            for (int v = 0; v < n_nodes; ++v)
                if (indeg[v] == 0) order[tail++] = v;
        /* pass 0 enumerates every node once to count in-degrees; pass 1 runs the queue */
        const int count = (pass == 0) ? n_nodes : 0;
        int idx = 0;
        for (;;) {
            int u;
            if (pass == 0) {
                if (idx >= count) break;
                u = idx++;
            } else {
                if (head >= tail) break;
                u = order[head++];
            }
            int to_sink_regular = 0;
            if (u != 0) {
                const int qi = node_q(&g, u), a = node_k(&g, u);
                for (int d = 1; d < ms && qi + d < lq; ++d)
                    for (int b = 0; b < top; ++b)
                        if (edge_ok(&g, qi, a, d, b)) {
                            const int v = 1 + (qi + d) * top + b;
                            if (v == g.sink) to_sink_regular = 1;
                            if (pass == 0) ++indeg[v];
                            else if (--indeg[v] == 0) order[tail++] = v;
                        }
            }
            if (!to_sink_regular && sink_ok(&g, u)) {
                if (pass == 0) ++indeg[g.sink];
                else if (--indeg[g.sink] == 0) order[tail++] = g.sink;
            }
        }
        if (pass == 1 && tail != n_nodes) { /* cannot happen: the graph is a DAG */
            nbox = -3;
            goto done;
        }
    }

    /* 4. up to max_path+1 longest-path extractions */
    for (int it = 0; it <= p->max_path; ++it) {
        /* DP in topological order */
        for (int t = 0; t < n_nodes; ++t) {
            const int v = order[t];
            float best = 0.0f;
            int arg = -1;
            if (v != 0) {
                const int qj = node_q(&g, v), b = node_k(&g, v);
                /* regular preds in insertion order: q_i asc (d desc), a asc */
                for (int d = ms - 1; d >= 1; --d) {
                    const int qi = qj - d;
                    if (qi < 0) continue;
                    for (int a = 0; a < top; ++a)
                        if (edge_ok(&g, qi, a, d, b)) {
                            const int u = 1 + qi * top + a;
                            const int64_t bit = edge_bit(&g, qi, a, d, b);
                            const int z = (zero[bit >> 3] >> (bit & 7)) & 1;
                            const float w = (z || v == g.sink) ? 0.0f : tsim[qj * top + b];
                            const float c = dist[u] + w;
                            if (arg < 0 || c > best) {
                                best = c;
                                arg = u;
                            }
                        }
                }
                if (v == g.sink) {
                    /* sink-rule preds not already regular, in node id order, weight 0 */
                    for (int u = 0; u < n_nodes - 1; ++u) {
                        if (!sink_ok(&g, u)) continue;
                        int regular = 0;
                        if (u != 0) {
                            const int qi = node_q(&g, u), a = node_k(&g, u);
                            const int d = qj - qi;
                            if (d >= 1 && d < ms) regular = edge_ok(&g, qi, a, d, b);
                        }
                        if (regular) continue;
                        const float c = dist[u] + 0.0f;
                        if (arg < 0 || c > best) {
                            best = c;
                            arg = u;
                        }
                    }
                }
            }
            if (arg < 0 || !(best >= 0.0f)) {
                dist[v] = 0.0f;
                par[v] = v;
            } else {
                dist[v] = best;
                par[v] = arg;
            }
        }
        /* end node: first in topological order with maximal dist */
        int vend = order[0];
        for (int t = 1; t < n_nodes; ++t)
            if (dist[order[t]] > dist[vend]) vend = order[t];
        int plen = 0;
        for (int v = vend;; v = par[v]) {
            path[plen++] = v;
            if (par[v] == v) break;
        }
        /* path[] is reversed (end -> start); zero the weights along it */
        for (int x = plen - 1; x >= 1; --x) {
            const int u = path[x], v = path[x - 1];
            if (u != 0) {
                const int qi = node_q(&g, u), a = node_k(&g, u);
                const int qj = node_q(&g, v), b = node_k(&g, v);
                const int d = qj - qi;
                if (d >= 1 && d < ms && edge_ok(&g, qi, a, d, b)) {
                    const int64_t bit = edge_bit(&g, qi, a, d, b);
                    zero[bit >> 3] |= (uint8_t)(1u << (bit & 7));
                }
            }
        }
        /* drop source and sink, walk in forward order */
        float score = 0.0f;
        int qmin = 0, qmax = 0, rmin = 0, rmax = 0, cnt = 0;
        for (int x = plen - 1; x >= 0; --x) {
            const int v = path[x];
            if (v == 0 || v == g.sink) continue;
            const int q = node_q(&g, v), r = node_r(&g, v);
            score += tsim[v - 1];
            if (cnt == 0) {
                qmin = qmax = q;
                rmin = rmax = r;
            } else {
                if (q < qmin) qmin = q;
                if (q > qmax) qmax = q;
                if (r < rmin) rmin = r;
                if (r > rmax) rmax = r;
            }
            ++cnt;
        }
        if (cnt == 0) break;
        if (!(score > 0.0f)) qmin = qmax = rmin = rmax = 0;
        const int dq = qmax - qmin, dr = rmax - rmin;
        const float ave = (float)(dr + dq) / 2.0f;
        int ok = (ave != 0.0f) && (score / ave > p->min_sim) && ((dr < dq ? dr : dq) > p->min_length);
        if (ok && nbox > 0) {
            float mx = -INFINITY;
            for (int64_t k = 0; k < nbox; ++k) {
                const int32_t *o = boxes + 4 * k;
                const float lt0 = (float)(qmin > o[0] ? qmin : o[0]);
                const float lt1 = (float)(rmin > o[1] ? rmin : o[1]);
                const float rb0 = (float)(qmax < o[2] ? qmax : o[2]);
                const float rb1 = (float)(rmax < o[3] ? rmax : o[3]);
                const float w = rb0 - lt0 > 0.0f ? rb0 - lt0 : 0.0f;
                const float hgt = rb1 - lt1 > 0.0f ? rb1 - lt1 : 0.0f;
                const float inter = w * hgt;
                const float aa = (float)dq * (float)dr;
                const float ab = (float)(o[2] - o[0]) * (float)(o[3] - o[1]);
                const float iou = inter / (aa + ab - inter);
                if (iou > mx) mx = iou;
            }
            ok = mx < p->max_iou;
        }
        if (ok) {
            if (nbox >= max_boxes) {
                nbox = -2;
                goto done;
            }
            boxes[4 * nbox + 0] = qmin;
            boxes[4 * nbox + 1] = rmin;
            boxes[4 * nbox + 2] = qmax;
            boxes[4 * nbox + 3] = rmax;
            ++nbox;
        }
    }

done:
    free(tidx); free(tsim); free(ilo); free(ihi); free(zero); free(indeg);
    free(order); free(dist); free(par); free(path);
    return nbox;
}
