/*
 * vsc_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's descriptor-search -> candidate ->
 * temporal-localization hot path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the shipped path
 * (vsc2022_amd + libvscmi.so) never does.
 *
 * What is restated, and from where (paths relative to /root/reference):
 *   - flat inner-product / L2 scoring          vsc/index.py:82,94 (faiss IndexFlat, third-party)
 *   - range search + adaptive global threshold vsc/index.py:142-165 and the third-party
 *     faiss.contrib.exhaustive_search.{exponential_query_iterator,range_search_max_results}
 *     (faiss ~1.7.x, NOT vendored under /root/reference; semantics per SURVEY.md Appendix A)
 *   - per-row k-NN                             vsc/index.py:167-177
 *   - (query video, ref video) regrouping      vsc/index.py:121-140
 *   - max aggregation + stable descending sort vsc/candidates.py:24-40
 *   - row L2 normalisation                     vsc/baseline/score_normalization.py:81-85 (sklearn normalize)
 *   - Temporal-Network localisation            vsc/baseline/localization.py:56-96 -> vcsl.vta (alipay/VCSL,
 *     third-party, source ABSENT from /root/reference: dangling symlink).  PARITY UNPINNED for TN:
 *     restated from SURVEY.md Appendix B + networkx 3.4.2 dag_longest_path semantics.
 *
 * Arithmetic contract (what "bit-exact" means in this repo): every similarity is the
 * fp32 fused-multiply-add chain over k = 0..d-1 in ascending k, starting from +0.0f:
 *     acc = fmaf(q[k], r[k], acc)
 * FAISS delegates to a BLAS sgemm whose summation order is unspecified, so FAISS itself
 * pins no bit pattern; this chain is the defined order that the gfx950 fp32 MFMA
 * (v_mfma_f32_32x32x2_f32) reproduces bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_METRIC_IP 0
#define ORC_METRIC_L2 1

#define JB 64 /* refs scored together; each lane keeps its own k-ordered chain */

/* ------------------------------------------------------------------ scoring */

/* pack refs [j0, j0+JB) transposed: rt[k*JB + jj] */
static void pack_panel(const float *r, int64_t nr, int64_t d, int64_t j0, float *rt) {
    for (int jj = 0; jj < JB; ++jj) {
        int64_t j = j0 + jj;
        if (j < nr) {
            const float *row = r + j * d;
            for (int64_t k = 0; k < d; ++k) rt[k * JB + jj] = row[k];
        } else {
            for (int64_t k = 0; k < d; ++k) rt[k * JB + jj] = 0.0f;
        }
    }
}

static inline void ip_panel(const float *q, const float *rt, int64_t d, float *acc) {
    for (int jj = 0; jj < JB; ++jj) acc[jj] = 0.0f;
    for (int64_t k = 0; k < d; ++k) {
        const float a = q[k];
        const float *b = rt + k * JB;
        for (int jj = 0; jj < JB; ++jj) acc[jj] = fmaf(a, b[jj], acc[jj]);
    }
}

static inline void l2_panel(const float *q, const float *rt, int64_t d, float *acc) {
    for (int jj = 0; jj < JB; ++jj) acc[jj] = 0.0f;
    for (int64_t k = 0; k < d; ++k) {
        const float a = q[k];
        const float *b = rt + k * JB;
        for (int jj = 0; jj < JB; ++jj) {
            const float df = a - b[jj];
            acc[jj] = fmaf(df, df, acc[jj]);
        }
    }
}

/* out[i*nr + j] = score(q_i, r_j).  faiss IndexFlat{IP,L2} distance (vsc/index.py:82). */
void orc_scores(const float *q, int64_t nq, const float *r, int64_t nr, int64_t d, int metric,
                float *out) {
    const int64_t npanel = (nr + JB - 1) / JB;
    /* (a few rows: one thread -- waking a team costs more than the work, and far more in a container whose CPU
     * quota is below the host's core count) */
#pragma omp parallel if (npanel > 1 && nq * nr * d >= (1 << 20))
    {
        float *rt = (float *)malloc(sizeof(float) * (size_t)(d > 0 ? d : 1) * JB);
        float acc[JB];
#pragma omp for schedule(static)
        for (int64_t p = 0; p < npanel; ++p) {
            const int64_t j0 = p * JB;
            pack_panel(r, nr, d, j0, rt);
            for (int64_t i = 0; i < nq; ++i) {
                if (metric == ORC_METRIC_IP) ip_panel(q + i * d, rt, d, acc);
                else l2_panel(q + i * d, rt, d, acc);
                for (int jj = 0; jj < JB && j0 + jj < nr; ++jj) out[i * nr + j0 + jj] = acc[jj];
            }
        }
        free(rt);
    }
}

/* --------------------------------------------------------- growable arrays */

typedef struct {
    int64_t *i;
    int64_t *j;
    float *s;
    int64_t n, cap;
} hits_t;

static int hits_reserve(hits_t *h, int64_t need) {
    if (need <= h->cap) return 0;
    int64_t cap = h->cap ? h->cap : 1024;
    while (cap < need) cap *= 2;
    int64_t *ni = (int64_t *)realloc(h->i, sizeof(int64_t) * (size_t)cap);
    if (!ni) return -1;
    h->i = ni;
    int64_t *nj = (int64_t *)realloc(h->j, sizeof(int64_t) * (size_t)cap);
    if (!nj) return -1;
    h->j = nj;
    float *ns = (float *)realloc(h->s, sizeof(float) * (size_t)cap);
    if (!ns) return -1;
    h->s = ns;
    h->cap = cap;
    return 0;
}

static void hits_free(hits_t *h) {
    free(h->i);
    free(h->j);
    free(h->s);
    memset(h, 0, sizeof(*h));
}

/* ------------------------------------------------------------ range search */

/* faiss IndexFlat.range_search for query rows [i0, i1): strict s > radius (IP) / s < radius (L2),
 * rows in ascending order, within a row refs in ascending order.  Appends to h. */
static int range_search_rows(const float *q, int64_t i0, int64_t i1, const float *r, int64_t nr,
                             int64_t d, int metric, float radius, hits_t *h) {
    const int64_t nrows = i1 - i0;
    if (nrows <= 0) return 0;
    const int64_t npanel = (nr + JB - 1) / JB;
    /* pass 1: count per row; pass 2: fill.  Scores are recomputed (deterministic). */
    int64_t *cnt = (int64_t *)calloc((size_t)nrows + 1, sizeof(int64_t));
    if (!cnt) return -1;
    int err = 0;
    for (int pass = 0; pass < 2 && !err; ++pass) {
        if (pass == 1) {
            int64_t tot = 0;
            for (int64_t t = 0; t < nrows; ++t) {
                int64_t c = cnt[t];
                cnt[t] = tot;
                tot += c;
            }
            cnt[nrows] = tot;
            if (hits_reserve(h, h->n + tot)) {
                err = -1;
                break;
            }
        }
#pragma omp parallel
        {
            float *rt = (float *)malloc(sizeof(float) * (size_t)(d > 0 ? d : 1) * JB);
            float acc[JB];
            /* thread-private per-row write cursors are unnecessary: parallelise over ROWS so a
             * row's hits are produced by one thread in ascending ref order. */
#pragma omp for schedule(dynamic, 4)
            for (int64_t t = 0; t < nrows; ++t) {
                const float *qrow = q + (i0 + t) * d;
                int64_t w = (pass == 1) ? h->n + cnt[t] : 0;
                int64_t c = 0;
                for (int64_t p = 0; p < npanel; ++p) {
                    const int64_t j0 = p * JB;
                    pack_panel(r, nr, d, j0, rt);
                    if (metric == ORC_METRIC_IP) ip_panel(qrow, rt, d, acc);
                    else l2_panel(qrow, rt, d, acc);
                    for (int jj = 0; jj < JB && j0 + jj < nr; ++jj) {
                        const int keep = (metric == ORC_METRIC_IP) ? (acc[jj] > radius) : (acc[jj] < radius);
                        if (keep) {
                            if (pass == 1) {
                                h->i[w] = i0 + t;
                                h->j[w] = j0 + jj;
                                h->s[w] = acc[jj];
                                ++w;
                            }
                            ++c;
                        }
                    }
                }
                if (pass == 0) cnt[t] = c;
            }
            free(rt);
        }
    }
    if (!err) h->n += cnt[nrows];
    free(cnt);
    return err;
}

/* A faster batch scorer used by the search below: parallel over ref panels so each packed
 * panel is reused by every row of the batch.  Produces the same hits in the same order. */
static int range_search_batch(const float *q, int64_t i0, int64_t i1, const float *r, int64_t nr,
                              int64_t d, int metric, float radius, hits_t *h, const float *rt_all) {
    const int64_t nrows = i1 - i0;
    if (nrows <= 0) return 0;
    const int64_t npanel = (nr + JB - 1) / JB;
    /* per (row, panel-chunk) counts would be large; instead keep per-thread hit lists per
     * panel range, then stitch in (row asc, ref asc) order with a counting pass. */
    int nth = 1;
#ifdef _OPENMP
    nth = omp_get_max_threads();
#endif
    if (npanel < nth) nth = (int)(npanel > 0 ? npanel : 1);
    hits_t *loc = (hits_t *)calloc((size_t)nth, sizeof(hits_t));
    int64_t *rowcnt = (int64_t *)calloc((size_t)nth * (size_t)nrows, sizeof(int64_t));
    if (!loc || !rowcnt) {
        free(loc);
        free(rowcnt);
        return -1;
    }
    int err = 0;
#pragma omp parallel num_threads(nth)
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        /* static contiguous panel ranges: thread t owns refs in ascending blocks */
        const int64_t p0 = npanel * tid / nth, p1 = npanel * (tid + 1) / nth;
        float *rt_own = (float *)malloc(sizeof(float) * (size_t)(d > 0 ? d : 1) * JB);
        float acc[JB];
        hits_t *L = &loc[tid];
        int64_t *rc = rowcnt + (size_t)tid * nrows;
        for (int64_t p = p0; p < p1; ++p) {
            const int64_t j0 = p * JB;
            const float *rt = rt_own;
            if (rt_all) rt = rt_all + (size_t)p * (size_t)d * JB;
            else pack_panel(r, nr, d, j0, rt_own);
            for (int64_t t = 0; t < nrows; ++t) {
                const float *qrow = q + (i0 + t) * d;
                if (metric == ORC_METRIC_IP) ip_panel(qrow, rt, d, acc);
                else l2_panel(qrow, rt, d, acc);
                for (int jj = 0; jj < JB && j0 + jj < nr; ++jj) {
                    const int keep = (metric == ORC_METRIC_IP) ? (acc[jj] > radius) : (acc[jj] < radius);
                    if (keep) {
                        if (hits_reserve(L, L->n + 1)) {
#pragma omp atomic write
                            err = -1;
                            break;
                        }
                        L->i[L->n] = t;
                        L->j[L->n] = j0 + jj;
                        L->s[L->n] = acc[jj];
                        ++L->n;
                        ++rc[t];
                    }
                }
            }
        }
        free(rt_own);
    }
    if (!err) {
        /* offsets: row-major over (row, thread) */
        int64_t tot = 0;
        int64_t *off = (int64_t *)malloc(sizeof(int64_t) * (size_t)nth * (size_t)nrows);
        if (!off) err = -1;
        else {
            for (int64_t t = 0; t < nrows; ++t)
                for (int th = 0; th < nth; ++th) {
                    off[(size_t)th * nrows + t] = tot;
                    tot += rowcnt[(size_t)th * nrows + t];
                }
            if (hits_reserve(h, h->n + tot)) err = -1;
            else {
                for (int th = 0; th < nth; ++th) {
                    hits_t *L = &loc[th];
                    int64_t *o = off + (size_t)th * nrows;
                    /* L is ordered by (panel asc, row asc, ref asc): scatter keeps ref order per row */
                    for (int64_t x = 0; x < L->n; ++x) {
                        const int64_t t = L->i[x];
                        const int64_t w = h->n + o[t]++;
                        h->i[w] = i0 + t;
                        h->j[w] = L->j[x];
                        h->s[w] = L->s[x];
                    }
                }
                h->n += tot;
            }
            free(off);
        }
    }
    for (int th = 0; th < nth; ++th) hits_free(&loc[th]);
    free(loc);
    free(rowcnt);
    return err;
}

/* C-callable single range search (faiss index.range_search contract): lims has nq+1 entries.
 * Two-call protocol: with D == NULL only lims is filled.  Returns number of hits or <0. */
int64_t orc_range_search(const float *q, int64_t nq, const float *r, int64_t nr, int64_t d, int metric,
                         float radius, int64_t *lims, float *D, int64_t *I, int64_t cap) {
    hits_t h;
    memset(&h, 0, sizeof(h));
    if (range_search_rows(q, 0, nq, r, nr, d, metric, radius, &h)) {
        hits_free(&h);
        return -1;
    }
    int64_t x = 0;
    for (int64_t i = 0; i < nq; ++i) {
        lims[i] = x;
        while (x < h.n && h.i[x] == i) ++x;
    }
    lims[nq] = h.n;
    if (D && I) {
        if (cap < h.n) {
            hits_free(&h);
            return -2;
        }
        memcpy(D, h.s, sizeof(float) * (size_t)h.n);
        memcpy(I, h.j, sizeof(int64_t) * (size_t)h.n);
    }
    const int64_t n = h.n;
    hits_free(&h);
    return n;
}

/* ----------------------------------------------------------- selection util */

static void swapf(float *a, float *b) {
    float t = *a;
    *a = *b;
    *b = t;
}

/* value of the element that would sit at position pos (0-based) in ASCENDING order */
static float nth_ascending(float *a, int64_t n, int64_t pos) {
    int64_t lo = 0, hi = n - 1;
    while (lo < hi) {
        /* median of three pivot, Hoare partition */
        const int64_t mid = lo + (hi - lo) / 2;
        if (a[mid] < a[lo]) swapf(&a[mid], &a[lo]);
        if (a[hi] < a[lo]) swapf(&a[hi], &a[lo]);
        if (a[hi] < a[mid]) swapf(&a[hi], &a[mid]);
        const float pv = a[mid];
        int64_t x = lo, y = hi;
        while (x <= y) {
            while (a[x] < pv) ++x;
            while (a[y] > pv) --y;
            if (x <= y) {
                swapf(&a[x], &a[y]);
                ++x;
                --y;
            }
        }
        if (pos <= y) hi = y;
        else if (pos >= x) lo = x;
        else return a[pos];
    }
    return a[pos];
}

/* stable merge sort of a permutation by (score desc|asc), ties keep ascending original position */
static void merge_sort_perm(int64_t *perm, int64_t *tmp, const float *s, int64_t n, int descending) {
    for (int64_t w = 1; w < n; w *= 2) {
        for (int64_t lo = 0; lo < n; lo += 2 * w) {
            int64_t mid = lo + w < n ? lo + w : n;
            int64_t hi = lo + 2 * w < n ? lo + 2 * w : n;
            int64_t a = lo, b = mid, o = lo;
            while (a < mid && b < hi) {
                const float sa = s[perm[a]], sb = s[perm[b]];
                /* take from the right run only if it is STRICTLY better */
                const int right = descending ? (sb > sa) : (sb < sa);
                tmp[o++] = right ? perm[b++] : perm[a++];
            }
            while (a < mid) tmp[o++] = perm[a++];
            while (b < hi) tmp[o++] = perm[b++];
        }
        memcpy(perm, tmp, sizeof(int64_t) * (size_t)n);
    }
}

/* ------------------------------------------- adaptive global-threshold search */

/*
 * VideoIndex._global_threshold_knn_search (vsc/index.py:142-165) over
 * faiss.contrib.exhaustive_search.range_search_max_results(index,
 *     exponential_query_iterator(Q), radius=-/+1e10, max_results=2K, min_results=K).
 *
 * Batches of query rows of size 32, 64, ... (doubling while bs < 20000).  After each batch, if the
 * number of kept hits exceeds 2K the radius becomes the (K+1)-th best kept score and every kept
 * hit is re-filtered with a STRICT comparison (hits tied with the radius are dropped).
 * Finally the hits, in (row asc, ref asc) order, are stably sorted by score (desc for IP, asc for
 * L2) and truncated to K (vsc/index.py:162-164).
 *
 * out_* must hold 2K+? entries: cap is checked.  Returns 0, or <0 on error (-2: cap too small).
 */
int orc_global_threshold_search(const float *q, int64_t nq, const float *r, int64_t nr, int64_t d,
                                int metric, int64_t K, int64_t *out_i, int64_t *out_j, float *out_s,
                                int64_t cap, int64_t *n_out, float *final_radius,
                                int64_t *n_rethreshold) {
    const int keep_max = (metric == ORC_METRIC_IP);
    float radius = keep_max ? (float)-1e10 : (float)1e10;
    const int64_t max_results = 2 * K, min_results = K;
    hits_t h;
    memset(&h, 0, sizeof(h));
    int64_t bs = 32, i0 = 0, nre = 0;
    int err = 0;
    /* transposed reference panels, packed once and shared by every batch (pure layout) */
    const int64_t npanel_all = (nr + JB - 1) / JB;
    float *rt_all = (float *)malloc(sizeof(float) * (size_t)(npanel_all > 0 ? npanel_all : 1) *
                                    (size_t)(d > 0 ? d : 1) * JB);
    if (rt_all) {
#pragma omp parallel for schedule(static)
        for (int64_t p = 0; p < npanel_all; ++p) pack_panel(r, nr, d, p * JB, rt_all + (size_t)p * (size_t)d * JB);
    }
    while (i0 < nq && !err) {
        const int64_t i1 = (i0 + bs < nq) ? i0 + bs : nq;
        if (range_search_batch(q, i0, i1, r, nr, d, metric, radius, &h, rt_all)) {
            err = -1;
            break;
        }
        if (h.n > max_results) {
            /* apply_maxres: radius = (min_results+1)-th best kept score */
            float *all = (float *)malloc(sizeof(float) * (size_t)h.n);
            if (!all) {
                err = -1;
                break;
            }
            memcpy(all, h.s, sizeof(float) * (size_t)h.n);
            if (keep_max) radius = nth_ascending(all, h.n, h.n - min_results - 1);
            else radius = nth_ascending(all, h.n, min_results);
            free(all);
            int64_t w = 0;
            for (int64_t x = 0; x < h.n; ++x) {
                const int keep = keep_max ? (h.s[x] > radius) : (h.s[x] < radius);
                if (keep) {
                    h.i[w] = h.i[x];
                    h.j[w] = h.j[x];
                    h.s[w] = h.s[x];
                    ++w;
                }
            }
            h.n = w;
            ++nre;
        }
        if (bs < 20000) bs *= 2;
        i0 = i1;
    }
    if (!err) {
        int64_t *perm = (int64_t *)malloc(sizeof(int64_t) * (size_t)(h.n ? h.n : 1));
        int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(h.n ? h.n : 1));
        if (!perm || !tmp) err = -1;
        else {
            for (int64_t x = 0; x < h.n; ++x) perm[x] = x;
            merge_sort_perm(perm, tmp, h.s, h.n, keep_max);
            const int64_t n = h.n > K ? K : h.n;
            if (n > cap) err = -2;
            else {
                for (int64_t x = 0; x < n; ++x) {
                    out_i[x] = h.i[perm[x]];
                    out_j[x] = h.j[perm[x]];
                    out_s[x] = h.s[perm[x]];
                }
                *n_out = n;
            }
        }
        free(perm);
        free(tmp);
    }
    if (final_radius) *final_radius = radius;
    if (n_rethreshold) *n_rethreshold = nre;
    hits_free(&h);
    free(rt_all);
    return err;
}

/* ------------------------------------------------------------------- k-NN */

/*
 * faiss index.search(x, k) as used by VideoIndex._knn_search (vsc/index.py:167-177) and
 * score_normalize (vsc/baseline/score_normalization.py:96).  Per row the k best refs ordered
 * by (score desc, ref asc) for IP, (dist asc, ref asc) for L2.  Slots beyond nr hold id -1 and
 * -FLT_MAX / +FLT_MAX.
 */
void orc_knn(const float *q, int64_t nq, const float *r, int64_t nr, int64_t d, int metric, int64_t k,
             float *out_s, int64_t *out_j) {
    const int keep_max = (metric == ORC_METRIC_IP);
    const int64_t npanel = (nr + JB - 1) / JB;
#pragma omp parallel
    {
        float *rt = (float *)malloc(sizeof(float) * (size_t)(d > 0 ? d : 1) * JB);
        float acc[JB];
#pragma omp for schedule(dynamic, 4)
        for (int64_t i = 0; i < nq; ++i) {
            float *bs = out_s + i * k;
            int64_t *bj = out_j + i * k;
            int64_t m = 0; /* filled */
            for (int64_t p = 0; p < npanel; ++p) {
                const int64_t j0 = p * JB;
                pack_panel(r, nr, d, j0, rt);
                if (keep_max) ip_panel(q + i * d, rt, d, acc);
                else l2_panel(q + i * d, rt, d, acc);
                for (int jj = 0; jj < JB && j0 + jj < nr; ++jj) {
                    const float s = acc[jj];
                    /* refs arrive in ascending order: a tie never displaces an earlier ref */
                    if (m == k) {
                        const int better = keep_max ? (s > bs[k - 1]) : (s < bs[k - 1]);
                        if (!better) continue;
                    }
                    int64_t pos = (m < k) ? m : k - 1;
                    while (pos > 0 && (keep_max ? (s > bs[pos - 1]) : (s < bs[pos - 1]))) {
                        bs[pos] = bs[pos - 1];
                        bj[pos] = bj[pos - 1];
                        --pos;
                    }
                    bs[pos] = s;
                    bj[pos] = j0 + jj;
                    if (m < k) ++m;
                }
            }
            for (; m < k; ++m) {
                bs[m] = keep_max ? -FLT_MAX : FLT_MAX;
                bj[m] = -1;
            }
        }
        free(rt);
    }
}

/* --------------------------------------------- regroup + max aggregation */

/*
 * VideoIndex.search regrouping (vsc/index.py:121-140) followed by MaxScoreAggregation and the
 * stable descending sort of CandidateGeneration.query (vsc/candidates.py:24-40).
 *
 * hits (hi, hj, hs) arrive in the order produced by the search (score-sorted list).  row2q / row2r
 * map a frame row to its video ordinal.  Pairs are emitted in first-appearance order, each with the
 * max score of its hits, then stably sorted by score descending.  Returns number of pairs.
 * out_first receives, per pair, the index of its first hit in the input list.
 */
int64_t orc_pair_max(const int64_t *hi, const int64_t *hj, const float *hs, int64_t n,
                     const int32_t *row2q, const int32_t *row2r, int32_t *out_q, int32_t *out_r,
                     float *out_s, int64_t *out_first) {
    if (n == 0) return 0;
    /* open-addressing hash on (qv, rv) */
    int64_t cap = 16;
    while (cap < 2 * n) cap *= 2;
    int64_t *slot = (int64_t *)malloc(sizeof(int64_t) * (size_t)cap);
    int32_t *pq = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int32_t *pr = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    float *ps = (float *)malloc(sizeof(float) * (size_t)n);
    int64_t *pf = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    if (!slot || !pq || !pr || !ps || !pf) {
        free(slot); free(pq); free(pr); free(ps); free(pf);
        return -1;
    }
    for (int64_t x = 0; x < cap; ++x) slot[x] = -1;
    int64_t np = 0;
    for (int64_t x = 0; x < n; ++x) {
        const int32_t qv = row2q[hi[x]], rv = row2r[hj[x]];
        uint64_t key = ((uint64_t)(uint32_t)qv << 32) | (uint32_t)rv;
        uint64_t hsh = key * 0x9E3779B97F4A7C15ull;
        int64_t at = (int64_t)(hsh >> 20) & (cap - 1);
        for (;;) {
            const int64_t pidx = slot[at];
            if (pidx < 0) {
                slot[at] = np;
                pq[np] = qv;
                pr[np] = rv;
                ps[np] = hs[x];
                pf[np] = x;
                ++np;
                break;
            }
            if (pq[pidx] == qv && pr[pidx] == rv) {
                /* np.max over the pair's scores (vsc/candidates.py:26) */
                if (hs[x] > ps[pidx]) ps[pidx] = hs[x];
                break;
            }
            at = (at + 1) & (cap - 1);
        }
    }
    int64_t *perm = (int64_t *)malloc(sizeof(int64_t) * (size_t)np);
    int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (size_t)np);
    if (!perm || !tmp) {
        free(slot); free(pq); free(pr); free(ps); free(pf); free(perm); free(tmp);
        return -1;
    }
    for (int64_t x = 0; x < np; ++x) perm[x] = x;
    merge_sort_perm(perm, tmp, ps, np, 1); /* sorted(..., reverse=True) is stable */
    for (int64_t x = 0; x < np; ++x) {
        out_q[x] = pq[perm[x]];
        out_r[x] = pr[perm[x]];
        out_s[x] = ps[perm[x]];
        if (out_first) out_first[x] = pf[perm[x]];
    }
    free(slot); free(pq); free(pr); free(ps); free(pf); free(perm); free(tmp);
    return np;
}

/* ----------------------------------------------------------- normalisation */

/* sklearn.preprocessing.normalize(X) row L2 (score_normalization.py:84, sscd_baseline.py:129-130):
 * norm = sqrt(sum x^2) with zero norms replaced by 1.  The sum is the ascending-k fp32 fma chain
 * (sklearn's einsum order is unspecified; compared with a tolerance against the reference). */
void orc_row_normalize(const float *x, int64_t n, int64_t d, float *out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const float *row = x + i * d;
        float acc = 0.0f;
        for (int64_t k = 0; k < d; ++k) acc = fmaf(row[k], row[k], acc);
        float nrm = sqrtf(acc);
        if (nrm == 0.0f) nrm = 1.0f;
        for (int64_t k = 0; k < d; ++k) out[i * d + k] = row[k] / nrm;
    }
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
