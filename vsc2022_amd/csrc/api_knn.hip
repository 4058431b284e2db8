// C ABI of libvscmi.so, part 3: vsc_index_knn (vsc/index.py:167-177, vsc/baseline/score_normalization.py:96) -- the
// exact fp32 kernel, and for large problems the pre-filtered route over reference ranges with rising thresholds.
#include "api_internal.h"

extern "C" {

// exact fp32 k-NN (inner product) of the packed query rows against the first nr reference rows
static int knn_exact_ip(vsc_index* idx, const float* qp, int64_t nq, int64_t nr, int k, float* ds, int64_t* dj) {
    // Query rows in slabs of 65536: the workgroups of one launch walk the references together and share
    // them in L2 only while there are few enough of them to stay in step (measured: 200 k rows in one
    // launch ran at half the rate of 65536)
    const int64_t slab = 65536;
    if (nq > slab) {
        for (int64_t i0 = 0; i0 < nq; i0 += slab)
            VSC_TRY(knn_exact_ip(idx, qp + i0 * idx->dpad, std::min(slab, nq - i0), nr, k, ds + i0 * k, dj + i0 * k));
        return VSC_OK;
    }
    const int tq = (int)((nq + 127) / 128);
    const int tr = (int)((nr + 127) / 128);
    // Runs per query tile.  With >= 256 query tiles one run each already fills the 256 CUs, and every
    // extra run starts with empty top-k lists and pays the insertion storm again (k = 20, 65536 x 110 k:
    // 113 TFLOP/s with one run per query tile, 93 with four).  With few query tiles the references are
    // split until ~1024 workgroups exist (2000 x 1 M: 39 TFLOP/s with 64 runs, 24 with 32).
    int nchunk = tq >= 256 ? 1 : (int)std::min<int64_t>(tr, (1024 + tq - 1) / tq);
    nchunk = std::min(nchunk, 64);
    if (idx->knn_nchunk > 0) nchunk = std::max(1, std::min(std::min(idx->knn_nchunk, tr), 64));
    const int64_t nq_pad = (int64_t)tq * 128;
    VSC_TRY(idx->ws.parts.reserve((size_t)nq_pad * nchunk * k * 4));
    VSC_TRY(idx->ws.partj.reserve((size_t)nq_pad * nchunk * k * 4));
    SimKnnArgs a{qp, idx->ref.as<float>(), idx->dpad, (int)nq, (int)nr, tq, tr, nchunk, k,
                 idx->ws.parts.as<float>(), idx->ws.partj.as<int32_t>(), idx->knn_first_tile ? 0 : 1};
    hipEvent_t stop;
    VSC_TRY(prof_begin(idx, &stop));
    VSC_TRY(launch_sim_knn(a, idx->stream));
    VSC_TRY(prof_end(idx, stop, 2.0 * (double)nq * (double)nr * (double)idx->dim));
    KnnMergeArgs m{idx->ws.parts.as<float>(), idx->ws.partj.as<int32_t>(), (int)nq, nchunk, k, ds, dj, 0};
    VSC_TRY(launch_knn_merge(m, idx->stream));
    return VSC_OK;
}

// One thresholded pass of the pre-filtered k-NN over the reference rows [r_begin, r_end): pre-filter + exact stage of
// all query rows with the per-row thresholds in ws.rowthr; the k-NN lists of the rows [0, r_begin) that ds / dj hold
// (r_begin > 0) re-enter the hit list first, so that the (row asc, score desc, ref asc) sort + cut at k that follows
// yields the k-NN over [0, r_end) -> ds / dj.  `per_row` = expected hits of the range per query row.
// VSC_ERR_OVERFLOW when the estimate was too small (ds / dj untouched: the overflow is seen before they are rewritten).
static int knn_threshold_pass(vsc_index* idx, const float* qp, int64_t nq, int64_t r_begin, int64_t r_end, int k,
                              double per_row, float* ds, int64_t* dj, bool use_i8 = false) {
    const int64_t nrange = r_end - r_begin;
    // Rows per launch.  Round 4 started from 32768 everywhere; every launch carries ~0.3 ms of its own (row sort +
    // quantisation of its panels, candidate count + compaction + sort, one sync, the exact stage's ramp), and over a
    // short reference range (the first ranges of a k-NN: 12 k / 48 k rows at k = 1) a 32768-row launch is 0.4 / 1.6 TOP:
    // too little for 256 workgroups to reach the kernel's rate (1100 / 2260 TOP/s against 2900).  The rows double until
    // rows x range reaches VSC_KNN_STEP_WORK x 32768 x 196608 or VSC_KNN_STEP_MAX rows.  configs[3], score normalisation
    // per step (one box): work 1 / 2 / 4 / 8+ with max 262144 ... 1 M: 791 / 782 / 773 / 746-748 ms (32768 rows
    // everywhere: 826); max 65536 / 131072: 809 / 768.  Default: 64 and 262144 = 262144 rows over every range of a 2 M
    // index.  VSC_KNN_STEP=<rows>: fixed
    int64_t step = 32768;
    while (step < idx->knn_step_max && step * nrange < (int64_t)(idx->knn_step_work * 32768.0 * 196608.0)) step *= 2;
    if (idx->knn_step > 0) step = idx->knn_step;
    int64_t cap = (int64_t)((double)nq * per_row) + (r_begin > 0 ? nq * k : 0) + (1 << 20);
    cap = std::min<int64_t>(cap, nq * (nrange + k) + 1024);
    if (idx->hit_cap_user > 0) {
        cap = idx->hit_cap_user;
        // the lists so far (nq x k triples) re-enter the hit buffer before anything else: a user capacity below that
        // cannot hold them (knn_seed_hits writes unconditionally) -- the caller falls back to the exact kernel
        if (r_begin > 0 && cap < nq * k + 1024) return VSC_ERR_OVERFLOW;
    }
    // The candidate list is consumed slab by slab, only the hits accumulate over the whole query set.  Its size grows
    // with rows per launch x expected hits per row (k = 20: ~1500 entries per row on int8): bounded by a budget
    // (VSC_CAND_BUDGET entries, default 2^28 = 8.6 GB of list) by halving the rows per launch, so that several ranks
    // can share a GPU and smaller devices do not run out of memory
    const double per_row_c = per_row * (use_i8 ? 6.0 : 1.0);  // (the int8 bound is looser: ~4-5x the candidates per hit)
    while (step > 32768 && (double)std::min(nq, step) * per_row_c > (double)idx->cand_budget) step /= 2;
    const int64_t slab_rows = std::min(nq, step);
    int64_t ccap = std::min<int64_t>((int64_t)((double)slab_rows * per_row_c) + (1 << 20),
                                     slab_rows * nrange + 1024);
    if (!use_i8) ccap = std::min(ccap, std::max<int64_t>(cap, 1024));
    VSC_TRY(ensure_hit_buffers(idx, cap, ccap, false));
    VSC_TRY(init_ctl(idx, 0.0f));
    SelectCtl* ctl = idx->ws.ctl.as<SelectCtl>();
    if (r_begin > 0)
        VSC_TRY(launch_knn_seed_hits(ds, dj, nq, k, idx->ws.hA[0].as<int32_t>(), idx->ws.hA[1].as<int32_t>(),
                                     idx->ws.hA[2].as<float>(), &ctl->n, idx->stream));
    for (int64_t i0 = 0; i0 < nq; i0 += step)
        VSC_TRY(enqueue_f16(idx, qp, i0, std::min(nq, i0 + step), cap, idx->ws.rowthr.as<float>(), ccap, r_end, use_i8, r_begin));
    SelectCtl h;
    VSC_HIP(hipMemcpyAsync(&h, idx->ws.ctl.p, sizeof(h), hipMemcpyDeviceToHost, idx->stream));
    VSC_HIP(hipStreamSynchronize(idx->stream));
    idx->stat_candidates += h.n_cand_total;  // (over the ranges of one k-NN: knn_prefiltered resets it)
    if (h.overflow) return VSC_ERR_OVERFLOW;
    VSC_TRY(knn_from_hits(idx->ws.hA[0].as<int32_t>(), idx->ws.hA[1].as<int32_t>(), idx->ws.hA[2].as<float>(),
                          (int64_t)h.n, nq, k, idx->ws.w0, idx->ws.w1, idx->ws.w2, idx->ws.w3, idx->ws.tmp, ds, dj,
                          idx->stream));
    return VSC_OK;
}

// Pre-filtered exact k-NN (inner product).  Any lower bound T_i of a row's final k-th best score is a valid
// threshold: the k-th best score against a SUBSET of the references is one.
//   1. exact fp32 k-NN (sim_knn_kernel) against the first S0 references -> lists + thresholds T_i(0);
//   2. the remaining references in RANGES [S_l, S_l+1) that grow by `ratio` (round 4; round 3 re-searched whole
//      prefixes): the pre-filter (int8, else fp16) + exact stage over a range with the thresholds of everything before
//      it -- every pair whose low-precision score + error bound reaches T_i goes to the exact stage, which keeps exact
//      score >= T_i --, merged with the lists so far by one sort + cut at k -> the lists over [0, S_l+1) and the
//      tighter T_i(l+1).  Every reference row is visited once; a range brings ~k (ratio - 1) hits per query row (x the
//      filter's inflation in candidates) whatever its size, so the hits a search re-scores fall from k nr / S_last
//      (round 3's final pass: 30 per row at k = 1, 139 at k = 20) to ~k (ratio - 1) per level.
// Same result as knn_exact_ip bit for bit.  Returns VSC_ERR_OVERFLOW when a hit estimate was too small (the caller
// then runs the exact kernel).
static int knn_prefiltered(vsc_index* idx, const float* qp, int64_t nq, int k, float* ds, int64_t* dj) {
    const int64_t nr = idx->ntotal;
    // the thresholded passes run on the int8 kernel when the index keeps an int8 image (VSC_I8_KNN=0: fp16)
    const bool knn_i8_env = idx->knn_i8;
    bool knn_i8 = idx->i8_mode == 2 || (i8_usable(idx) && knn_i8_env);
    const double subset_factor = idx->knn_subset_factor;
    const bool levels = idx->knn_two_level;
    const int s0_div = idx->knn_s0_div;
    // without levels: S0 = sqrt(300 k nr) balances the exact pass (~2 dim S0 / 1e14 s per row) against the per-hit cost of
    // the one pass over the rest (k nr / S0 hits per row, ~1 ns each).  With levels the exact kernel -- a tenth of the
    // pre-filter's rate -- only has to get the thresholds started: S0 = that / 28, at least 1024 rows (/ 7 and 4096 until
    // the threshold passes over the short first ranges became cheap -- 262144 query rows per launch: configs[3] step
    // 2053 -> 2027 ms, 200 k x 2 M k-NN 169 -> 160 ms at k = 1 and 281 -> 273 ms at k = 20, profiles/r04_knn_launch_rows.md).
    // (range boundaries sit on col-steps; on whole 64-row wave tiles when VSC_PREFILTER=2, the tests' switch, forces the
    // levels on small problems)
    const int64_t unit = idx->prefilter_force ? 64 : F16P_COL_STEP;
    const int64_t S_one = std::min<int64_t>(
        nr, round_up64(std::max<int64_t>((int64_t)std::sqrt(subset_factor * k * (double)nr), 4096), unit));
    const int64_t S0_small = std::min<int64_t>(
        nr, round_up64(std::max<int64_t>(S_one / s0_div, idx->prefilter_force ? (int64_t)k : (int64_t)idx->knn_s0_min), unit));
    const bool refine = levels && (idx->prefilter_force ? nr >= 2 * S0_small
                                                        : (nr >= 8 * S_one && (double)nq * (double)nr >= 4e10));
    const int64_t S0 = refine ? S0_small : S_one;
    if (S0 < k) return VSC_ERR_OVERFLOW;
    idx->stat_candidates = 0;
    VSC_TRY(knn_exact_ip(idx, qp, nq, S0, k, ds, dj));
    const int64_t rows_h = round_up64(nq, ROW_PAD_H) + ROW_PAD_H;
    VSC_TRY(idx->ws.rowthr.reserve((size_t)rows_h * 4));
    VSC_TRY(launch_knn_row_thr(ds, nq, k, idx->ws.rowthr.as<float>(), rows_h, idx->stream));
    // ranges: [S0, r S0), [r S0, r^2 S0), ... -- the last one runs to nr (and swallows a remainder shorter than half a
    // range).  Every range costs its share of ONE pass over the references plus k (ratio - 1) hits per row (x the
    // filter's inflation) and the launches' own overhead (~0.3 ms per 32768-row slab: quantisation, sorts, one sync).
    // VSC_KNN_RATIO; measured at 200 k x 2 M (profiles/r04_knn_levels.md).
    const double ratio_env = idx->knn_ratio;
    const double ratio = !refine ? 1e30 : idx->prefilter_force ? 3.0 : (ratio_env > 1.0 ? ratio_env : 4.0);
    int64_t S_last = S0;
    while (S_last < nr) {
        int64_t S1 = ratio >= 1e29 ? nr : std::min<int64_t>(nr, round_up64((int64_t)(ratio * (double)S_last), unit));
        if (nr - S1 < (S1 - S_last) / 2) S1 = nr;
        // expected hits of the range per row: k (S1 - S_last) / S_last; generous factor
        const double per_row = (double)k * ((double)(S1 - S_last) / (double)S_last) * 4.0 + 16.0;
        int rc = knn_threshold_pass(idx, qp, nq, S_last, S1, k, per_row, ds, dj, knn_i8);
        if (rc == VSC_ERR_OVERFLOW && knn_i8 && idx->i8_mode != 2) {
            // (ds / dj still hold the lists over [0, S_last): the overflow is detected before they are rewritten)
            knn_i8 = false;
            idx->stat_i8_fallbacks += 1;
            rc = knn_threshold_pass(idx, qp, nq, S_last, S1, k, per_row, ds, dj, false);
        }
        if (rc != VSC_OK) return rc;
        S_last = S1;
        if (S_last < nr) VSC_TRY(launch_knn_row_thr(ds, nq, k, idx->ws.rowthr.as<float>(), rows_h, idx->stream));
    }
    return VSC_OK;
}

int vsc_index_knn(vsc_index_t* idx, const float* q, int64_t nq, int q_mem, int k, float* out_s,
                  int64_t* out_j, int out_mem) {
    if (!idx || nq < 0 || k <= 0 || k > 4096 || (nq > 0 && (!q || !out_s || !out_j))) {
        set_error("vsc_index_knn: invalid argument (k must be in 1..4096, got %d)", k);
        return VSC_ERR_INVALID;
    }
    if (nq == 0) return VSC_OK;
    if (nq >= 0x7fffff00LL) {
        set_error("vsc_index_knn: too many query rows");
        return VSC_ERR_INVALID;
    }
    VSC_HIP(hipSetDevice(idx->device));
    const bool ip = idx->metric == VSC_METRIC_INNER_PRODUCT;
    const int64_t nr = idx->ntotal;
    // The pre-filtered route pays off once the matrix is large (it adds sorts and an exact pass over 1/16 of
    // the references); VSC_PREFILTER=2 forces it for the tests.
    // k > 64 (the wavefront-sorted lists of the MFMA kernels hold one entry per lane): explicit score matrix in row
    // chunks + k rounds of wave arg-best -- the same fp32 chains, API completeness rather than speed
    const bool wide = k > 64;
    const bool pre = !wide && ip && idx->prefilter && nr >= k &&
                     (idx->prefilter_force || ((double)nq * (double)nr >= 4e9 && nr >= 65536));
    float* qp = nullptr;
    VSC_TRY(pack_queries(idx, q, nq, q_mem, &qp, pre));
    if (pre) VSC_TRY(i8_prepare(idx));
    float* ds = out_s;
    int64_t* dj = out_j;
    if (out_mem == VSC_MEM_HOST) {
        VSC_TRY(idx->ws.out[0].reserve((size_t)nq * k * 4));
        VSC_TRY(idx->ws.out[1].reserve((size_t)nq * k * 8));
        ds = idx->ws.out[0].as<float>();
        dj = idx->ws.out[1].as<int64_t>();
    }
    if (ip && nr > 0 && !wide) {
        int rc = pre ? knn_prefiltered(idx, qp, nq, k, ds, dj) : VSC_ERR_OVERFLOW;
        if (rc == VSC_ERR_OVERFLOW) rc = knn_exact_ip(idx, qp, nq, nr, k, ds, dj);
        VSC_TRY(rc);
    } else {
        // generic metric (or empty index): explicit score matrix, one run per row
        VSC_TRY(idx->ws.parts.reserve((size_t)nq * k * 4));
        VSC_TRY(idx->ws.partj.reserve((size_t)nq * k * 4));
        const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(nq, (int64_t)(1ll << 28) / std::max<int64_t>(nr, 1)));
        VSC_TRY(idx->ws.mat.reserve((size_t)chunk * std::max<int64_t>(nr, 1) * 4));
        for (int64_t r0 = 0; r0 < nq; r0 += chunk) {
            const int rows = (int)std::min(chunk, nq - r0);
            ScoreMatArgs sm{qp + r0 * idx->dpad, idx->ref.as<float>(), idx->dpad, idx->dim, rows, (int)nr,
                            idx->metric, idx->ws.mat.as<float>()};
            VSC_TRY(launch_score_matrix(sm, idx->stream));
            MatKnnArgs mk{idx->ws.mat.as<float>(), rows, (int)nr, k, idx->ws.parts.as<float>() + r0 * k,
                          idx->ws.partj.as<int32_t>() + r0 * k};
            VSC_TRY(launch_matrix_knn(mk, idx->stream));
        }
        KnnMergeArgs m{idx->ws.parts.as<float>(), idx->ws.partj.as<int32_t>(), (int)nq, 1, k, ds, dj, ip ? 0 : 1};
        VSC_TRY(launch_knn_merge(m, idx->stream));
    }
    if (out_mem == VSC_MEM_HOST) {
        VSC_HIP(hipMemcpyAsync(out_s, ds, (size_t)nq * k * 4, hipMemcpyDeviceToHost, idx->stream));
        VSC_HIP(hipMemcpyAsync(out_j, dj, (size_t)nq * k * 8, hipMemcpyDeviceToHost, idx->stream));
    }
    VSC_HIP(hipStreamSynchronize(idx->stream));
    VSC_TRY(prof_collect(idx));
    return VSC_OK;
}


}  // extern "C"
