// C ABI of libvscmi.so, part 2: the thresholded searches -- the batch schedule of vsc_index_global_topk
// (vsc/index.py:142-165), its seeded form, vsc_index_candidates, vsc_index_range_search -- and the batch machinery
// (pre-filter launch + candidate list + exact stage) that the k-NN's threshold passes share (api_knn.hip).
#include "api_internal.h"

// The candidate list of one pre-filter launch (cand_list.h): `ccap` entries in per-wave segments + the chunked tail
// behind them.  Fills the list fields that SimF16Args / SimF16PArgs / SimI8PArgs share and keeps the geometry for
// the exact stage.
struct CandList {
    int grid = 0, seg_cap = 0, tail_shift = 6;
    int64_t tail_base = 0;
    long long tail_cap = 0;
};
template <class Args>
static int cand_list_setup(vsc_index* idx, int64_t ccap, int grid, Args& f, CandList& cl) {
    SelectCtl* ctl = idx->ws.ctl.as<SelectCtl>();
    cl.grid = grid;
    cl.seg_cap = (int)std::min<int64_t>(ccap / (grid * 8), 0x7fffffff);
    cl.tail_base = (int64_t)cl.seg_cap * grid * 8;
    cl.tail_cap = cand_entries(ccap) - cl.tail_base;
    cl.tail_shift = tail_chunk_shift_for(cl.tail_cap, grid * 8);
    VSC_TRY(idx->ws.tailfill.reserve((size_t)((cl.tail_cap >> cl.tail_shift) + 2) * sizeof(int)));
    f.out_i = idx->ws.ci.as<int32_t>();
    f.out_j = idx->ws.cj.as<int32_t>();
    f.seg_cap = cl.seg_cap;
    f.seg_count = idx->ws.segcnt.as<int>();
    f.tail_base = cl.tail_base;
    f.tail_cap = cl.tail_cap;
    f.tail_shift = cl.tail_shift;
    f.tail_fill = idx->ws.tailfill.as<int>();
    f.tail_count = &ctl->n_tail;
    f.overflow = &ctl->overflow;
    return VSC_OK;
}


// fp16 pre-filter + exact re-scoring of query rows [i0, i1): appends to hit buffer A every (row, ref, score)
// with score > *radius -- or, when `row_thr` (one threshold per query row, padded like the fp16 query
// image) is given, with score >= row_thr[row].
int enqueue_f16(vsc_index* idx, const float* qpacked, int64_t i0, int64_t i1, int64_t cap,
                const float* row_thr, int64_t ccap, int64_t nr_limit, bool use_i8, int64_t nr_begin) {
    if (ccap < 0) ccap = cap;  // capacity of the candidate list (cap: of the hit list)
    // [nr_begin, nr_limit): search only these reference rows (the levels of the k-NN); nr_begin a multiple of 64
    // (whole wave tiles of the fragment-major images).  The kernels see the images from row nr_begin on and emit
    // refs relative to it; the exact stage adds the offset back (RescoreArgs::j0).
    const int64_t nr_end = nr_limit >= 0 ? std::min<int64_t>(nr_limit, idx->ntotal) : idx->ntotal;
    if (nr_begin < 0 || nr_begin % 64 != 0 || nr_begin > nr_end) {
        set_error("enqueue_f16: reference range [%lld, %lld) does not start on a 64-row tile", (long long)nr_begin, (long long)nr_end);
        return VSC_ERR_INVALID;
    }
    const int64_t nrefs = nr_end - nr_begin;  // rows the kernels see
    SelectCtl* ctl = idx->ws.ctl.as<SelectCtl>();
    const int nqb = (int)(i1 - i0);
    {
        // 1. fp16 pre-filter: candidates = pairs whose fp16 score + error bound exceeds the threshold
        const double D = (double)idx->dpadh;
        // |fp16 score - exact score| <= c1 |q||r| + c2 (|q| + |r|) + c3   (|x| = L2 norm):
        //   rounding to fp16: |x - h(x)| <= 2^-11 |x| + 2^-25 per element (normal / subnormal range)
        //     => sum |q r - h(q) h(r)| <= (2^-10 + 2^-22) |q||r| + 2^-25 * 1.001 * sqrt(D) (|q|+|r|) + D 2^-50
        //   accumulation: the exact fp32 fma chain (D roundings) and the MFMA's fp32 accumulation
        //     (D/16 instructions of 16 products + addend) each stay within 2^-23 |q||r| per operation
        const float c1 = (float)(ldexp(1.0, -10) + ldexp(1.0, -22) + (2.0 * D + D / 16.0 + 16.0) * ldexp(1.0, -23));
        const float c2 = (float)(ldexp(1.0, -25) * 1.001 * sqrt(D));
        const float c3 = (float)(D * ldexp(1.0, -50));
        int grid = 0;
        CandList cl;
        const int32_t* cand_perm = nullptr;  // set when the candidate list holds positions of a permuted int8 launch
        hipEvent_t stop;
        int pcls = 1;
        if (use_i8 && idx->i8_mode) {
            // int8 panel kernel (sim_i8p.hip): this launch's rows are quantised first, one scale per 128-row panel
            SimI8PArgs f;
            sim_f16p_plan(nqb, nrefs, &f.npanel, &f.nsteps, &f.slice, &grid);
            {
                // work order: slice-major items.  r03 (32x32x32 kernel): items of 16 col-steps (4 MiB of the int8 image at
                // 512-d: what an XCD's L2 holds) +3 % on the bench over panel-major (2431 -> 2507-2515 TOP/s).  r04
                // (16x16x64 kernel, configs[3]): 8 / 16 / 32 / 64 col-steps 2456 / 2390 / 2363 / 2368 ms per query set
                // -- the faster K loop makes the hand-over (panel load + two barriers) the larger share: 32.
                // VSC_I8P_ORDER=0: panel-major with stealing as in sim_f16p
                const int slice_env = idx->i8p_slice;
                f.order = idx->i8p_order;
                if (f.order == 1) f.slice = std::max(1, std::min(f.nsteps, slice_env > 0 ? slice_env : 32));
                else if (slice_env > 0) f.slice = std::max(1, std::min(f.nsteps, slice_env));
            }
            // work items of two panels (wave tiles of 256 rows x 32 columns: half the reference bytes per MFMA) where the
            // launch is large enough; the quantised image then holds an even number of panels.  VSC_I8P_PAIR=0: off
            f.pair = idx->i8p_pair && sim_i8p_pairs(idx->dpad8, f.npanel, f.nsteps, f.slice, idx->i8p_pair == 2) ? 1 : 0;
            const int npanel_q = f.pair ? (f.npanel + 1) & ~1 : f.npanel;
            VSC_TRY(idx->ws.slices.reserve(((size_t)f.npanel + 1) * sizeof(int)));
            VSC_TRY(idx->ws.q8.reserve((size_t)npanel_q * F16P_PANEL_ROWS * idx->dpad8));
            VSC_TRY(idx->ws.pstat.reserve((size_t)npanel_q * sizeof(float4)));
            hipEvent_t prep_stop;
            VSC_TRY(prof_begin(idx, &prep_stop, 6));
            const int32_t* perm = nullptr;
            float* rt_pos = nullptr;
            const float* thr_src = row_thr ? row_thr + i0 : nullptr;
            if (idx->i8_ex.n > 0 || idx->i8_mu_on) {
                // coordinates the images leave out (all references agree on them) act through the rows' thresholds:
                // t_row - sum_c q_c v_c, with t_row the row's k-NN threshold or the search radius; a centred reference
                // image (quant_i8.hip) adds the rows' x . mu the same way
                VSC_TRY(idx->ws.rt8b.reserve((size_t)nqb * sizeof(float)));
                const float *cen = nullptr, *cmag = nullptr;
                if (idx->i8_mu_on) {
                    VSC_TRY(idx->ws.rt8d.reserve((size_t)2 * nqb * sizeof(float)));
                    VSC_TRY(launch_row_center(qpacked + i0 * idx->dpad, idx->dpad, nqb, idx->i8_mu.as<float>(),
                                              idx->ws.rt8d.as<float>(), idx->ws.rt8d.as<float>() + nqb, idx->stream));
                    cen = idx->ws.rt8d.as<float>();
                    cmag = cen + nqb;
                }
                VSC_TRY(launch_row_bias_thresholds(qpacked + i0 * idx->dpad, idx->dpad, nqb, thr_src, &ctl->radius,
                                                   idx->i8_ex, cen, cmag, idx->ws.rt8b.as<float>(), idx->stream));
                thr_src = idx->ws.rt8b.as<float>();
            }
            // VSC_I8_SORT=0: rows in their own order (A/B; the kernel then gates blocks of unrelated thresholds)
            const bool sort_rows = idx->i8_sort_rows;
            if (thr_src && !sort_rows) {
                VSC_TRY(idx->ws.rt8.reserve((size_t)npanel_q * F16P_PANEL_ROWS * sizeof(float)));
                rt_pos = idx->ws.rt8.as<float>();
            } else if (thr_src) {
                // thresholds that differ from row to row: the launch sees its rows sorted by threshold (the kernel
                // gates a tile by its panel's smallest threshold and a 16-row block by the block's)
                VSC_TRY(idx->ws.rt8.reserve((size_t)npanel_q * F16P_PANEL_ROWS * sizeof(float)));
                rt_pos = idx->ws.rt8.as<float>();
                if (idx->i8_group_shift > 0 && !row_thr && nqb >= (4 << idx->i8_group_shift)) {
                    // ... and, inside groups of 512 positions of that order, by the rows' largest element (sortpairs.hip).
                    // Only for the radius search over excluded coordinates (thresholds = radius - the rows' bias: a
                    // narrow spread): configs[3] 1162 -> 1086 M candidates, exact stage 367 -> 341 ms (groups of 256 /
                    // 512 / 1024 / 2048 / 4096: 1110 / 1086 / 1090 / 1125 / 1205 M).  The k-NN's thresholds -- each
                    // row's best score so far -- spread far more: there the same grouping cost 2 % (892 -> 907 ms).
                    VSC_TRY(idx->ws.rt8c.reserve((size_t)nqb * sizeof(float)));
                    VSC_TRY(launch_row_absmax(qpacked + i0 * idx->dpad, idx->dpad, nqb, idx->i8_ex, idx->ws.rt8c.as<float>(),
                                              idx->stream));
                    VSC_TRY(sort_rows_by_threshold_then_scale(thr_src, idx->ws.rt8c.as<float>(), nqb, idx->i8_group_shift,
                                                              idx->ws.w0, idx->ws.w1, idx->ws.w2, idx->ws.w3, idx->ws.tmp,
                                                              &perm, idx->stream));
                } else
                VSC_TRY(sort_rows_by_threshold(thr_src, nqb, idx->ws.w0, idx->ws.w1, idx->ws.w2, idx->ws.w3, idx->ws.tmp,
                                               &perm, idx->stream));
            }
            else if (sort_rows && nqb >= 2 * F16P_PANEL_ROWS) {
                // one threshold for all rows (the search radius): sort by the rows' largest element instead, so that
                // a panel's shared scale is close to what each of its rows would have chosen (VSC_I8_SORT=0: off)
                VSC_TRY(idx->ws.rt8b.reserve((size_t)nqb * sizeof(float)));
                VSC_TRY(launch_row_absmax(qpacked + i0 * idx->dpad, idx->dpad, nqb, idx->i8_ex, idx->ws.rt8b.as<float>(),
                                          idx->stream));
                VSC_TRY(sort_rows_by_threshold(idx->ws.rt8b.as<float>(), nqb, idx->ws.w0, idx->ws.w1, idx->ws.w2, idx->ws.w3,
                                               idx->ws.tmp, &perm, idx->stream));
            }
            VSC_TRY(launch_quant_query_panels(qpacked + i0 * idx->dpad, idx->dpad, nqb, npanel_q, idx->ws.q8.p, idx->dpad8,
                                              idx->ws.pstat.as<float4>(), perm, thr_src, rt_pos, idx->i8_ex, idx->stream));
            f.Q = idx->ws.q8.p;
            f.pstat = idx->ws.pstat.as<float4>();
            f.Rf = static_cast<const char*>(idx->ref8.p) + nr_begin * idx->dpad8;  // (whole 64-row tiles: dpad8 x 64 B each)
            f.rmeta = idx->ref8m.as<float4>() + nr_begin;
            f.dpad8 = idx->dpad8;
            f.nq = nqb;
            f.i0 = (int)i0;
            f.nr = (int)nrefs;
            f.next_slice = idx->ws.slices.as<int>();
            // the exact fp32 chain is within dpad 2^-24 |q||r| (1 + tiny) of the real inner product
            f.c_acc = (float)(((double)idx->dpad + 2.0) * ldexp(1.0, -23));
            f.radius = &ctl->radius;
            f.row_thr = rt_pos;
            cand_perm = perm;
            VSC_TRY(cand_list_setup(idx, ccap, grid, f, cl));
            VSC_TRY(prof_end(idx, prep_stop, 0.0, 6));
            VSC_TRY(prof_begin(idx, &stop, 5));  // (the kernel alone: what the roofline figure is about)
            VSC_TRY(launch_sim_i8p(f, grid, idx->stream));
            pcls = 5;
        } else if (idx->frag) {
            // panel-stationary kernel (sim_f16p.hip): LDS-resident query panels x the fragment-major reference image
            SimF16PArgs f;
            sim_f16p_plan(nqb, nrefs, &f.npanel, &f.nsteps, &f.slice, &grid);
            VSC_TRY(idx->ws.slices.reserve((size_t)f.npanel * sizeof(int)));
            f.Q = idx->ws.qh.as<_Float16>() + i0 * idx->dpadh;
            f.Rf = static_cast<const char*>(idx->refh.p) + nr_begin * idx->dpadh * 2;
            f.qn = idx->ws.qn.as<float>() + i0;
            f.rn = idx->refn.as<float>() + nr_begin;
            f.dpadh = idx->dpadh;
            f.nq = nqb;
            f.i0 = (int)i0;
            f.nr = (int)nrefs;
            f.next_slice = idx->ws.slices.as<int>();
            f.c1 = c1; f.c2 = c2; f.c3 = c3;
            f.radius = &ctl->radius;
            f.row_thr = row_thr ? row_thr + i0 : nullptr;
            VSC_TRY(cand_list_setup(idx, ccap, grid, f, cl));
            VSC_TRY(prof_begin(idx, &stop, 1));
            VSC_TRY(launch_sim_f16p(f, grid, idx->stream));
        } else {
            // dims > 512: 256x256 LDS-ring kernel (sim_f16.hip) on the natural image
            SimF16Args f;
            f.Q = idx->ws.qh.as<_Float16>() + i0 * idx->dpadh;
            f.R = idx->refh.as<_Float16>() + nr_begin * idx->dpadh;
            f.qn = idx->ws.qn.as<float>() + i0;
            f.rn = idx->refn.as<float>() + nr_begin;
            f.dpadh = idx->dpadh;
            f.nq = nqb;
            f.i0 = (int)i0;
            f.nr = (int)nrefs;
            f.tq = (nqb + 255) / 256;
            f.tr = (int)((nrefs + 255) / 256);
            f.c1 = c1; f.c2 = c2; f.c3 = c3;
            f.radius = &ctl->radius;
            f.row_thr = row_thr ? row_thr + i0 : nullptr;
            grid = sim_f16_grid(f.tq, f.tr);
            VSC_TRY(cand_list_setup(idx, ccap, grid, f, cl));
            VSC_TRY(prof_begin(idx, &stop, 1));
            VSC_TRY(launch_sim_f16(f, idx->stream));
        }
        VSC_TRY(prof_end(idx, stop, 2.0 * (double)nqb * (double)nrefs * (double)idx->dim, pcls));
        // 2. exact scores of the candidates; those above the radius join the kept hits
        RescoreArgs r;
        r.Q = qpacked;
        r.R = idx->ref.as<float>();
        r.dpad = idx->dpad;
        r.cand_i = idx->ws.ci.as<int32_t>();
        r.cand_j = idx->ws.cj.as<int32_t>();
        r.n_seg = cl.grid * 8;
        r.seg_cap = cl.seg_cap;
        r.seg_count = idx->ws.segcnt.as<int>();
        r.tail_base = cl.tail_base;
        r.tail_cap = cl.tail_cap;
        r.tail_count = &ctl->n_tail;
        r.tail_shift = cl.tail_shift;
        r.tail_fill = idx->ws.tailfill.as<int>();
        r.perm = cand_perm;
        r.perm_i0 = (int)i0;
        r.n_cand_total = &ctl->n_cand_total;
        r.radius = &ctl->radius;
        r.out_i = idx->ws.hA[0].as<int32_t>();
        r.out_j = idx->ws.hA[1].as<int32_t>();
        r.out_s = idx->ws.hA[2].as<float>();
        r.counter = &ctl->n;
        r.cap = cap;
        r.overflow = &ctl->overflow;
        r.row_thr = row_thr;
        r.j0 = (int)nr_begin;
        VSC_TRY(prof_begin(idx, &stop, 2));
        // The candidates are compacted out of the waves' segments, sorted by reference row and re-scored as one dense
        // list (sim_f16.hip, "candidates ordered by reference row"): 74 -> 54 ms per bench step, k-NN k = 20 140 ->
        // 100 ms.  It needs the candidate count on the host (buffer sizes, grid of the sort): one stream sync per
        // launch, ~20 us against launches of 3-30 ms.  VSC_RESCORE_SORT=0: the segments as they are.
        const bool by_ref = idx->rescore_by_ref;
        if (by_ref) {
            // count first (one tiny kernel + the stream sync the sort needs anyway), then size the four dense lists of
            // the sort from what the launch really left behind -- not from the list's capacity (96 bytes
            // per unit of capacity, 26 GB for a default range search whose launches hold a few percent of that)
            VSC_TRY(idx->ws.csn.reserve(3 * sizeof(unsigned long long)));
            const int n_chunks_max = (int)std::min<long long>((cl.tail_cap >> cl.tail_shift) + 1, 1 << 20);
            VSC_TRY(launch_cand_count(r, n_chunks_max, idx->ws.csn.as<unsigned long long>() + 2, idx->stream));
            unsigned long long n_c = 0;
            VSC_HIP(hipMemcpyAsync(&n_c, idx->ws.csn.as<unsigned long long>() + 2, sizeof(n_c), hipMemcpyDeviceToHost, idx->stream));
            VSC_HIP(hipStreamSynchronize(idx->stream));
            // (grown in steps of a quarter so that launches of slowly varying size do not reallocate every time)
            const size_t cap_e = (size_t)(n_c + n_c / 4 + 4096);
            for (auto& b : idx->ws.cs)
                if (b.bytes < (size_t)(n_c + 1) * sizeof(uint32_t)) VSC_TRY(b.reserve(cap_e * sizeof(uint32_t)));
            VSC_TRY(launch_cand_compact(r, n_chunks_max, idx->ws.cs[0].as<uint32_t>(), idx->ws.cs[2].as<uint32_t>(),
                                        idx->ws.csn.as<unsigned long long>(), idx->stream));
            const uint32_t *sj = nullptr, *si = nullptr;
            VSC_TRY(sort_candidates_by_ref(idx->ws.cs[0].as<uint32_t>(), idx->ws.cs[1].as<uint32_t>(), idx->ws.cs[2].as<uint32_t>(),
                                           idx->ws.cs[3].as<uint32_t>(), (int64_t)n_c, nr_end, idx->ws.cstmp, &sj, &si, idx->stream));
            // VSC_I8_SCREEN=1: int8 launches pass an fp16 screen first (sim_f16.hip: f16_screen_kernel).  Measured
            // neutral and therefore OFF by default: 29 % of the int8 candidates survive it (bench, 128 M -> 37 M per
            // step), the screen moves half the bytes per pair (23.8 ms) and the exact stage then costs 35.6 instead of
            // 59.8 ms -- both stages gather one query row per pair from the Infinity Cache at ~6 TB/s, which is the
            // bound (profiles/r03_prefilter_attribution.md).  Kept because it pays once the survivor share drops
            // (descriptors with outlier coordinates widen the int8 bound, not the fp16 one).
            const bool screen = idx->i8_screen;
            if (pcls == 5 && screen && n_c > 0) {
                ScreenArgs sa;
                sa.Qh = idx->ws.qh.as<_Float16>();
                sa.qn = idx->ws.qn.as<float>();
                sa.Rh = idx->refh.as<_Float16>();
                sa.rn = idx->refn.as<float>();
                sa.dpadh = idx->dpadh;
                sa.frag = idx->frag ? 1 : 0;
                sa.c1 = c1; sa.c2 = c2; sa.c3 = c3;
                sa.radius = &ctl->radius;
                sa.row_thr = row_thr;
                sa.sj = sj;
                sa.si = si;
                sa.n = (long long)n_c;
                sa.out_j = sj == idx->ws.cs[0].as<uint32_t>() ? idx->ws.cs[1].as<uint32_t>() : idx->ws.cs[0].as<uint32_t>();
                sa.out_i = si == idx->ws.cs[2].as<uint32_t>() ? idx->ws.cs[3].as<uint32_t>() : idx->ws.cs[2].as<uint32_t>();
                sa.n_out = idx->ws.csn.as<unsigned long long>() + 1;
                sa.n_cand_total = &ctl->n_cand_total;
                sa.overflow = &ctl->overflow;
                VSC_TRY(launch_f16_screen(sa, idx->stream));
                VSC_TRY(launch_rescore_dense(r, sa.out_j, sa.out_i, (long long)n_c, idx->stream, sa.n_out));
                if (idx->debug_screen) {
                    unsigned long long n_s = 0;
                    VSC_HIP(hipMemcpyAsync(&n_s, sa.n_out, sizeof(n_s), hipMemcpyDeviceToHost, idx->stream));
                    VSC_HIP(hipStreamSynchronize(idx->stream));
                    fprintf(stderr, "[vscmi] fp16 screen: %llu of %llu int8 candidates left (rows %d)\n", n_s, n_c, nqb);
                }
            } else {
                VSC_TRY(launch_rescore_dense(r, sj, si, (long long)n_c, idx->stream));
            }
        } else {
            VSC_TRY(launch_rescore(r, idx->stream));
        }
        VSC_TRY(prof_end(idx, stop, 0.0, 2));
    }
    return VSC_OK;
}

// Append every (row, ref) of query rows [i0, i1) with score > *radius (score space: IP as is, L2
// negated) to the hit buffer A.
static int enqueue_batch(vsc_index* idx, const float* qpacked, int64_t i0, int64_t i1, int64_t cap,
                         bool use_f16 = false, bool use_i8 = false) {
    SelectCtl* ctl = idx->ws.ctl.as<SelectCtl>();
    const int nqb = (int)(i1 - i0);
    if (use_f16) return enqueue_f16(idx, qpacked, i0, i1, cap, nullptr, -1, -1, use_i8);
    if (idx->metric == VSC_METRIC_INNER_PRODUCT) {
        SimThreshArgs a;
        a.Q = qpacked + i0 * idx->dpad;
        a.R = idx->ref.as<float>();
        a.dpad = idx->dpad;
        a.nq = nqb;
        a.i0 = (int)i0;
        a.nr = (int)idx->ntotal;
        a.tq = (nqb + 127) / 128;
        a.tr = (int)((idx->ntotal + 127) / 128);
        a.radius = &ctl->radius;
        a.out_i = idx->ws.hA[0].as<int32_t>();
        a.out_j = idx->ws.hA[1].as<int32_t>();
        a.out_s = idx->ws.hA[2].as<float>();
        a.counter = &ctl->n;
        a.cap = cap;
        a.overflow = &ctl->overflow;
        hipEvent_t stop;
        VSC_TRY(prof_begin(idx, &stop));
        VSC_TRY(launch_sim_thresh(a, idx->stream));
        VSC_TRY(prof_end(idx, stop, 2.0 * (double)nqb * (double)idx->ntotal * (double)idx->dim));
        return VSC_OK;
    }
    // generic metric: explicit score matrix in row chunks
    const int64_t nr = idx->ntotal;
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(nqb, (int64_t)(1ll << 28) / std::max<int64_t>(nr, 1)));
    VSC_TRY(idx->ws.mat.reserve((size_t)chunk * nr * 4));
    for (int64_t r0 = i0; r0 < i1; r0 += chunk) {
        const int rows = (int)std::min(chunk, i1 - r0);
        ScoreMatArgs m{qpacked + r0 * idx->dpad, idx->ref.as<float>(), idx->dpad, idx->dim, rows, (int)nr,
                       idx->metric, idx->ws.mat.as<float>()};
        VSC_TRY(launch_score_matrix(m, idx->stream));
        MatThreshArgs t{idx->ws.mat.as<float>(), rows, (int)nr, (int)r0, &ctl->radius,
                        idx->ws.hA[0].as<int32_t>(), idx->ws.hA[1].as<int32_t>(), idx->ws.hA[2].as<float>(),
                        &ctl->n, cap, &ctl->overflow};
        VSC_TRY(launch_matrix_thresh(t, idx->stream));
    }
    return VSC_OK;
}

int init_ctl(vsc_index* idx, float radius_score_space) {
    // (a one-workgroup kernel instead of a host-staged copy + synchronisation: the sharded schedule calls the seeded
    // search once per batch and rank, and every host round trip is ~40 us of idle GPU there -- profiles/r06_rank_work.md)
    return launch_ctl_init(idx->ws.ctl.as<SelectCtl>(), radius_score_space, idx->stream);
}


// The body of vsc_index_global_topk.  seeded = false: the reference's schedule (batches of 32, 64, ... rows doubling
// while < 20000, radius from -1e10).  seeded = true (vsc_index_global_topk_seeded): the caller already knows a radius
// below the K-th best score -- every batch is a steady 32768-row batch from the first row on, pre-filtered from the
// first row on; the re-threshold rule stays (kept > 2K: radius <- (K+1)-th best), so the buffers stay bounded when the
// seed was low.
static int global_topk_impl(vsc_index_t* idx, const float* q, int64_t nq, int q_mem, int64_t K, bool seeded,
                            float radius0, int32_t* out_i, int32_t* out_j, float* out_s, int64_t cap_out, int out_mem,
                            int64_t* n_out, float* final_radius) {
    if (!idx || nq < 0 || K < 0 || !n_out || (nq > 0 && !q)) {
        set_error("vsc_index_global_topk: invalid argument");
        return VSC_ERR_INVALID;
    }
    *n_out = 0;
    const bool ip = idx->metric == VSC_METRIC_INNER_PRODUCT;
    if (final_radius) *final_radius = ip ? -1e10f : 1e10f;
    if (nq == 0 || idx->ntotal == 0) return VSC_OK;
    VSC_HIP(hipSetDevice(idx->device));
    float* qp = nullptr;
    VSC_TRY(pack_queries(idx, q, nq, q_mem, &qp, idx->prefilter));
    VSC_TRY(i8_prepare(idx));
    const int64_t cap_max = nq * idx->ntotal + 1024;  // the whole score matrix always fits
    int64_t cap = idx->hit_cap_user;
    if (cap <= 0) cap = std::max(std::max<int64_t>(32 * idx->ntotal, 2 * K) + 2 * K + 1024, idx->hit_cap_learned);
    cap = std::min<int64_t>(cap, cap_max);
    SelectCtl h;
    bool allow_i8 = i8_usable(idx);
    for (;;) {
        bool used_i8 = false;
        VSC_TRY(ensure_hit_buffers(idx, cap));
        // initial radius -1e10 (IP) / +1e10 (L2) -> -1e10 in score space either way (vsc/index.py:146)
        VSC_TRY(init_ctl(idx, seeded ? (ip ? radius0 : -radius0) : -1e10f));
        SelectCtl* ctl = idx->ws.ctl.as<SelectCtl>();
        // exponential_query_iterator: 32, 64, ... doubling while bs < 20000
        int64_t bs = seeded ? 32768 : 32, i0 = 0;
        while (i0 < nq) {
            const int64_t i1 = std::min(nq, i0 + bs);
            // (seeded: the radius is already near its final value -- the expected density is that of the whole search)
            const double seen = seeded ? (double)nq : (double)i0;
            // After i0 rows the radius sits near the K-th best of i0 * ntotal scores, so about
            // K / (i0 * ntotal) of this batch's pairs are hits.  While that density is high the
            // exact kernel is cheaper than pre-filtering and re-scoring nearly everything
            // (exact: ~7.5 ps per pair; re-scoring: ~0.5 ns per candidate; measured optimum near 2 % with the segment-wise exact stage, 5 % with the sorted one).
            // (option "density_hint": the caller knows the batch's expected hit density better than K / (rows x refs) says --
            // the sharded schedule's budgets K are several times the hits it expects, which sent int8 batches to fp16)
            const double dens = idx->density_hint > 0.0 ? idx->density_hint
                                                         : (seen > 0 ? (double)K / (seen * (double)idx->ntotal) : 1.0);
            const bool f16 = idx->prefilter_force || (idx->prefilter && (seen > 0 || idx->density_hint > 0.0) && dens < idx->prefilter_density);
            // ... and once it is low enough that the int8 kernel's 4-5x candidates cost less than the fp16 kernel's
            // second half (the bound of 8-bit rows is ~16x looser), the batch runs on int8
            const bool i8 = f16 && allow_i8 && (idx->i8_mode == 2 || dens < idx->i8_density);
            used_i8 |= i8;
            VSC_TRY(enqueue_batch(idx, qp, i0, i1, cap, f16, i8));
            // The re-threshold round is predicated on the device (kept > 2K) and costs twelve launches even when it does
            // nothing.  Behind the LAST batch the control block is read back anyway: the round is enqueued there only
            // if that read says it has work to do (the one-batch calls of the sharded schedule almost never do).
            if (i1 < nq) {
                hipEvent_t stop;
                VSC_TRY(prof_begin(idx, &stop, 3));
                VSC_TRY(enqueue_rethreshold(ctl, idx->ws.hA[0].as<int32_t>(), idx->ws.hA[1].as<int32_t>(),
                                            idx->ws.hA[2].as<float>(), idx->ws.hB[0].as<int32_t>(),
                                            idx->ws.hB[1].as<int32_t>(), idx->ws.hB[2].as<float>(),
                                            (unsigned long long)K, idx->stream));
                VSC_TRY(prof_end(idx, stop, 0.0, 3));
            }
            if (!seeded && bs < 20000) bs *= 2;
            i0 = i1;
        }
        VSC_HIP(hipMemcpyAsync(&h, ctl, sizeof(h), hipMemcpyDeviceToHost, idx->stream));
        VSC_HIP(hipStreamSynchronize(idx->stream));
        if (!h.overflow && h.n > 2ull * (unsigned long long)K) {
            hipEvent_t stop;
            VSC_TRY(prof_begin(idx, &stop, 3));
            VSC_TRY(enqueue_rethreshold(ctl, idx->ws.hA[0].as<int32_t>(), idx->ws.hA[1].as<int32_t>(),
                                        idx->ws.hA[2].as<float>(), idx->ws.hB[0].as<int32_t>(),
                                        idx->ws.hB[1].as<int32_t>(), idx->ws.hB[2].as<float>(),
                                        (unsigned long long)K, idx->stream));
            VSC_TRY(prof_end(idx, stop, 0.0, 3));
            VSC_HIP(hipMemcpyAsync(&h, ctl, sizeof(h), hipMemcpyDeviceToHost, idx->stream));
            VSC_HIP(hipStreamSynchronize(idx->stream));
        }
        VSC_TRY(prof_collect(idx));
        idx->stat_candidates = h.n_cand_total;
        if (!h.overflow) break;
        if (used_i8 && (h.overflow & 2) && idx->i8_mode != 2) {
            // the candidate list overflowed with int8 batches in the schedule: their bound may simply be too loose for
            // these rows -- same buffers, fp16 pre-filter throughout
            allow_i8 = false;
            idx->stat_i8_fallbacks += 1;
            if (idx->debug_i8)
                fprintf(stderr, "[vscmi] int8 batches overflowed the candidate list (cap %lld, candidates so far %llu, tail %llu, "
                        "kept %llu): fp16 pre-filter for this search\n", (long long)cap, h.n_cand_total, h.n_tail, h.n);
            continue;
        }
        // A batch emitted more hits than the buffer holds (heavy score ties keep the radius low).
        // The schedule is deterministic, so simply rerun it with a larger buffer.
        if (idx->hit_cap_user > 0 || cap >= cap_max) {
            set_error("global_topk: kept-hit buffer (%lld entries) overflowed; raise it with "
                      "vsc_index_set_hit_capacity", (long long)cap);
            return VSC_ERR_OVERFLOW;
        }
        cap = std::min<int64_t>(cap * 4, cap_max);
        idx->hit_cap_learned = cap;  // the next search of this handle starts here instead of overflowing again
    }
    if (final_radius) *final_radius = ip ? h.radius : -h.radius;
    const int64_t n = (int64_t)h.n;
    const int64_t m = std::min(n, K);
    if (m > cap_out) {
        *n_out = m;
        set_error("global_topk: output capacity %lld < %lld", (long long)cap_out, (long long)m);
        return VSC_ERR_CAPACITY;
    }
    int32_t *di = out_i, *dj = out_j;
    float* ds = out_s;
    if (out_mem == VSC_MEM_HOST) {
        VSC_TRY(idx->ws.out[0].reserve((size_t)std::max<int64_t>(m, 1) * 4));
        VSC_TRY(idx->ws.out[1].reserve((size_t)std::max<int64_t>(m, 1) * 4));
        VSC_TRY(idx->ws.out[2].reserve((size_t)std::max<int64_t>(m, 1) * 4));
        di = idx->ws.out[0].as<int32_t>();
        dj = idx->ws.out[1].as<int32_t>();
        ds = idx->ws.out[2].as<float>();
    }
    int64_t mm = 0;
    hipEvent_t sort_stop;
    VSC_TRY(prof_begin(idx, &sort_stop, 4));
    if (idx->sort_hits || !ip) {
        VSC_TRY(sort_hits_topk(idx->ws.hA[0].as<int32_t>(), idx->ws.hA[1].as<int32_t>(), idx->ws.hA[2].as<float>(),
                               n, K, nq, idx->ntotal, idx->ws.w0, idx->ws.w1, idx->ws.w2, idx->ws.w3, idx->ws.tmp, di, dj, ds,
                               ip ? 0 : 1, &mm, idx->stream));
    } else {
        // option "sort_hits" = 0 (the column-sharded schedule, vsc2022_amd/dist.py: a batch's hits only join a list that
        // is counted, filtered and sorted ONCE at the end; inner product only): the kept hits as they lie; a
        // list of K entries may be a truncated one (n > K) -- the caller then asks again with a larger budget
        mm = m;
        if (mm > 0) {
            VSC_HIP(hipMemcpyAsync(di, idx->ws.hA[0].p, (size_t)mm * 4, hipMemcpyDeviceToDevice, idx->stream));
            VSC_HIP(hipMemcpyAsync(dj, idx->ws.hA[1].p, (size_t)mm * 4, hipMemcpyDeviceToDevice, idx->stream));
            VSC_HIP(hipMemcpyAsync(ds, idx->ws.hA[2].p, (size_t)mm * 4, hipMemcpyDeviceToDevice, idx->stream));
        }
    }
    VSC_TRY(prof_end(idx, sort_stop, 12.0 * (double)n, 4));  // (row, ref, score) of every kept hit in
    if (out_mem == VSC_MEM_HOST && mm > 0) {
        VSC_HIP(hipMemcpyAsync(out_i, di, (size_t)mm * 4, hipMemcpyDeviceToHost, idx->stream));
        VSC_HIP(hipMemcpyAsync(out_j, dj, (size_t)mm * 4, hipMemcpyDeviceToHost, idx->stream));
        VSC_HIP(hipMemcpyAsync(out_s, ds, (size_t)mm * 4, hipMemcpyDeviceToHost, idx->stream));
    }
    VSC_HIP(hipStreamSynchronize(idx->stream));
    VSC_TRY(prof_collect(idx));
    *n_out = mm;
    return VSC_OK;
}


// The proven route of vsc_index_global_topk (inner product, large query sets).
//
// What the reference returns (vsc/index.py:142-165): range_search_max_results leaves every pair with s > tau_final, where
// tau_final -- the radius after the last re-threshold event -- is the (K+1)-th best score of the ROW PREFIX searched up to
// that event (or -1e10 without an event); then the stable sort and the cut at K.  A prefix holds fewer pairs than the
// whole matrix, so tau_final <= s_(K+1), the (K+1)-th best score of ALL pairs.  Hence, whenever s_K > s_(K+1):
//     {s > tau_final}  contains  {s >= s_K}  = exactly K pairs,  and the first K of the sorted list are those K pairs
// -- the reference's result IS the exact top-K under (score desc, row asc, ref asc), whatever its batch schedule did on
// the way ("What's weak" 1; the query-sharded pipeline rests on the same argument, vsc2022_amd/dist.py).
// The two results differ only when a tie sits on the cut (s_K == s_(K+1)) AND the schedule's last event ends on that very
// score; that case is not decided here: the schedule is replayed (the caller's fall-through).
//
// So: (1) the reference's schedule over a strided SAMPLE of the query rows (<= 4096) gives a radius just below the cut --
// the 1.25 K (sample share)-th best sample score; (2) ALL rows run as steady 32768-row batches from that radius with the
// budget K + 1 (vsc_index_global_topk_seeded's body: pre-filtered from the first row on, re-threshold rule active); (3) the
// run returns every pair above its final radius: with >= K + 1 of them the K-th and (K+1)-th best of the whole matrix are
// known exactly, and s_K > s_(K+1) proves the first K to be the reference's result.  Anything else -- seed too high,
// a tie on the cut, an overflow -- leaves *done false.  What the doubling batches 32 ... 16384 of the schedule cost at
// BASELINE configs[3] (each hands ~K hits to the exact stage at a radius far below the final one): ~280 of 2050 ms.
static int global_topk_proven(vsc_index_t* idx, const float* q, int64_t nq, int q_mem, int64_t K, int32_t* out_i,
                              int32_t* out_j, float* out_s, int64_t cap_out, int out_mem, int64_t* n_out,
                              float* final_radius, bool* done) {
    *done = false;
    const int64_t nr = idx->ntotal;
    const bool force = idx->topk_shortcut == 2;
    if (idx->metric != VSC_METRIC_INNER_PRODUCT || idx->hit_cap_user > 0 || K < 1 || nq < 4 || nr < 1) return VSC_OK;
    const double pairs = (double)nq * (double)nr;
    if ((double)K + 1.0 > (force ? pairs - 1.0 : 0.25 * pairs) || cap_out < K) return VSC_OK;
    if (!force && (nq < 65536 || pairs < 4e10 || nq < 16 * idx->topk_sample_rows)) return VSC_OK;
    VSC_HIP(hipSetDevice(idx->device));
    Workspace& ws = idx->ws;
    // ---- 1. the sample: every stride-th row, the reference's own schedule over them
    const int64_t stride = std::max<int64_t>(2, nq / std::max<int64_t>(2, std::min(idx->topk_sample_rows, nq / 2)));
    const int64_t ns = (nq + stride - 1) / stride;
    VSC_TRY(ws.sample.reserve((size_t)ns * idx->dim * sizeof(float)));
    VSC_HIP(hipMemcpy2DAsync(ws.sample.p, (size_t)idx->dim * sizeof(float), q, (size_t)stride * idx->dim * sizeof(float),
                             (size_t)idx->dim * sizeof(float), (size_t)ns, q_mem == VSC_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice,
                             idx->stream));
    const double share = (double)ns / (double)nq;
    // (x = the sample's share of the K best: the seed is the score below which 1.25 x + 4 sqrt(x) + 8 sample pairs lie -- a
    // quarter of slack for rows that are not like the sample, four standard deviations of the count itself for small K)
    const double x = (double)K * share;
    const int64_t k_tot = std::max<int64_t>(1, (int64_t)std::ceil(1.25 * x + 4.0 * std::sqrt(x) + 8.0));
    const int64_t k_s = (int64_t)std::min((double)ns * (double)nr, std::ceil(2.0 * x + 8.0 * std::sqrt(x)) + 1024.0);
    for (auto& b : ws.sk) VSC_TRY(b.reserve((size_t)std::max<int64_t>(k_s, 1) * 4));
    int64_t n_s = 0;
    float r_s = -1e10f;
    int rc = global_topk_impl(idx, ws.sample.as<float>(), ns, VSC_MEM_DEVICE, k_s, false, 0.0f, ws.sk[0].as<int32_t>(),
                              ws.sk[1].as<int32_t>(), ws.sk[2].as<float>(), k_s, VSC_MEM_DEVICE, &n_s, &r_s);
    if (rc == VSC_ERR_OVERFLOW || rc == VSC_ERR_CAPACITY) return VSC_OK;
    VSC_TRY(rc);
    const unsigned long long cand_sample = idx->stat_candidates;
    float seed = -1e10f;
    if (n_s >= k_tot) {
        float tau = 0.0f;
        VSC_HIP(hipMemcpyAsync(&tau, ws.sk[2].as<float>() + (k_tot - 1), sizeof(float), hipMemcpyDeviceToHost, idx->stream));
        VSC_HIP(hipStreamSynchronize(idx->stream));
        if (tau == tau && std::fabs(tau) < 1e10f) seed = std::nextafterf(tau, -INFINITY);
    } else if (r_s == r_s && std::fabs(r_s) < 1e10f) {
        seed = r_s;  // (fewer sample hits than asked for: everything above the sample's own final radius)
    }
    // ---- 2. all rows as steady batches from the seed, budget K + 1
    for (auto& b : ws.tk) VSC_TRY(b.reserve((size_t)(K + 1) * 4));
    int64_t n2 = 0;
    float r2 = -1e10f;
    rc = global_topk_impl(idx, q, nq, q_mem, K + 1, true, seed, ws.tk[0].as<int32_t>(), ws.tk[1].as<int32_t>(),
                          ws.tk[2].as<float>(), K + 1, VSC_MEM_DEVICE, &n2, &r2);
    idx->stat_candidates += cand_sample;
    idx->last_topk_route = 2;
    if (rc == VSC_ERR_OVERFLOW || rc == VSC_ERR_CAPACITY) return VSC_OK;
    VSC_TRY(rc);
    if (idx->debug_i8)
        fprintf(stderr, "[vscmi] proven top-K route: %lld sample rows (stride %lld), %lld sample hits, seed %.9g (the %lld-th best); "
                "steady run: %lld hits, radius %.9g (K = %lld)\n", (long long)ns, (long long)stride, (long long)n_s, (double)seed,
                (long long)k_tot, (long long)n2, (double)r2, (long long)K);
    if (n2 < K + 1) return VSC_OK;  // seed too high (or ties dropped by an event of the steady run): nothing is proven
    // ---- 3. the proof: s_K > s_(K+1)
    float cut[2] = {0.0f, 0.0f};
    VSC_HIP(hipMemcpyAsync(cut, ws.tk[2].as<float>() + (K - 1), 2 * sizeof(float), hipMemcpyDeviceToHost, idx->stream));
    VSC_HIP(hipStreamSynchronize(idx->stream));
    if (idx->debug_i8) fprintf(stderr, "[vscmi] proven top-K route: s_K = %.9g, s_(K+1) = %.9g\n", (double)cut[0], (double)cut[1]);
    if (!(cut[0] > cut[1])) return VSC_OK;  // a tie on the cut: the schedule's final radius decides -- replay it
    const hipMemcpyKind kind = out_mem == VSC_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    VSC_HIP(hipMemcpyAsync(out_i, ws.tk[0].p, (size_t)K * 4, kind, idx->stream));
    VSC_HIP(hipMemcpyAsync(out_j, ws.tk[1].p, (size_t)K * 4, kind, idx->stream));
    VSC_HIP(hipMemcpyAsync(out_s, ws.tk[2].p, (size_t)K * 4, kind, idx->stream));
    VSC_HIP(hipStreamSynchronize(idx->stream));
    *n_out = K;
    // (the steady run's last radius: every returned hit lies above it.  The reference schedule's own final radius is
    // not computed on this route -- it is not part of what vsc/index.py returns; topk_shortcut = 0 yields it)
    if (final_radius) *final_radius = r2;
    idx->last_topk_route = 1;
    *done = true;
    return VSC_OK;
}

extern "C" {

int vsc_index_global_topk(vsc_index_t* idx, const float* q, int64_t nq, int q_mem, int64_t K,
                          int32_t* out_i, int32_t* out_j, float* out_s, int64_t cap_out, int out_mem,
                          int64_t* n_out, float* final_radius) {
    if (idx) idx->last_topk_route = 0;
    if (idx && idx->topk_shortcut && nq > 0 && q && n_out && K >= 0 && out_i && out_j && out_s) {
        bool done = false;
        VSC_TRY(global_topk_proven(idx, q, nq, q_mem, K, out_i, out_j, out_s, cap_out, out_mem, n_out, final_radius, &done));
        if (done) return VSC_OK;
    }
    return global_topk_impl(idx, q, nq, q_mem, K, false, 0.0f, out_i, out_j, out_s, cap_out, out_mem, n_out, final_radius);
}

int vsc_index_global_topk_seeded(vsc_index_t* idx, const float* q, int64_t nq, int q_mem, int64_t K, float radius0,
                                 int32_t* out_i, int32_t* out_j, float* out_s, int64_t cap_out, int out_mem,
                                 int64_t* n_out, float* final_radius) {
    if (!(radius0 == radius0) || std::fabs(radius0) > 1e10f) {
        set_error("vsc_index_global_topk_seeded: the seed radius must be a finite score (got %g)", (double)radius0);
        return VSC_ERR_INVALID;
    }
    return global_topk_impl(idx, q, nq, q_mem, K, true, radius0, out_i, out_j, out_s, cap_out, out_mem, n_out, final_radius);
}

int vsc_index_candidates(vsc_index_t* idx, const float* q, int64_t nq, int q_mem, int64_t K,
                         const int32_t* row2q, const int32_t* row2r, int32_t* out_q, int32_t* out_r,
                         float* out_s, int64_t cap_out, int64_t* n_pairs, int64_t* n_hits) {
    if (!idx || nq < 0 || K < 0 || !n_pairs || (nq > 0 && (!q || !row2q || !row2r))) {
        set_error("vsc_index_candidates: invalid argument");
        return VSC_ERR_INVALID;
    }
    *n_pairs = 0;
    if (n_hits) *n_hits = 0;
    if (idx->metric != VSC_METRIC_INNER_PRODUCT) {
        set_error("vsc_index_candidates: max aggregation needs a larger-is-better metric (inner product)");
        return VSC_ERR_INVALID;
    }
    if (nq == 0 || idx->ntotal == 0 || K == 0) return VSC_OK;
    VSC_HIP(hipSetDevice(idx->device));
    // 1. the score-sorted top-K hits stay in HBM
    const int64_t hcap = std::max<int64_t>(1, std::min<int64_t>(K, nq * idx->ntotal));
    for (int c = 0; c < 3; ++c) VSC_TRY(idx->cand[c].reserve((size_t)hcap * 4));
    int64_t n = 0;
    float radius = 0.0f;
    VSC_TRY(vsc_index_global_topk(idx, q, nq, q_mem, K, idx->cand[0].as<int32_t>(), idx->cand[1].as<int32_t>(),
                                  idx->cand[2].as<float>(), hcap, VSC_MEM_DEVICE, &n, &radius));
    if (n_hits) *n_hits = n;
    if (n == 0) return VSC_OK;
    // 2. (query video, ref video) max aggregation on the device
    Workspace& ws = idx->ws;
    VSC_TRY(ws.maps0.reserve((size_t)nq * 4));
    VSC_TRY(ws.maps1.reserve((size_t)idx->ntotal * 4));
    VSC_HIP(hipMemcpyAsync(ws.maps0.p, row2q, (size_t)nq * 4, hipMemcpyHostToDevice, idx->stream));
    VSC_HIP(hipMemcpyAsync(ws.maps1.p, row2r, (size_t)idx->ntotal * 4, hipMemcpyHostToDevice, idx->stream));
    for (int c = 0; c < 3; ++c) VSC_TRY(ws.out[c].reserve((size_t)n * 4));
    VSC_TRY(ws.out[3].reserve((size_t)n * 8));
    int64_t np = 0;
    VSC_TRY(pair_max_device(idx->cand[0].as<int32_t>(), idx->cand[1].as<int32_t>(), idx->cand[2].as<float>(), n,
                            ws.maps0.as<int32_t>(), ws.maps1.as<int32_t>(), nq, idx->ntotal, ws.w0, ws.w1, ws.w2, ws.w3, ws.tmp,
                            ws.cnt, ws.out[0].as<int32_t>(), ws.out[1].as<int32_t>(), ws.out[2].as<float>(),
                            ws.out[3].as<int64_t>(), n, &np, idx->stream));
    *n_pairs = np;
    if (np > cap_out) {
        set_error("vsc_index_candidates: output capacity %lld < %lld pairs", (long long)cap_out, (long long)np);
        return VSC_ERR_CAPACITY;
    }
    if (np > 0) {
        VSC_HIP(hipMemcpyAsync(out_q, ws.out[0].p, (size_t)np * 4, hipMemcpyDeviceToHost, idx->stream));
        VSC_HIP(hipMemcpyAsync(out_r, ws.out[1].p, (size_t)np * 4, hipMemcpyDeviceToHost, idx->stream));
        VSC_HIP(hipMemcpyAsync(out_s, ws.out[2].p, (size_t)np * 4, hipMemcpyDeviceToHost, idx->stream));
    }
    VSC_HIP(hipStreamSynchronize(idx->stream));
    return VSC_OK;
}

int vsc_index_range_search(vsc_index_t* idx, const float* q, int64_t nq, int q_mem, float radius,
                           int64_t* lims, float* D, int64_t* I, int64_t cap_out, int64_t* n_out) {
    if (!idx || nq < 0 || !lims || !n_out || (nq > 0 && !q)) {
        set_error("vsc_index_range_search: invalid argument");
        return VSC_ERR_INVALID;
    }
    *n_out = 0;
    for (int64_t i = 0; i <= nq; ++i) lims[i] = 0;
    if (nq == 0 || idx->ntotal == 0) return VSC_OK;
    const bool ip = idx->metric == VSC_METRIC_INNER_PRODUCT;
    VSC_HIP(hipSetDevice(idx->device));
    float* qp = nullptr;
    VSC_TRY(pack_queries(idx, q, nq, q_mem, &qp, idx->prefilter));
    VSC_TRY(i8_prepare(idx));
    int64_t cap = idx->hit_cap_user > 0 ? idx->hit_cap_user : std::min<int64_t>(nq * idx->ntotal, (int64_t)1 << 28);
    cap = std::max<int64_t>(cap, 1024);
    VSC_TRY(ensure_hit_buffers(idx, cap));
    VSC_TRY(init_ctl(idx, ip ? radius : -radius));
    SelectCtl* ctl = idx->ws.ctl.as<SelectCtl>();
    const int64_t step = 32768;
    // fixed radius: the pre-filter is used throughout (a radius so low that most pairs pass would
    // overflow the hit capacity on either route)
    for (int64_t i0 = 0; i0 < nq; i0 += step)
        VSC_TRY(enqueue_batch(idx, qp, i0, std::min(nq, i0 + step), cap, idx->prefilter, idx->i8_mode == 2));
    SelectCtl h;
    VSC_HIP(hipMemcpyAsync(&h, ctl, sizeof(h), hipMemcpyDeviceToHost, idx->stream));
    VSC_HIP(hipStreamSynchronize(idx->stream));
    VSC_TRY(prof_collect(idx));
    if (h.overflow) {
        set_error("range_search: more than %lld hits; raise vsc_index_set_hit_capacity", (long long)cap);
        return VSC_ERR_OVERFLOW;
    }
    const int64_t n = (int64_t)h.n;
    *n_out = n;
    if (n == 0) return VSC_OK;
    // rows ascending, refs ascending (reuse B as the sorted target)
    VSC_TRY(sort_hits_rowcol(idx->ws.hA[0].as<int32_t>(), idx->ws.hA[1].as<int32_t>(), idx->ws.hA[2].as<float>(), n,
                             idx->ws.w0, idx->ws.w1, idx->ws.w2, idx->ws.w3, idx->ws.tmp, idx->ws.hB[0].as<int32_t>(),
                             idx->ws.hB[1].as<int32_t>(), idx->ws.hB[2].as<float>(), ip ? 0 : 1, idx->stream));
    std::vector<int32_t> hi((size_t)n);
    VSC_HIP(hipMemcpyAsync(hi.data(), idx->ws.hB[0].p, (size_t)n * 4, hipMemcpyDeviceToHost, idx->stream));
    VSC_HIP(hipStreamSynchronize(idx->stream));
    for (int64_t x = 0; x < n; ++x) lims[hi[(size_t)x] + 1] += 1;
    for (int64_t i = 0; i < nq; ++i) lims[i + 1] += lims[i];
    if (!D || !I || cap_out < n) {
        if (D || I) {
            set_error("range_search: output capacity %lld < %lld", (long long)cap_out, (long long)n);
            return VSC_ERR_CAPACITY;
        }
        return VSC_OK;  // size query
    }
    std::vector<int32_t> hj((size_t)n);
    VSC_HIP(hipMemcpyAsync(hj.data(), idx->ws.hB[1].p, (size_t)n * 4, hipMemcpyDeviceToHost, idx->stream));
    VSC_HIP(hipMemcpyAsync(D, idx->ws.hB[2].p, (size_t)n * 4, hipMemcpyDeviceToHost, idx->stream));
    VSC_HIP(hipStreamSynchronize(idx->stream));
    for (int64_t x = 0; x < n; ++x) I[x] = hj[(size_t)x];
    return VSC_OK;
}


}  // extern "C"
