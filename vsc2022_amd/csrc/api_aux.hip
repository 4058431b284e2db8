// C ABI of libvscmi.so, part 4: the entry points that own no index handle (vsc_pair_max, vsc_row_normalize) and the
// Temporal-Network localisation contexts (vsc_tn_*).
#include "api_internal.h"

extern "C" {

// ------------------------------------------------------------------ stand-alone device ops

struct DeviceCtx {
    hipStream_t stream = nullptr;      // own_stream, or the caller's (vsc_set_aux_stream)
    hipStream_t own_stream = nullptr;
    Workspace ws;
    std::mutex mu;
};
static DeviceCtx* device_ctx(int device) {
    static std::mutex g_mu;
    static std::vector<DeviceCtx*> ctxs;
    std::lock_guard<std::mutex> lk(g_mu);
    if ((int)ctxs.size() <= device) ctxs.resize(device + 1, nullptr);
    if (!ctxs[device]) {
        DeviceCtx* c = new DeviceCtx();
        if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
            delete c;
            return nullptr;
        }
        c->stream = c->own_stream;
        ctxs[device] = c;
    }
    return ctxs[device];
}

// fetch `bytes` of a caller array into device memory (no copy if already there)
static int to_device(const void* p, size_t bytes, int mem, DevBuf& buf, const void** out, hipStream_t s) {
    if (mem == VSC_MEM_DEVICE) {
        *out = p;
        return VSC_OK;
    }
    VSC_TRY(buf.reserve(std::max<size_t>(bytes, 16)));
    if (bytes) VSC_HIP(hipMemcpyAsync(buf.p, p, bytes, hipMemcpyHostToDevice, s));
    *out = buf.p;
    return VSC_OK;
}

int vsc_set_aux_stream(int device, void* hip_stream, int own) {
    VSC_TRY(check_device(device));
    VSC_HIP(hipSetDevice(device));
    DeviceCtx* c = device_ctx(device);
    if (!c) {
        set_error("vsc_set_aux_stream: device context unavailable");
        return VSC_ERR_HIP;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->stream == c->own_stream) VSC_HIP(hipStreamSynchronize(c->own_stream));
    else (void)hipDeviceSynchronize();  // (a caller's stream may be gone: never touched again, see vsc_index_set_stream)
    c->stream = own ? c->own_stream : (hipStream_t)hip_stream;
    return VSC_OK;
}

int vsc_pair_max(const int32_t* hit_i, const int32_t* hit_j, const float* hit_s, int64_t n,
                 int hits_mem, const int32_t* row2q, int64_t nq_rows, const int32_t* row2r,
                 int64_t nr_rows, int maps_mem, int32_t* out_q, int32_t* out_r, float* out_s,
                 int64_t* out_first, int64_t cap, int out_mem, int64_t* n_pairs, int device) {
    if (n < 0 || !n_pairs || (n > 0 && (!hit_i || !hit_j || !hit_s || !row2q || !row2r))) {
        set_error("vsc_pair_max: invalid argument");
        return VSC_ERR_INVALID;
    }
    *n_pairs = 0;
    if (n == 0) return VSC_OK;
    VSC_TRY(check_device(device));
    VSC_HIP(hipSetDevice(device));
    DeviceCtx* c = device_ctx(device);
    if (!c) {
        set_error("vsc_pair_max: cannot create device context");
        return VSC_ERR_HIP;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    Workspace& ws = c->ws;
    const void *di, *dj, *ds, *dq, *dr;
    VSC_TRY(to_device(hit_i, (size_t)n * 4, hits_mem, ws.hA[0], &di, c->stream));
    VSC_TRY(to_device(hit_j, (size_t)n * 4, hits_mem, ws.hA[1], &dj, c->stream));
    VSC_TRY(to_device(hit_s, (size_t)n * 4, hits_mem, ws.hA[2], &ds, c->stream));
    VSC_TRY(to_device(row2q, (size_t)nq_rows * 4, maps_mem, ws.maps0, &dq, c->stream));
    VSC_TRY(to_device(row2r, (size_t)nr_rows * 4, maps_mem, ws.maps1, &dr, c->stream));
    int32_t *oq = out_q, *orr = out_r;
    float* os = out_s;
    int64_t* of = out_first;
    const int64_t ocap = out_mem == VSC_MEM_HOST ? n : cap;
    if (out_mem == VSC_MEM_HOST) {
        VSC_TRY(ws.out[0].reserve((size_t)n * 4));
        VSC_TRY(ws.out[1].reserve((size_t)n * 4));
        VSC_TRY(ws.out[2].reserve((size_t)n * 4));
        VSC_TRY(ws.out[3].reserve((size_t)n * 8));
        oq = ws.out[0].as<int32_t>();
        orr = ws.out[1].as<int32_t>();
        os = ws.out[2].as<float>();
        of = ws.out[3].as<int64_t>();
    }
    int64_t np = 0;
    AuxTimer tm;
    tm.begin(0, c->stream);
    VSC_TRY(pair_max_device((const int32_t*)di, (const int32_t*)dj, (const float*)ds, n, (const int32_t*)dq,
                            (const int32_t*)dr, nq_rows, nr_rows, ws.w0, ws.w1, ws.w2, ws.w3, ws.tmp, ws.cnt, oq, orr, os, of,
                            ocap, &np, c->stream));
    tm.end(12.0 * (double)n + 20.0 * (double)np, c->stream);  // hits in, (q, r, score, first hit) per pair out
    *n_pairs = np;
    if (out_mem == VSC_MEM_HOST) {
        if (np > cap) {
            set_error("vsc_pair_max: output capacity %lld < %lld pairs", (long long)cap, (long long)np);
            return VSC_ERR_CAPACITY;
        }
        VSC_HIP(hipMemcpyAsync(out_q, oq, (size_t)np * 4, hipMemcpyDeviceToHost, c->stream));
        VSC_HIP(hipMemcpyAsync(out_r, orr, (size_t)np * 4, hipMemcpyDeviceToHost, c->stream));
        VSC_HIP(hipMemcpyAsync(out_s, os, (size_t)np * 4, hipMemcpyDeviceToHost, c->stream));
        if (out_first) VSC_HIP(hipMemcpyAsync(out_first, of, (size_t)np * 8, hipMemcpyDeviceToHost, c->stream));
    }
    VSC_HIP(hipStreamSynchronize(c->stream));
    tm.collect();
    return VSC_OK;
}

int vsc_sort_hits(const int32_t* hit_i, const int32_t* hit_j, const float* hit_s, int64_t n, int hits_mem, int64_t max_row,
                  int64_t max_ref, int32_t* out_i, int32_t* out_j, float* out_s, int out_mem, int device) {
    if (n < 0 || (n > 0 && (!hit_i || !hit_j || !hit_s || !out_i || !out_j || !out_s))) {
        set_error("vsc_sort_hits: invalid argument");
        return VSC_ERR_INVALID;
    }
    if (n == 0) return VSC_OK;
    VSC_TRY(check_device(device));
    VSC_HIP(hipSetDevice(device));
    DeviceCtx* c = device_ctx(device);
    if (!c) {
        set_error("vsc_sort_hits: cannot create device context");
        return VSC_ERR_HIP;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    Workspace& ws = c->ws;
    const void *di, *dj, *ds;
    VSC_TRY(to_device(hit_i, (size_t)n * 4, hits_mem, ws.hA[0], &di, c->stream));
    VSC_TRY(to_device(hit_j, (size_t)n * 4, hits_mem, ws.hA[1], &dj, c->stream));
    VSC_TRY(to_device(hit_s, (size_t)n * 4, hits_mem, ws.hA[2], &ds, c->stream));
    int32_t *oi = out_i, *oj = out_j;
    float* os = out_s;
    if (out_mem == VSC_MEM_HOST) {
        for (int k = 0; k < 3; ++k) VSC_TRY(ws.out[k].reserve((size_t)n * 4));
        oi = ws.out[0].as<int32_t>();
        oj = ws.out[1].as<int32_t>();
        os = ws.out[2].as<float>();
    }
    const int64_t all = (int64_t)1 << 31;
    int64_t m = 0;
    VSC_TRY(sort_hits_topk((const int32_t*)di, (const int32_t*)dj, (const float*)ds, n, n, max_row > 0 ? max_row : all,
                           max_ref > 0 ? max_ref : all, ws.w0, ws.w1, ws.w2, ws.w3, ws.tmp, oi, oj, os, 0, &m, c->stream));
    if (out_mem == VSC_MEM_HOST) {
        VSC_HIP(hipMemcpyAsync(out_i, oi, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        VSC_HIP(hipMemcpyAsync(out_j, oj, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        VSC_HIP(hipMemcpyAsync(out_s, os, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    }
    VSC_HIP(hipStreamSynchronize(c->stream));
    return VSC_OK;
}

// One level of an order statistic over unsorted scores that are spread over ranks (include/vscmi.h).  Device pointers
// only; on a caller's stream (vsc_set_aux_stream) the work is merely enqueued -- the caller's next operation on that
// stream (the all-reduce of the histogram) is ordered behind it --, on the library's own stream the call waits.
int vsc_score_histogram(const float* scores, int64_t n, const int64_t* state, int shift, int64_t* hist, int device) {
    if (n < 0 || (n > 0 && !scores) || !state || !hist || (shift != 0 && shift != 8 && shift != 16 && shift != 24)) {
        set_error("vsc_score_histogram: invalid argument");
        return VSC_ERR_INVALID;
    }
    VSC_TRY(check_device(device));
    VSC_HIP(hipSetDevice(device));
    DeviceCtx* c = device_ctx(device);
    if (!c) {
        set_error("vsc_score_histogram: cannot create device context");
        return VSC_ERR_HIP;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    VSC_TRY(launch_score_hist(scores, (long long)n, reinterpret_cast<const long long*>(state), shift,
                              reinterpret_cast<long long*>(hist), c->stream));
    if (c->stream == c->own_stream) VSC_HIP(hipStreamSynchronize(c->stream));
    return VSC_OK;
}

int vsc_score_pick(const int64_t* hist, int64_t* state, int shift, int device) {
    if (!hist || !state || (shift != 0 && shift != 8 && shift != 16 && shift != 24)) {
        set_error("vsc_score_pick: invalid argument");
        return VSC_ERR_INVALID;
    }
    VSC_TRY(check_device(device));
    VSC_HIP(hipSetDevice(device));
    DeviceCtx* c = device_ctx(device);
    if (!c) {
        set_error("vsc_score_pick: cannot create device context");
        return VSC_ERR_HIP;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    VSC_TRY(launch_score_pick(reinterpret_cast<const long long*>(hist), reinterpret_cast<long long*>(state), shift, c->stream));
    if (c->stream == c->own_stream) VSC_HIP(hipStreamSynchronize(c->stream));
    return VSC_OK;
}

int vsc_filter_hits(const int32_t* hit_i, const int32_t* hit_j, const float* hit_s, int64_t n, float radius, int32_t* out_i,
                    int32_t* out_j, float* out_s, int64_t* n_out, int device) {
    if (n < 0 || !n_out || (n > 0 && (!hit_i || !hit_j || !hit_s || !out_i || !out_j || !out_s)) || !(radius == radius)) {
        set_error("vsc_filter_hits: invalid argument");
        return VSC_ERR_INVALID;
    }
    *n_out = 0;
    if (n == 0) return VSC_OK;
    VSC_TRY(check_device(device));
    VSC_HIP(hipSetDevice(device));
    DeviceCtx* c = device_ctx(device);
    if (!c) {
        set_error("vsc_filter_hits: cannot create device context");
        return VSC_ERR_HIP;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    VSC_TRY(c->ws.cnt.reserve(sizeof(unsigned long long)));
    VSC_TRY(launch_filter_hits(hit_i, hit_j, hit_s, (long long)n, radius, out_i, out_j, out_s, c->ws.cnt.as<unsigned long long>(),
                               c->stream));
    unsigned long long kept = 0;
    VSC_HIP(hipMemcpyAsync(&kept, c->ws.cnt.p, sizeof(kept), hipMemcpyDeviceToHost, c->stream));
    VSC_HIP(hipStreamSynchronize(c->stream));
    *n_out = (int64_t)kept;
    return VSC_OK;
}

int vsc_argsort_scores(const float* scores, int64_t n, int mem, int32_t* perm, int perm_mem, int device) {
    if (n < 0 || (n > 0 && (!scores || !perm))) {
        set_error("vsc_argsort_scores: invalid argument");
        return VSC_ERR_INVALID;
    }
    if (n == 0) return VSC_OK;
    VSC_TRY(check_device(device));
    VSC_HIP(hipSetDevice(device));
    DeviceCtx* c = device_ctx(device);
    if (!c) {
        set_error("vsc_argsort_scores: cannot create device context");
        return VSC_ERR_HIP;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    Workspace& ws = c->ws;
    const void* ds;
    VSC_TRY(to_device(scores, (size_t)n * 4, mem, ws.hA[2], &ds, c->stream));
    const int32_t* p = nullptr;
    VSC_TRY(argsort_scores_desc((const float*)ds, n, ws.w0, ws.w1, ws.w2, ws.w3, ws.tmp, &p, c->stream));
    VSC_HIP(hipMemcpyAsync(perm, p, (size_t)n * 4, perm_mem == VSC_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice,
                           c->stream));
    VSC_HIP(hipStreamSynchronize(c->stream));
    return VSC_OK;
}

int vsc_merge_topk(const float* scores, const int64_t* ids, int64_t nq, int m, int k, float* out_s, int64_t* out_ids,
                   int device) {
    if (nq < 0 || m <= 0 || k <= 0 || m > 1024 || k > m || (nq > 0 && (!scores || !ids || !out_s || !out_ids))) {
        set_error("vsc_merge_topk: invalid argument (1 <= k <= m <= 1024 candidates per row)");
        return VSC_ERR_INVALID;
    }
    if (nq == 0) return VSC_OK;
    VSC_TRY(check_device(device));
    VSC_HIP(hipSetDevice(device));
    DeviceCtx* c = device_ctx(device);
    if (!c) {
        set_error("vsc_merge_topk: cannot create device context");
        return VSC_ERR_HIP;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    VSC_TRY(launch_merge_topk(scores, reinterpret_cast<const long long*>(ids), (long long)nq, m, k, out_s,
                              reinterpret_cast<long long*>(out_ids), c->stream));
    VSC_HIP(hipStreamSynchronize(c->stream));
    return VSC_OK;
}

int vsc_row_normalize(const float* x, int64_t n, int dim, int x_mem, float* out, int out_mem, int device) {
    if (n < 0 || dim <= 0 || (n > 0 && (!x || !out))) {
        set_error("vsc_row_normalize: invalid argument");
        return VSC_ERR_INVALID;
    }
    if (n == 0) return VSC_OK;
    VSC_TRY(check_device(device));
    VSC_HIP(hipSetDevice(device));
    DeviceCtx* c = device_ctx(device);
    if (!c) {
        set_error("vsc_row_normalize: cannot create device context");
        return VSC_ERR_HIP;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    const void* dx;
    VSC_TRY(to_device(x, (size_t)n * dim * 4, x_mem, c->ws.stage, &dx, c->stream));
    float* dout = out;
    if (out_mem == VSC_MEM_HOST) {
        VSC_TRY(c->ws.out[0].reserve((size_t)n * dim * 4));
        dout = c->ws.out[0].as<float>();
    }
    VSC_TRY(launch_row_normalize((const float*)dx, n, dim, dout, c->stream));
    if (out_mem == VSC_MEM_HOST)
        VSC_HIP(hipMemcpyAsync(out, dout, (size_t)n * dim * 4, hipMemcpyDeviceToHost, c->stream));
    VSC_HIP(hipStreamSynchronize(c->stream));
    return VSC_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------ TN launches

// Split the pairs of one call into launches by LDS footprint and run them.  `base` carries
// everything except the per-launch fields.  In forward_sim mode (base.sims_in set) tiles are read
// in place; otherwise they live in LDS when they fit the launch's budget and spill to `slab`.
// Pairs whose working state does not fit LDS, or whose node / frame indices do not fit 16 bits
// (query videos beyond ~1000 frames at the default parameters, references beyond 32767), run
// from an HBM state slab with 32-bit indices, in chunks of bounded memory.
static int tn_run_buckets(TnPairArgs base, const std::vector<int32_t>& lqs, const std::vector<int32_t>& lrs,
                          DevBuf& d_work, DevBuf& slab, hipStream_t stream) {
    const int64_t n_pairs = (int64_t)lqs.size();
    const int ms = base.prm.tn_max_step > 1 ? base.prm.tn_max_step : 1;
    const int topc = base.prm.tn_top_k;
    const bool fused = base.sims_in == nullptr;
    constexpr size_t LDS_STATE_MAX = 150 * 1024;
    constexpr int64_t BIG_CHUNK_BYTES = (int64_t)4 << 30;  // state + similarity slab of one launch of the HBM route
    struct Bucket { int max_lq; int64_t max_tile; std::vector<int32_t> work; int seen_lq; int64_t seen_tile; };
    Bucket buckets[4] = {{64, 4096, {}, 0, 0}, {256, 24576, {}, 0, 0}, {0x7fffffff, 0, {}, 0, 0}, {0x7fffffff, 0, {}, 0, 0}};
    for (int64_t p = 0; p < n_pairs; ++p) {
        const int64_t lq = lqs[(size_t)p], lr = lrs[(size_t)p];
        if (lq * std::max<int64_t>(1, topc) * ms * topc > 0x7fff0000LL || lq * lr > ((int64_t)1 << 40)) {
            set_error("TN: a %lld x %lld frame pair is beyond the supported size", (long long)lq, (long long)lr);
            return VSC_ERR_INVALID;
        }
        int b = 2;
        if (lq * topc + 1 > 32767 || lr > 32767 || tn_state_bytes_host((int)std::max<int64_t>(lq, 1), topc, ms) > LDS_STATE_MAX) b = 3;
        else if (lq <= 64 && (!fused || lq * lr <= 4096)) b = 0;
        else if (lq <= 256 && (!fused || lq * lr <= 24576)) b = 1;
        buckets[b].work.push_back((int32_t)p);
        buckets[b].seen_lq = std::max<int>(buckets[b].seen_lq, (int)lq);
        buckets[b].seen_tile = std::max<int64_t>(buckets[b].seen_tile, lq * lr);
    }
    VSC_TRY(d_work.reserve((size_t)std::max<int64_t>(n_pairs, 1) * 4));
    DevBuf big_state;  // HBM route only; released on return
    struct Release { DevBuf& b; ~Release() { b.release(); } } release_big{big_state};
    int64_t woff = 0;
    for (int b = 0; b < 4; ++b) {
        Bucket& B = buckets[b];
        if (B.work.empty()) continue;
        const bool big = b == 3;
        const int max_lq = std::max(1, B.seen_lq);
        const size_t state = tn_state_bytes_host(max_lq, topc, ms, big ? 4 : 2);
        int tile_floats = 0;
        int64_t slab_floats = 0;
        if (fused) {
            if (b < 2) tile_floats = (int)std::min<int64_t>(B.seen_tile, B.max_tile);
            else slab_floats = (B.seen_tile + 63) / 64 * 64;
        }
        // pairs per launch: everything, or as many as the HBM route's memory bound allows
        int64_t per_launch = (int64_t)B.work.size();
        if (big) per_launch = std::max<int64_t>(1, std::min<int64_t>(per_launch, BIG_CHUNK_BYTES / (int64_t)(state + (size_t)slab_floats * 4)));
        if (slab_floats) VSC_TRY(slab.reserve((size_t)slab_floats * 4 * (size_t)per_launch));
        if (big) VSC_TRY(big_state.reserve(state * (size_t)per_launch));
        const size_t lds = big ? 0 : state + (size_t)tile_floats * 4;
        for (int64_t c0 = 0; c0 < (int64_t)B.work.size(); c0 += per_launch) {
            const int64_t cn = std::min<int64_t>(per_launch, (int64_t)B.work.size() - c0);
            int32_t* dwork = d_work.as<int32_t>() + woff;
            VSC_HIP(hipMemcpyAsync(dwork, B.work.data() + c0, (size_t)cn * 4, hipMemcpyHostToDevice, stream));
            TnPairArgs a = base;
            a.work = dwork;
            a.n_work = (int)cn;
            a.max_lq = max_lq;
            a.lds_tile_floats = tile_floats;
            a.slab = slab.as<float>();
            a.slab_floats = slab_floats;
            a.state = big ? big_state.as<char>() : nullptr;
            a.state_bytes = (int64_t)state;
            // algorithmic bytes of the launch: the descriptor rows of every pair once (fused) or its matrix
            // (forward_sim), + the boxes out
            double bytes = 0.0;
            for (int64_t x = c0; x < c0 + cn; ++x) {
                const int32_t p = B.work[(size_t)x];
                const double lq = lqs[(size_t)p], lr = lrs[(size_t)p];
                bytes += fused ? 4.0 * base.dpad * (lq + lr) : 4.0 * lq * lr;
                bytes += 4.0 + 20.0 * VSC_TN_MAX_BOXES;
            }
            AuxTimer tm;
            tm.begin(1, stream);
            VSC_TRY(launch_tn_pairs(a, lds, stream));
            tm.end(bytes, stream);
            VSC_HIP(hipStreamSynchronize(stream));  // B.work (host), the slab and the state are reused
            tm.collect();
            woff += cn;
        }
    }
    return VSC_OK;
}

// ------------------------------------------------------------------------ TN context

struct vsc_tn_ctx {
    int device = 0, dim = 0, dpad = 0;
    int64_t n_qvid = 0, n_rvid = 0;
    std::vector<int64_t> q_off, r_off;  // host copies
    DevBuf qfeat, rfeat, d_qoff, d_roff;
    DevBuf d_pq, d_pr, d_work, d_nbox, d_boxes, d_bmax, slab, sims;
    Workspace ws;
    hipStream_t stream = nullptr;      // own_stream, or the caller's (vsc_tn_set_stream)
    hipStream_t own_stream = nullptr;
};

extern "C" {

int vsc_tn_set_stream(vsc_tn_ctx_t* c, void* hip_stream, int own) {
    if (!c) {
        set_error("vsc_tn_set_stream: invalid argument");
        return VSC_ERR_INVALID;
    }
    VSC_HIP(hipSetDevice(c->device));
    if (c->stream == c->own_stream) VSC_HIP(hipStreamSynchronize(c->own_stream));
    else (void)hipDeviceSynchronize();  // (a caller's stream may be gone: never touched again, see vsc_index_set_stream)
    c->stream = own ? c->own_stream : (hipStream_t)hip_stream;
    return VSC_OK;
}

int vsc_tn_create(const float* qfeat, const int64_t* q_off, int64_t n_qvid, const float* rfeat,
                  const int64_t* r_off, int64_t n_rvid, int dim, int feat_mem, int device,
                  vsc_tn_ctx_t** out) {
    if (!out || dim <= 0 || n_qvid < 0 || n_rvid < 0 || !q_off || !r_off) {
        set_error("vsc_tn_create: invalid argument");
        return VSC_ERR_INVALID;
    }
    VSC_TRY(check_device(device));
    VSC_HIP(hipSetDevice(device));
    vsc_tn_ctx* c = new vsc_tn_ctx();
    c->device = device;
    c->dim = dim;
    c->dpad = round_up(dim, K_PAD);
    c->n_qvid = n_qvid;
    c->n_rvid = n_rvid;
    c->q_off.assign(q_off, q_off + n_qvid + 1);
    c->r_off.assign(r_off, r_off + n_rvid + 1);
    int rc = VSC_OK;
    auto fail = [&](int code) {
        vsc_tn_destroy(c);
        return code;
    };
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) == hipSuccess) c->stream = c->own_stream;
    if (!c->own_stream) {
        set_error("hipStreamCreate failed");
        delete c;
        return VSC_ERR_HIP;
    }
    const int64_t nq = c->q_off.back(), nr = c->r_off.back();
    // +32 rows of slack: the 32-row MFMA blocks of the last video read past its end
    const int64_t q_rows = round_up64(nq + 32, ROW_PAD), r_rows = round_up64(nr + 32, ROW_PAD);
    if ((rc = c->qfeat.reserve((size_t)q_rows * c->dpad * 4)) != VSC_OK) return fail(rc);
    if ((rc = c->rfeat.reserve((size_t)r_rows * c->dpad * 4)) != VSC_OK) return fail(rc);
    if ((rc = pack_into(qfeat, nq, dim, feat_mem, c->qfeat.as<float>(), q_rows, c->dpad, c->ws, c->stream)) != VSC_OK) return fail(rc);
    if ((rc = pack_into(rfeat, nr, dim, feat_mem, c->rfeat.as<float>(), r_rows, c->dpad, c->ws, c->stream)) != VSC_OK) return fail(rc);
    if ((rc = c->d_qoff.reserve((size_t)(n_qvid + 1) * 8)) != VSC_OK) return fail(rc);
    if ((rc = c->d_roff.reserve((size_t)(n_rvid + 1) * 8)) != VSC_OK) return fail(rc);
    if (hipMemcpyAsync(c->d_qoff.p, c->q_off.data(), (size_t)(n_qvid + 1) * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipMemcpyAsync(c->d_roff.p, c->r_off.data(), (size_t)(n_rvid + 1) * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) {
        set_error("vsc_tn_create: offset upload failed");
        return fail(VSC_ERR_HIP);
    }
    *out = c;
    return VSC_OK;
}

int vsc_tn_set_queries(vsc_tn_ctx_t* c, const float* qfeat, const int64_t* q_off, int64_t n_qvid, int feat_mem) {
    if (!c || n_qvid < 0 || !q_off || (n_qvid > 0 && q_off[n_qvid] > 0 && !qfeat)) {
        set_error("vsc_tn_set_queries: invalid argument");
        return VSC_ERR_INVALID;
    }
    VSC_HIP(hipSetDevice(c->device));
    // everything that can fail (allocation, packing, upload) runs on local state first: the context keeps its old,
    // consistent query side if any of it does, and takes the new offsets only once the device holds the new rows
    std::vector<int64_t> off(q_off, q_off + n_qvid + 1);
    const int64_t nq = off.back();
    const int64_t q_rows = round_up64(nq + 32, ROW_PAD);  // (+32: see vsc_tn_create)
    int rc = c->qfeat.reserve((size_t)q_rows * c->dpad * 4);
    if (rc == VSC_OK) rc = pack_into(qfeat, nq, c->dim, feat_mem, c->qfeat.as<float>(), q_rows, c->dpad, c->ws, c->stream);
    if (rc == VSC_OK) rc = c->d_qoff.reserve((size_t)(n_qvid + 1) * 8);
    if (rc == VSC_OK) {
        hipError_t e = hipMemcpyAsync(c->d_qoff.p, off.data(), (size_t)(n_qvid + 1) * 8, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            set_error("vsc_tn_set_queries: upload of the query offsets failed: %s", hipGetErrorString(e));
            rc = VSC_ERR_HIP;
        }
    }
    if (rc != VSC_OK) {
        // the packed rows may be half written: an empty query side is the only state that cannot index past them
        c->n_qvid = 0;
        c->q_off.assign(1, 0);
        return rc;
    }
    c->n_qvid = n_qvid;
    c->q_off.swap(off);
    return VSC_OK;
}

int vsc_tn_destroy(vsc_tn_ctx_t* c) {
    if (!c) return VSC_OK;
    (void)hipSetDevice(c->device);
    if (c->stream == c->own_stream) { if (c->own_stream) (void)hipStreamSynchronize(c->own_stream); }
    else (void)hipDeviceSynchronize();
    c->qfeat.release(); c->rfeat.release(); c->d_qoff.release(); c->d_roff.release();
    c->d_pq.release(); c->d_pr.release(); c->d_work.release(); c->d_nbox.release();
    c->d_boxes.release(); c->d_bmax.release(); c->slab.release(); c->sims.release();
    c->ws.release();
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return VSC_OK;
}

int vsc_tn_localize(vsc_tn_ctx_t* c, const int32_t* pair_q, const int32_t* pair_r, int64_t n_pairs,
                    int pairs_mem, const vsc_tn_params* params, float bias, int32_t* out_nbox,
                    int32_t* out_boxes, float* out_boxmax, int out_mem) {
    if (!c || n_pairs < 0 || !params || (n_pairs > 0 && (!pair_q || !pair_r || !out_nbox || !out_boxes || !out_boxmax))) {
        set_error("vsc_tn_localize: invalid argument");
        return VSC_ERR_INVALID;
    }
    if (n_pairs == 0) return VSC_OK;
    if (params->tn_top_k < 1 || params->tn_top_k > 16 || params->tn_max_step < 1 || params->tn_max_step > 64 ||
        params->max_path < 0 || params->max_path >= VSC_TN_MAX_BOXES) {
        // max_path + 1 extractions can accept max_path + 1 boxes: more than the output holds would silently change
        // the IoU-suppression history
        set_error("vsc_tn_localize: unsupported TN parameters (tn_top_k 1..16, tn_max_step 1..64, max_path 0..%d)",
                  VSC_TN_MAX_BOXES - 1);
        return VSC_ERR_INVALID;
    }
    VSC_HIP(hipSetDevice(c->device));
    // pair lists on both sides: host for bucketing, device for the kernel
    std::vector<int32_t> hq((size_t)n_pairs), hr((size_t)n_pairs);
    VSC_TRY(c->d_pq.reserve((size_t)n_pairs * 4));
    VSC_TRY(c->d_pr.reserve((size_t)n_pairs * 4));
    if (pairs_mem == VSC_MEM_HOST) {
        memcpy(hq.data(), pair_q, (size_t)n_pairs * 4);
        memcpy(hr.data(), pair_r, (size_t)n_pairs * 4);
        VSC_HIP(hipMemcpyAsync(c->d_pq.p, pair_q, (size_t)n_pairs * 4, hipMemcpyHostToDevice, c->stream));
        VSC_HIP(hipMemcpyAsync(c->d_pr.p, pair_r, (size_t)n_pairs * 4, hipMemcpyHostToDevice, c->stream));
    } else {
        VSC_HIP(hipMemcpyAsync(hq.data(), pair_q, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, c->stream));
        VSC_HIP(hipMemcpyAsync(hr.data(), pair_r, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, c->stream));
        VSC_HIP(hipMemcpyAsync(c->d_pq.p, pair_q, (size_t)n_pairs * 4, hipMemcpyDeviceToDevice, c->stream));
        VSC_HIP(hipMemcpyAsync(c->d_pr.p, pair_r, (size_t)n_pairs * 4, hipMemcpyDeviceToDevice, c->stream));
    }
    VSC_HIP(hipStreamSynchronize(c->stream));
    std::vector<int32_t> lqs((size_t)n_pairs), lrs((size_t)n_pairs);
    for (int64_t p = 0; p < n_pairs; ++p) {
        const int32_t qv = hq[(size_t)p], rv = hr[(size_t)p];
        if (qv < 0 || qv >= c->n_qvid || rv < 0 || rv >= c->n_rvid) {
            set_error("vsc_tn_localize: pair %lld has video ordinal out of range", (long long)p);
            return VSC_ERR_INVALID;
        }
        lqs[(size_t)p] = (int32_t)std::min<int64_t>(c->q_off[qv + 1] - c->q_off[qv], 0x7fffffff);
        lrs[(size_t)p] = (int32_t)std::min<int64_t>(c->r_off[rv + 1] - c->r_off[rv], 0x7fffffff);
    }
    int32_t* d_nbox = out_nbox;
    int32_t* d_boxes = out_boxes;
    float* d_bmax = out_boxmax;
    if (out_mem == VSC_MEM_HOST) {
        VSC_TRY(c->d_nbox.reserve((size_t)n_pairs * 4));
        VSC_TRY(c->d_boxes.reserve((size_t)n_pairs * VSC_TN_MAX_BOXES * 16));
        VSC_TRY(c->d_bmax.reserve((size_t)n_pairs * VSC_TN_MAX_BOXES * 4));
        d_nbox = c->d_nbox.as<int32_t>();
        d_boxes = c->d_boxes.as<int32_t>();
        d_bmax = c->d_bmax.as<float>();
    }
    TnPairArgs base;
    memset(&base, 0, sizeof(base));
    base.qfeat = c->qfeat.as<float>();
    base.rfeat = c->rfeat.as<float>();
    base.q_off = c->d_qoff.as<int64_t>();
    base.r_off = c->d_roff.as<int64_t>();
    base.dpad = c->dpad;
    base.pair_q = c->d_pq.as<int32_t>();
    base.pair_r = c->d_pr.as<int32_t>();
    base.prm = *params;
    base.bias = bias;
    base.out_nbox = d_nbox;
    base.out_boxes = d_boxes;
    base.out_boxmax = d_bmax;
    VSC_TRY(tn_run_buckets(base, lqs, lrs, c->d_work, c->slab, c->stream));
    if (out_mem == VSC_MEM_HOST) {
        VSC_HIP(hipMemcpyAsync(out_nbox, d_nbox, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, c->stream));
        VSC_HIP(hipMemcpyAsync(out_boxes, d_boxes, (size_t)n_pairs * VSC_TN_MAX_BOXES * 16, hipMemcpyDeviceToHost, c->stream));
        VSC_HIP(hipMemcpyAsync(out_boxmax, d_bmax, (size_t)n_pairs * VSC_TN_MAX_BOXES * 4, hipMemcpyDeviceToHost, c->stream));
    }
    VSC_HIP(hipStreamSynchronize(c->stream));
    return VSC_OK;
}

int vsc_tn_forward_sim(const float* sims, const int64_t* sims_off, const int32_t* lq, const int32_t* lr,
                       int64_t n_pairs, const vsc_tn_params* params, int32_t* out_nbox, int32_t* out_boxes,
                       float* out_boxmax, int device) {
    if (n_pairs < 0 || !params || (n_pairs > 0 && (!sims || !sims_off || !lq || !lr || !out_nbox || !out_boxes || !out_boxmax))) {
        set_error("vsc_tn_forward_sim: invalid argument");
        return VSC_ERR_INVALID;
    }
    if (n_pairs == 0) return VSC_OK;
    if (params->tn_top_k < 1 || params->tn_top_k > 16 || params->tn_max_step < 1 || params->tn_max_step > 64 ||
        params->max_path < 0 || params->max_path >= VSC_TN_MAX_BOXES) {
        // max_path + 1 extractions can accept max_path + 1 boxes: more than the output holds would silently change
        // the IoU-suppression history
        set_error("vsc_tn_forward_sim: unsupported TN parameters (tn_top_k 1..16, tn_max_step 1..64, max_path 0..%d)",
                  VSC_TN_MAX_BOXES - 1);
        return VSC_ERR_INVALID;
    }
    VSC_TRY(check_device(device));
    VSC_HIP(hipSetDevice(device));
    DeviceCtx* c = device_ctx(device);
    if (!c) {
        set_error("vsc_tn_forward_sim: cannot create device context");
        return VSC_ERR_HIP;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    Workspace& ws = c->ws;
    const int64_t total = sims_off[n_pairs];
    VSC_TRY(ws.mat.reserve((size_t)std::max<int64_t>(total, 1) * 4));
    VSC_TRY(ws.w0.reserve((size_t)(n_pairs + 1) * 8));
    VSC_TRY(ws.w2.reserve((size_t)n_pairs * 4));
    VSC_TRY(ws.w3.reserve((size_t)n_pairs * 4));
    VSC_TRY(ws.out[0].reserve((size_t)n_pairs * 4));
    VSC_TRY(ws.out[1].reserve((size_t)n_pairs * VSC_TN_MAX_BOXES * 16));
    VSC_TRY(ws.out[2].reserve((size_t)n_pairs * VSC_TN_MAX_BOXES * 4));
    if (total) VSC_HIP(hipMemcpyAsync(ws.mat.p, sims, (size_t)total * 4, hipMemcpyHostToDevice, c->stream));
    VSC_HIP(hipMemcpyAsync(ws.w0.p, sims_off, (size_t)(n_pairs + 1) * 8, hipMemcpyHostToDevice, c->stream));
    VSC_HIP(hipMemcpyAsync(ws.w2.p, lq, (size_t)n_pairs * 4, hipMemcpyHostToDevice, c->stream));
    VSC_HIP(hipMemcpyAsync(ws.w3.p, lr, (size_t)n_pairs * 4, hipMemcpyHostToDevice, c->stream));
    TnPairArgs base;
    memset(&base, 0, sizeof(base));
    base.prm = *params;
    base.bias = 0.0f;
    base.out_nbox = ws.out[0].as<int32_t>();
    base.out_boxes = ws.out[1].as<int32_t>();
    base.out_boxmax = ws.out[2].as<float>();
    base.sims_in = ws.mat.as<float>();
    base.sims_off = ws.w0.as<int64_t>();
    base.sims_lq = ws.w2.as<int32_t>();
    base.sims_lr = ws.w3.as<int32_t>();
    std::vector<int32_t> lqs(lq, lq + n_pairs), lrs(lr, lr + n_pairs);
    VSC_TRY(tn_run_buckets(base, lqs, lrs, ws.maps0, ws.maps1, c->stream));
    VSC_HIP(hipMemcpyAsync(out_nbox, ws.out[0].p, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, c->stream));
    VSC_HIP(hipMemcpyAsync(out_boxes, ws.out[1].p, (size_t)n_pairs * VSC_TN_MAX_BOXES * 16, hipMemcpyDeviceToHost, c->stream));
    VSC_HIP(hipMemcpyAsync(out_boxmax, ws.out[2].p, (size_t)n_pairs * VSC_TN_MAX_BOXES * 4, hipMemcpyDeviceToHost, c->stream));
    VSC_HIP(hipStreamSynchronize(c->stream));
    return VSC_OK;
}

int vsc_tn_similarity(vsc_tn_ctx_t* c, int32_t q_vid, int32_t r_vid, float bias, float* out, int64_t cap,
                      int32_t* lq_out, int32_t* lr_out) {
    if (!c || q_vid < 0 || q_vid >= c->n_qvid || r_vid < 0 || r_vid >= c->n_rvid) {
        set_error("vsc_tn_similarity: invalid argument");
        return VSC_ERR_INVALID;
    }
    const int64_t lq = c->q_off[q_vid + 1] - c->q_off[q_vid], lr = c->r_off[r_vid + 1] - c->r_off[r_vid];
    if (lq_out) *lq_out = (int32_t)lq;
    if (lr_out) *lr_out = (int32_t)lr;
    if (lq * lr > cap || !out) {
        set_error("vsc_tn_similarity: output capacity %lld < %lld", (long long)cap, (long long)(lq * lr));
        return VSC_ERR_CAPACITY;
    }
    if (lq * lr == 0) return VSC_OK;
    VSC_HIP(hipSetDevice(c->device));
    VSC_TRY(c->sims.reserve((size_t)lq * lr * 4));
    TnSimsArgs a{c->qfeat.as<float>(), c->rfeat.as<float>(), c->q_off[q_vid], c->r_off[r_vid], (int)lq, (int)lr,
                 c->dpad, bias, c->sims.as<float>()};
    VSC_TRY(launch_tn_sims(a, c->stream));
    VSC_HIP(hipMemcpyAsync(out, c->sims.p, (size_t)lq * lr * 4, hipMemcpyDeviceToHost, c->stream));
    VSC_HIP(hipStreamSynchronize(c->stream));
    return VSC_OK;
}

}  // extern "C"

