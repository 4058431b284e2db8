// C ABI of libvscmi.so (declared in include/vscmi.h), part 1: errors, handles, options, streams, the images of the
// reference rows (add / int8 upkeep), query packing, hit buffers, profile counters.  Host orchestration only: every
// arithmetic step runs in the HIP kernels of the sibling translation units.  The searches live in api_search.hip and
// api_knn.hip, the handle-less entry points and the Temporal Network in api_aux.hip.
#include "api_internal.h"

namespace vscmi {

// ---- errors
static thread_local std::string g_err;
void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

// ---- debug aid (see vscmi_common.h)
bool poison_mode() {
    static const bool on = [] {
        const char* e = getenv("VSC_POISON_ALLOC");
        return e && e[0] == '1';
    }();
    return on;
}

int check_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("no HIP device visible (libvscmi needs an MI355X / gfx950 GPU)");
        return VSC_ERR_NODEVICE;
    }
    if (device < 0 || device >= n) {
        set_error("device %d out of range (have %d)", device, n);
        return VSC_ERR_INVALID;
    }
    hipDeviceProp_t p;
    VSC_HIP(hipGetDeviceProperties(&p, device));
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; libvscmi is built for gfx950 only", device, p.gcnArchName);
        return VSC_ERR_NODEVICE;
    }
    return VSC_OK;
}


int pack_half_any(const float* x, int64_t n, int dim, const HalfImage& h, int64_t r0, int64_t rows_out,
                         hipStream_t stream) {
    if (h.frag)
        return launch_pack_half_frag(x, n, dim, h.rows, h.norms + r0, h.row0 + r0, rows_out, h.dpadh, stream);
    return launch_pack_half(x, n, dim, h.rows + r0 * h.dpadh, h.norms + r0, rows_out, h.dpadh, stream);
}

int pack_into(const float* x, int64_t n, int dim, int mem, float* dst, int64_t rows_out, int dpad,
                     Workspace& ws, hipStream_t stream, const HalfImage& h) {
    if (mem == VSC_MEM_DEVICE || n == 0) {
        VSC_TRY(launch_pack_rows(x, n, dim, dst, rows_out, dpad, stream));
        if (h.rows) VSC_TRY(pack_half_any(x, n, dim, h, 0, h.rows_out, stream));
        return VSC_OK;
    }
    const int64_t chunk_rows = std::max<int64_t>(1, (int64_t)(256ll << 20) / ((int64_t)dim * 4));
    VSC_TRY(ws.stage.reserve((size_t)std::min(chunk_rows, n) * dim * 4));
    for (int64_t r0 = 0; r0 < n; r0 += chunk_rows) {
        const int64_t rows = std::min(chunk_rows, n - r0);
        VSC_HIP(hipMemcpyAsync(ws.stage.p, x + r0 * dim, (size_t)rows * dim * 4, hipMemcpyHostToDevice, stream));
        const bool last = (r0 + rows == n);
        const int64_t out_rows = last ? rows_out - r0 : rows;
        VSC_TRY(launch_pack_rows(ws.stage.as<float>(), rows, dim, dst + r0 * dpad, out_rows, dpad, stream));
        if (h.rows) VSC_TRY(pack_half_any(ws.stage.as<float>(), rows, dim, h, r0, last ? h.rows_out - r0 : rows, stream));
        VSC_HIP(hipStreamSynchronize(stream));  // staging buffer is reused
    }
    return VSC_OK;
}

}  // namespace vscmi

int prof_begin(vsc_index* idx, hipEvent_t* stop_out, int cls) {
    *stop_out = nullptr;
    if (!idx->prof) return VSC_OK;
    if (idx->ev_used == idx->ev_pool.size()) {
        hipEvent_t a, b;
        VSC_HIP(hipEventCreate(&a));
        VSC_HIP(hipEventCreate(&b));
        idx->ev_pool.emplace_back(a, b);
        idx->ev_class.push_back(0);
    }
    idx->ev_class[idx->ev_used] = cls;
    auto& e = idx->ev_pool[idx->ev_used++];
    VSC_HIP(hipEventRecord(e.first, idx->stream));
    *stop_out = e.second;
    return VSC_OK;
}
// `work`: algorithmic flops (classes 0, 1) or bytes (class 2) of the launch
int prof_end(vsc_index* idx, hipEvent_t stop, double work, int cls) {
    if (!stop) return VSC_OK;
    VSC_HIP(hipEventRecord(stop, idx->stream));
    idx->pending_work[cls] += work;
    return VSC_OK;
}
// call after a stream sync
int prof_collect(vsc_index* idx) {
    for (size_t e = 0; e < idx->ev_used; ++e) {
        float ms = 0.0f;
        VSC_HIP(hipEventElapsedTime(&ms, idx->ev_pool[e].first, idx->ev_pool[e].second));
        idx->prof_ms[idx->ev_class[e]] += ms;
        idx->prof_launches[idx->ev_class[e]] += 1;
    }
    for (int c = 0; c < 7; ++c) {
        idx->prof_work[c] += idx->pending_work[c];
        idx->pending_work[c] = 0.0;
    }
    idx->ev_used = 0;
    return VSC_OK;
}


AuxProf g_aux;

// ------------------------------------------------------------------ options
// One table for the environment switches (read when a handle is created) and vsc_index_set_option / _get_option.
struct OptionName { const char* name; const char* env; };
static const OptionName kOptions[] = {
    {"prefilter", "VSC_PREFILTER"}, {"prefilter_density", "VSC_PREFILTER_DENSITY"}, {"f16_kernel", "VSC_F16_KERNEL"},
    {"i8", "VSC_I8"}, {"i8_density", "VSC_I8_DENSITY"}, {"i8_max_rel", "VSC_I8_MAX_REL"}, {"i8_exclude", "VSC_I8_EXCLUDE"},
    {"i8_sort", "VSC_I8_SORT"}, {"i8_center", "VSC_I8_CENTER"}, {"i8_group", "VSC_I8_GROUP"}, {"i8p_order", "VSC_I8P_ORDER"}, {"i8p_slice", "VSC_I8P_SLICE"},
    {"i8p_pair", "VSC_I8P_PAIR"}, {"i8_screen", "VSC_I8_SCREEN"}, {"i8_knn", "VSC_I8_KNN"}, {"knn_step", "VSC_KNN_STEP"},
    {"knn_step_max", "VSC_KNN_STEP_MAX"}, {"knn_step_work", "VSC_KNN_STEP_WORK"}, {"rescore_sort", "VSC_RESCORE_SORT"},
    {"knn_levels", "VSC_KNN_LEVELS"}, {"knn_subset", "VSC_KNN_SUBSET"}, {"knn_s0div", "VSC_KNN_S0DIV"},
    {"knn_s0min", "VSC_KNN_S0MIN"}, {"knn_ratio", "VSC_KNN_RATIO"}, {"knn_nchunk", "VSC_KNN_NCHUNK"}, {"knn_first_tile", "VSC_KNN_FIRST_TILE"},
    {"cand_budget", "VSC_CAND_BUDGET"}, {"debug_i8", "VSC_DEBUG_I8"}, {"debug_screen", "VSC_DEBUG_SCREEN"},
    {"topk_shortcut", "VSC_TOPK_SHORTCUT"}, {"topk_sample", "VSC_TOPK_SAMPLE"}, {"sort_hits", "VSC_SORT_HITS"}, {"density_hint", "VSC_DENSITY_HINT"},
};

// Options that decide which images of the reference rows are kept can only change while the index is empty.
static int option_needs_empty(const vsc_index* idx, const char* name) {
    if (idx->ntotal == 0) return VSC_OK;
    set_error("vsc_index_set_option: '%s' decides which images of the reference rows exist and can only be set while "
              "the index is empty", name);
    return VSC_ERR_INVALID;
}

static int apply_option(vsc_index* idx, const char* name, double v) {
    auto is = [&](const char* n) { return strcmp(name, n) == 0; };
    const bool ip = idx->metric == VSC_METRIC_INNER_PRODUCT;
    if (is("prefilter")) {  // 0 off, 1 by density (default), 2 every batch / every k-NN (tests)
        const int m = (int)v;
        if (m < 0 || m > 2) goto bad;
        if ((m != 0) != idx->prefilter) VSC_TRY(option_needs_empty(idx, name));
        idx->prefilter = ip && m != 0;
        idx->prefilter_force = idx->prefilter && m == 2;
        if (!idx->prefilter) idx->i8_mode = 0;
        return VSC_OK;
    }
    if (is("i8")) {  // 0 no int8 image, 1 by density (default), 2 every pre-filtered batch (tests)
        const int m = (int)v;
        if (m < 0 || m > 2) goto bad;
        const int want = (idx->prefilter && idx->dpad8 <= I8P_MAX_DPAD8) ? m : 0;
        if ((want != 0) != (idx->i8_mode != 0)) VSC_TRY(option_needs_empty(idx, name));
        idx->i8_mode = want;
        return VSC_OK;
    }
    if (is("f16_kernel")) {  // 1 = the 256x256 LDS-ring kernel instead of the panel-stationary one (A/B)
        const bool frag = !(v != 0.0) && idx->dpadh <= F16P_MAX_DPADH;
        if (frag != idx->frag) VSC_TRY(option_needs_empty(idx, name));
        idx->frag = frag;
        return VSC_OK;
    }
    if (is("i8_exclude")) {
        const bool e = v != 0.0;
        if (e != idx->i8_exclude) VSC_TRY(option_needs_empty(idx, name));
        idx->i8_exclude = e;
        return VSC_OK;
    }
    if (is("i8_center")) {  // 0 never, 1 by the mean's share of the rows' energy (default), 2 always (tests)
        const int m = (int)v;
        if (m < 0 || m > 2) goto bad;
        if (m != idx->i8_center) VSC_TRY(option_needs_empty(idx, name));
        idx->i8_center = m;
        return VSC_OK;
    }
    if (is("prefilter_density")) { if (!(v > 0.0)) goto bad; idx->prefilter_density = v; return VSC_OK; }
    if (is("i8_density")) { if (!(v > 0.0)) goto bad; idx->i8_density = v; return VSC_OK; }
    if (is("i8_max_rel")) { if (!(v > 0.0)) goto bad; idx->i8_max_rel = v; return VSC_OK; }
    if (is("i8_sort")) { idx->i8_sort_rows = v != 0.0; return VSC_OK; }
    if (is("i8_group")) { idx->i8_group_shift = std::max(0, std::min(16, (int)v)); return VSC_OK; }
    if (is("i8p_order")) { idx->i8p_order = (int)v == 1 ? 1 : 0; return VSC_OK; }
    if (is("i8p_slice")) { if (v < 0.0) goto bad; idx->i8p_slice = (int)v; return VSC_OK; }
    if (is("i8p_pair")) { idx->i8p_pair = std::max(0, std::min(2, (int)v)); return VSC_OK; }
    if (is("i8_screen")) { idx->i8_screen = v == 1.0; return VSC_OK; }
    if (is("i8_knn")) { idx->knn_i8 = v != 0.0; return VSC_OK; }
    if (is("knn_step")) { idx->knn_step = std::max<int64_t>(0, (int64_t)v) / 256 * 256; return VSC_OK; }
    if (is("knn_step_max")) { idx->knn_step_max = std::max<int64_t>(32768, (int64_t)v); return VSC_OK; }
    if (is("knn_step_work")) { idx->knn_step_work = std::max(0.0, v); return VSC_OK; }
    if (is("rescore_sort")) { idx->rescore_by_ref = v != 0.0; return VSC_OK; }
    if (is("knn_levels")) { idx->knn_two_level = v != 1.0; return VSC_OK; }  // 1 = one refinement level only
    if (is("knn_subset")) { if (!(v > 0.0)) goto bad; idx->knn_subset_factor = v; return VSC_OK; }
    if (is("knn_s0div")) { idx->knn_s0_div = v > 0.0 ? (int)v : 28; return VSC_OK; }
    if (is("knn_s0min")) { idx->knn_s0_min = v >= 64.0 ? (int)v : 1024; return VSC_OK; }
    if (is("knn_ratio")) { idx->knn_ratio = v; return VSC_OK; }
    if (is("knn_nchunk")) { idx->knn_nchunk = (int)v; return VSC_OK; }
    if (is("knn_first_tile")) { idx->knn_first_tile = v != 0.0; return VSC_OK; }
    if (is("cand_budget")) { if (!(v >= 1048576.0)) goto bad; idx->cand_budget = (int64_t)v; return VSC_OK; }
    if (is("debug_i8")) { idx->debug_i8 = v != 0.0; return VSC_OK; }
    if (is("debug_screen")) { idx->debug_screen = v != 0.0; return VSC_OK; }
    if (is("topk_shortcut")) { const int m = (int)v; if (m < 0 || m > 2) goto bad; idx->topk_shortcut = m; return VSC_OK; }
    if (is("topk_sample")) { if (!(v >= 2.0)) goto bad; idx->topk_sample_rows = (int64_t)v; return VSC_OK; }
    if (is("sort_hits")) { idx->sort_hits = v != 0.0; return VSC_OK; }
    if (is("density_hint")) { if (!(v >= 0.0)) goto bad; idx->density_hint = v; return VSC_OK; }
    set_error("vsc_index_set_option: unknown option '%s'", name);
    return VSC_ERR_INVALID;
bad:
    set_error("vsc_index_set_option: value %g is out of range for '%s'", v, name);
    return VSC_ERR_INVALID;
}

static int read_option(const vsc_index* idx, const char* name, double* out) {
    auto is = [&](const char* n) { return strcmp(name, n) == 0; };
    if (is("prefilter")) *out = idx->prefilter ? (idx->prefilter_force ? 2 : 1) : 0;
    else if (is("i8")) *out = idx->i8_mode;
    else if (is("f16_kernel")) *out = idx->frag ? 0 : 1;
    else if (is("i8_exclude")) *out = idx->i8_exclude;
    else if (is("prefilter_density")) *out = idx->prefilter_density;
    else if (is("i8_density")) *out = idx->i8_density;
    else if (is("i8_max_rel")) *out = idx->i8_max_rel;
    else if (is("i8_sort")) *out = idx->i8_sort_rows;
    else if (is("i8_group")) *out = idx->i8_group_shift;
    else if (is("i8p_order")) *out = idx->i8p_order;
    else if (is("i8p_slice")) *out = idx->i8p_slice;
    else if (is("i8p_pair")) *out = idx->i8p_pair;
    else if (is("i8_screen")) *out = idx->i8_screen;
    else if (is("i8_knn")) *out = idx->knn_i8;
    else if (is("knn_step")) *out = (double)idx->knn_step;
    else if (is("knn_step_max")) *out = (double)idx->knn_step_max;
    else if (is("knn_step_work")) *out = idx->knn_step_work;
    else if (is("rescore_sort")) *out = idx->rescore_by_ref;
    else if (is("knn_levels")) *out = idx->knn_two_level ? 0 : 1;
    else if (is("knn_subset")) *out = idx->knn_subset_factor;
    else if (is("knn_s0div")) *out = idx->knn_s0_div;
    else if (is("knn_s0min")) *out = idx->knn_s0_min;
    else if (is("knn_ratio")) *out = idx->knn_ratio;
    else if (is("knn_nchunk")) *out = idx->knn_nchunk;
    else if (is("knn_first_tile")) *out = idx->knn_first_tile;
    else if (is("cand_budget")) *out = (double)idx->cand_budget;
    else if (is("debug_i8")) *out = idx->debug_i8;
    else if (is("debug_screen")) *out = idx->debug_screen;
    else if (is("topk_shortcut")) *out = idx->topk_shortcut;
    else if (is("topk_sample")) *out = (double)idx->topk_sample_rows;
    else if (is("sort_hits")) *out = idx->sort_hits;
    else if (is("density_hint")) *out = idx->density_hint;
    else if (is("last_topk_route")) *out = idx->last_topk_route;  // (read-only: what the last vsc_index_global_topk did)
    else if (is("i8_center")) *out = idx->i8_center;
    else if (is("i8_center_on")) *out = idx->i8_mu_on;        // (read-only: is the int8 reference image centred?)
    else if (is("i8_center_share")) *out = idx->i8_mu_ratio;  // (read-only: |mean|^2 / mean |row|^2 when it was decided)
    else if (is("i8_fallbacks")) *out = (double)idx->stat_i8_fallbacks;  // (read-only: searches that left int8 for fp16)
    else {
        set_error("vsc_index_get_option: unknown option '%s'", name);
        return VSC_ERR_INVALID;
    }
    return VSC_OK;
}

extern "C" {

int vsc_aux_profile(int enable) {
    g_aux.on = enable != 0;
    return VSC_OK;
}
int vsc_aux_profile_read(int cls, double* ms, int64_t* calls, double* bytes, int reset) {
    if (cls < 0 || cls > 1) return VSC_ERR_INVALID;
    std::lock_guard<std::mutex> lk(g_aux.mu);
    if (ms) *ms = g_aux.ms[cls];
    if (calls) *calls = g_aux.n[cls];
    if (bytes) *bytes = g_aux.bytes[cls];
    if (reset) { g_aux.ms[cls] = 0.0; g_aux.bytes[cls] = 0.0; g_aux.n[cls] = 0; }
    return VSC_OK;
}

const char* vsc_last_error(void) { return g_err.c_str(); }
int vsc_version(void) { return 100; }

int vsc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int d = 0; d < n; ++d) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, d) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

int vsc_index_create(int dim, int metric, int device, vsc_index_t** out) {
    if (!out || dim <= 0 || (metric != VSC_METRIC_INNER_PRODUCT && metric != VSC_METRIC_L2)) {
        set_error("vsc_index_create: invalid argument (dim=%d metric=%d)", dim, metric);
        return VSC_ERR_INVALID;
    }
    VSC_TRY(check_device(device));
    VSC_HIP(hipSetDevice(device));
    vsc_index* idx = new vsc_index();
    idx->dim = dim;
    idx->dpad = round_up(dim, K_PAD);
    idx->dpadh = round_up(dim, 128);
    idx->frag = idx->dpadh <= F16P_MAX_DPADH;
    idx->metric = metric;
    idx->dpad8 = round_up(dim, 256);
    idx->prefilter = metric == VSC_METRIC_INNER_PRODUCT;
    idx->i8_mode = (idx->prefilter && idx->dpad8 <= I8P_MAX_DPAD8) ? 1 : 0;
    // every switch of include/vscmi.h: the environment supplies the handle's initial options, vsc_index_set_option
    // changes them afterwards (the same names without the VSC_ prefix, lower case)
    for (const OptionName& o : kOptions) {
        const char* v = getenv(o.env);
        if (!v || !v[0]) continue;
        double x = atof(v);
        if (strcmp(o.name, "f16_kernel") == 0) x = v[0] == 'r' ? 1.0 : 0.0;           // VSC_F16_KERNEL=ring
        if (strcmp(o.name, "debug_i8") == 0 || strcmp(o.name, "debug_screen") == 0) x = 1.0;  // (set = on)
        (void)apply_option(idx, o.name, x);  // (an out-of-range value in the environment keeps the default, as before)
    }
    idx->device = device;
    hipError_t e = hipStreamCreateWithFlags(&idx->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        delete idx;
        return VSC_ERR_HIP;
    }
    idx->stream = idx->own_stream;
    int rc = set_thresh_kernel_attrs();
    if (rc != VSC_OK) {
        (void)hipStreamDestroy(idx->own_stream);
        delete idx;
        return rc;
    }
    *out = idx;
    return VSC_OK;
}

int vsc_index_destroy(vsc_index_t* idx) {
    if (!idx) return VSC_OK;
    (void)hipSetDevice(idx->device);
    // a caller's stream (vsc_index_set_stream) may be gone by now -- a torch side stream freed before the handle is
    // collected --: its handle is never touched again; the device is drained instead
    if (idx->stream == idx->own_stream) (void)hipStreamSynchronize(idx->own_stream);
    else (void)hipDeviceSynchronize();
    idx->ref.release();
    idx->refh.release();
    idx->refn.release();
    idx->ref8.release();
    idx->ref8m.release();
    idx->i8_mu.release();
    for (auto& b : idx->cand) b.release();
    idx->ws.release();
    for (auto& e : idx->ev_pool) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    (void)hipStreamDestroy(idx->own_stream);
    delete idx;
    return VSC_OK;
}

int vsc_index_set_stream(vsc_index_t* idx, void* hip_stream, int own) {
    if (!idx) {
        set_error("vsc_index_set_stream: invalid argument");
        return VSC_ERR_INVALID;
    }
    VSC_HIP(hipSetDevice(idx->device));
    // nothing of this handle is left on the stream it leaves (a caller's stream is not touched: it may already be
    // destroyed -- every entry point returns with its work complete, the device-wide drain is a formality)
    if (idx->stream == idx->own_stream) VSC_HIP(hipStreamSynchronize(idx->own_stream));
    else (void)hipDeviceSynchronize();
    VSC_TRY(prof_collect(idx));
    idx->stream = own ? idx->own_stream : (hipStream_t)hip_stream;  // (NULL = HIP's default stream, torch's default)
    return VSC_OK;
}

int vsc_index_set_option(vsc_index_t* idx, const char* name, double value) {
    if (!idx || !name) {
        set_error("vsc_index_set_option: invalid argument");
        return VSC_ERR_INVALID;
    }
    return apply_option(idx, name, value);
}

int vsc_index_get_option(const vsc_index_t* idx, const char* name, double* value) {
    if (!idx || !name || !value) {
        set_error("vsc_index_get_option: invalid argument");
        return VSC_ERR_INVALID;
    }
    return read_option(idx, name, value);
}

int64_t vsc_index_ntotal(const vsc_index_t* idx) { return idx ? idx->ntotal : 0; }
int vsc_index_dim(const vsc_index_t* idx) { return idx ? idx->dim : 0; }
int vsc_index_metric(const vsc_index_t* idx) { return idx ? idx->metric : 0; }

int vsc_index_set_hit_capacity(vsc_index_t* idx, int64_t cap) {
    if (!idx || cap < 0) {
        set_error("vsc_index_set_hit_capacity: invalid argument");
        return VSC_ERR_INVALID;
    }
    idx->hit_cap_user = cap;
    return VSC_OK;
}

int vsc_index_sync(vsc_index_t* idx) {
    if (!idx) return VSC_ERR_INVALID;
    VSC_HIP(hipSetDevice(idx->device));
    VSC_HIP(hipStreamSynchronize(idx->stream));
    return VSC_OK;
}

}  // extern "C"

// (Re)write rows [row0, row0 + rows) of the int8 image and their meta from the packed fp32 rows, with the index's
// current set of excluded coordinates; the first `count_rows - row0` of them enter the looseness statistic.
// The centre as the kernels read it: packed order, zero on the coordinates the image leaves out.
static int i8_upload_centre(vsc_index* idx) {
    if (!idx->i8_mu_on) return VSC_OK;
    std::vector<float> eff(idx->i8_mu_host);
    for (int c = 0; c < idx->i8_ex.n; ++c) eff[(size_t)k_slot(idx->i8_ex.idx[c])] = 0.0f;
    VSC_TRY(idx->i8_mu.reserve((size_t)idx->dpad * sizeof(float)));
    VSC_HIP(hipMemcpyAsync(idx->i8_mu.p, eff.data(), (size_t)idx->dpad * sizeof(float), hipMemcpyHostToDevice, idx->stream));
    VSC_HIP(hipStreamSynchronize(idx->stream));  // eff is a local
    idx->i8_mu_ex = idx->i8_ex;
    return VSC_OK;
}

// Decide ONCE, from the rows present at the first catch-up, whether the image is centred and on what: the mean of those
// rows (any fixed vector would be exact; the mean is what shortens the rows most).  Centring is on when the mean carries
// at least 2 % of the rows' energy on the kept coordinates (option i8_center = 1), always (2) or never (0).
static int i8_decide_centre(vsc_index* idx, int64_t rows) {
    idx->i8_mu_decided = true;
    idx->i8_mu_on = false;
    if (idx->i8_center == 0 || rows <= 0) return VSC_OK;
    const int dpad = idx->dpad;
    VSC_TRY(idx->ws.tmp.reserve((size_t)2 * dpad * sizeof(double)));
    double* d_sum = idx->ws.tmp.as<double>();
    VSC_TRY(launch_col_sums(idx->ref.as<float>(), rows, dpad, d_sum, d_sum + dpad, idx->stream));
    std::vector<double> h((size_t)2 * dpad);
    VSC_HIP(hipMemcpyAsync(h.data(), d_sum, h.size() * sizeof(double), hipMemcpyDeviceToHost, idx->stream));
    VSC_HIP(hipStreamSynchronize(idx->stream));
    idx->i8_mu_host.assign((size_t)dpad, 0.0f);
    double mu2 = 0.0, e2 = 0.0;
    bool finite = true;
    for (int k = 0; k < idx->dim; ++k) {
        if (idx->i8_ex.holds(k)) continue;
        const int p = k_slot(k);
        const double m = h[(size_t)p] / (double)rows;
        finite = finite && std::isfinite(m) && std::isfinite(h[(size_t)dpad + p]);
        idx->i8_mu_host[(size_t)p] = (float)m;
        mu2 += m * m;
        e2 += h[(size_t)dpad + p] / (double)rows;
    }
    idx->i8_mu_ratio = e2 > 0.0 ? mu2 / e2 : 0.0;
    if (!finite) return VSC_OK;  // (rows holding inf / NaN: no centre; such rows pass every pair anyway)
    idx->i8_mu_on = idx->i8_center == 2 || idx->i8_mu_ratio >= 0.02;
    if (idx->debug_i8)
        fprintf(stderr, "[vscmi] int8 reference image: mean carries %.4f of the rows' energy over %lld rows -> %s\n", idx->i8_mu_ratio,
                (long long)rows, idx->i8_mu_on ? "centred" : "not centred");
    return i8_upload_centre(idx);
}

static int i8_quantise(vsc_index* idx, int64_t row0, int64_t rows, int64_t count_rows) {
    if (rows <= 0) return VSC_OK;
    if (idx->i8_mu_on) {
        // the excluded set changed since the centre was uploaded: zero it on the coordinates that are left out now
        bool same = idx->i8_mu_ex.n == idx->i8_ex.n;
        for (int c = 0; same && c < idx->i8_ex.n; ++c) same = idx->i8_mu_ex.idx[c] == idx->i8_ex.idx[c];
        if (!same) VSC_TRY(i8_upload_centre(idx));
    }
    VSC_TRY(launch_quant_ref_frag(idx->ref.as<float>(), idx->dpad, idx->ref8.p, idx->ref8m.as<float4>(), row0, rows,
                                  idx->dpad8, idx->i8_ex, idx->i8_mu_on ? idx->i8_mu.as<float>() : nullptr, idx->stream));
    const int64_t real = std::max<int64_t>(0, std::min(row0 + rows, count_rows) - row0);
    VSC_TRY(idx->ws.cnt.reserve(2 * sizeof(double)));
    VSC_TRY(launch_meta_looseness(idx->ref8m.as<float4>() + row0, real, idx->ws.cnt.as<double>(), idx->stream));
    double h2[2] = {0.0, 0.0};
    VSC_HIP(hipMemcpyAsync(h2, idx->ws.cnt.p, sizeof(h2), hipMemcpyDeviceToHost, idx->stream));
    VSC_HIP(hipStreamSynchronize(idx->stream));
    idx->i8_loose_sum += h2[0];
    idx->i8_loose_cnt += h2[1];
    idx->i8_rows = std::max(idx->i8_rows, row0 + rows);
    return VSC_OK;
}

// After `n` rows were appended at `first_new` (already packed): fold their per-coordinate min / max into the index's,
// re-derive the set of coordinates on which ALL rows agree (up to 8, largest magnitude first, zero values are
// pointless) and either quantise just the new rows (set unchanged) or mark the whole image stale.
static int i8_after_add(vsc_index* idx, int64_t first_new, int64_t n, int64_t need_rows) {
    const int dpad = idx->dpad;
    VSC_TRY(idx->ws.tmp.reserve((size_t)2 * dpad * sizeof(unsigned)));
    unsigned* d_mn = idx->ws.tmp.as<unsigned>();
    unsigned* d_mx = d_mn + dpad;
    VSC_TRY(launch_dim_minmax(idx->ref.as<float>() + first_new * dpad, n, dpad, d_mn, d_mx, idx->stream));
    std::vector<unsigned> mn((size_t)dpad), mx((size_t)dpad);
    VSC_HIP(hipMemcpyAsync(mn.data(), d_mn, (size_t)dpad * sizeof(unsigned), hipMemcpyDeviceToHost, idx->stream));
    VSC_HIP(hipMemcpyAsync(mx.data(), d_mx, (size_t)dpad * sizeof(unsigned), hipMemcpyDeviceToHost, idx->stream));
    VSC_HIP(hipStreamSynchronize(idx->stream));
    if (idx->cmin_key.empty()) {
        idx->cmin_key.assign((size_t)idx->dim, 0xffffffffu);
        idx->cmax_key.assign((size_t)idx->dim, 0u);
    }
    for (int k = 0; k < idx->dim; ++k) {
        const int p = k_slot(k);
        idx->cmin_key[(size_t)k] = std::min(idx->cmin_key[(size_t)k], mn[(size_t)p]);
        idx->cmax_key[(size_t)k] = std::max(idx->cmax_key[(size_t)k], mx[(size_t)p]);
    }
    // constant coordinates, largest magnitude first
    std::vector<std::pair<float, int>> cst;
    for (int k = 0; k < idx->dim; ++k)
        if (idx->cmin_key[(size_t)k] == idx->cmax_key[(size_t)k]) {
            const float v = key2f(idx->cmin_key[(size_t)k]);
            if (std::isfinite(v) && v != 0.0f) cst.emplace_back(-std::fabs(v), k);
        }
    std::sort(cst.begin(), cst.end());
    ExcludedDims ex;
    const bool no_ex = !idx->i8_exclude;
    for (size_t c = 0; c < cst.size() && ex.n < I8_MAX_EXCLUDED && !no_ex; ++c) {
        ex.idx[ex.n] = cst[c].second;
        ex.val[ex.n] = key2f(idx->cmin_key[(size_t)cst[c].second]);
        ++ex.n;
    }
    bool same = ex.n == idx->i8_ex.n;
    for (int c = 0; same && c < ex.n; ++c) same = ex.idx[c] == idx->i8_ex.idx[c] && ex.val[c] == idx->i8_ex.val[c];
    idx->i8_ex = ex;
    if (!same && first_new > 0) idx->i8_dirty = true;  // the rows quantised so far left other coordinates out
    // the centre of the image: decided once, as soon as there are enough rows for their mean to mean something (a search
    // after the first few rows must not fix it on one video's descriptors); rows quantised before that are rewritten
    if (!idx->i8_mu_decided && (idx->i8_center != 1 || first_new + n >= 1024)) {
        VSC_TRY(i8_decide_centre(idx, first_new + n));
        if (idx->i8_mu_on && first_new > 0) idx->i8_dirty = true;
    }
    if (idx->i8_dirty) return VSC_OK;  // everything is rewritten before the next search anyway
    return i8_quantise(idx, first_new, need_rows - first_new, first_new + n);
}

extern "C" {

int vsc_index_add(vsc_index_t* idx, const float* x, int64_t n, int x_mem) {
    if (!idx || n < 0 || (n > 0 && !x)) {
        set_error("vsc_index_add: invalid argument");
        return VSC_ERR_INVALID;
    }
    if (n == 0) return VSC_OK;
    if (idx->ntotal + n >= 0x7fffff00LL) {
        set_error("vsc_index_add: more than 2^31 reference rows");
        return VSC_ERR_INVALID;
    }
    VSC_HIP(hipSetDevice(idx->device));
    const int64_t need_rows = round_up64(idx->ntotal + n, ROW_PAD_REF);
    if (need_rows > idx->cap_rows) {
        // grow geometrically; keep the old rows
        int64_t cap = std::max<int64_t>(need_rows, idx->cap_rows + idx->cap_rows / 2);
        cap = round_up64(cap, ROW_PAD_REF);
        DevBuf nb, nh, nn, n8, n8m;
        VSC_TRY(nb.reserve((size_t)cap * idx->dpad * 4));
        if (idx->prefilter) {
            // (+ one col-step of rows: a launch over the reference range [b, e) walks whole col-steps FROM b, and b is
            // only tile-aligned when the tests force the k-NN's levels on small indexes -- the per-row tables are read
            // with plain loads up to b + round_up(e - b, 512) <= cap + 511; the images go through bounds-checked
            // buffer descriptors)
            VSC_TRY(nh.reserve((size_t)(cap + F16P_COL_STEP) * idx->dpadh * 2));
            VSC_TRY(nn.reserve((size_t)(cap + F16P_COL_STEP) * 4));
        }
        if (idx->i8_mode) {
            VSC_TRY(n8.reserve((size_t)(cap + F16P_COL_STEP) * idx->dpad8));
            VSC_TRY(n8m.reserve((size_t)(cap + F16P_COL_STEP) * sizeof(float4)));
        }
        if (idx->ntotal > 0) {
            VSC_HIP(hipMemcpyAsync(nb.p, idx->ref.p, (size_t)idx->ntotal * idx->dpad * 4,
                                   hipMemcpyDeviceToDevice, idx->stream));
            if (idx->prefilter) {
                // (fragment-major: whole 64-row tiles; the padding rows of the last one are rewritten below)
                VSC_HIP(hipMemcpyAsync(nh.p, idx->refh.p, (size_t)round_up64(idx->ntotal, 64) * idx->dpadh * 2,
                                       hipMemcpyDeviceToDevice, idx->stream));
                VSC_HIP(hipMemcpyAsync(nn.p, idx->refn.p, (size_t)idx->ntotal * 4, hipMemcpyDeviceToDevice,
                                       idx->stream));
            }
            if (idx->i8_mode) {
                VSC_HIP(hipMemcpyAsync(n8.p, idx->ref8.p, (size_t)round_up64(idx->ntotal, 64) * idx->dpad8,
                                       hipMemcpyDeviceToDevice, idx->stream));
                VSC_HIP(hipMemcpyAsync(n8m.p, idx->ref8m.p, (size_t)idx->ntotal * sizeof(float4), hipMemcpyDeviceToDevice,
                                       idx->stream));
            }
            VSC_HIP(hipStreamSynchronize(idx->stream));
        }
        idx->ref.release();
        idx->refh.release();
        idx->refn.release();
        idx->ref8.release();
        idx->ref8m.release();
        idx->ref = nb;
        idx->refh = nh;
        idx->refn = nn;
        idx->ref8 = n8;
        idx->ref8m = n8m;
        idx->cap_rows = cap;
    }
    float* dst = idx->ref.as<float>() + idx->ntotal * idx->dpad;
    HalfImage h;
    if (idx->prefilter) {
        h.frag = idx->frag;
        h.row0 = idx->ntotal;
        h.rows = idx->frag ? idx->refh.as<_Float16>() : idx->refh.as<_Float16>() + idx->ntotal * idx->dpadh;
        h.norms = idx->refn.as<float>() + idx->ntotal;
        h.rows_out = need_rows - idx->ntotal;
        h.dpadh = idx->dpadh;
    }
    VSC_TRY(pack_into(x, n, idx->dim, x_mem, dst, need_rows - idx->ntotal, idx->dpad, idx->ws, idx->stream, h));
    VSC_HIP(hipStreamSynchronize(idx->stream));
    idx->ntotal += n;  // (the int8 image catches up in i8_prepare, before the next search)
    return VSC_OK;
}

}  // extern "C"

// Call before a search that may use the int8 kernel: brings the image up to date when the set of excluded
// coordinates changed since it was written.
int i8_prepare(vsc_index* idx) {
    if (!idx->i8_mode) return VSC_OK;
    if (idx->i8_seen < idx->ntotal) {
        const int64_t first_new = idx->i8_seen;
        const int rc = i8_after_add(idx, first_new, idx->ntotal - first_new, round_up64(idx->ntotal, ROW_PAD_REF));
        if (rc != VSC_OK) {
            // (allocation / HIP failure half way: the rows stay "unseen" -- the min / max fold is idempotent -- and the
            // whole image is rewritten before the next search may use it)
            idx->i8_dirty = true;
            return rc;
        }
        idx->i8_seen = idx->ntotal;
    }
    if (!idx->i8_dirty) return VSC_OK;
    idx->i8_loose_sum = idx->i8_loose_cnt = 0.0;
    VSC_TRY(i8_quantise(idx, 0, round_up64(idx->ntotal, ROW_PAD_REF), idx->ntotal));
    idx->i8_dirty = false;
    return VSC_OK;
}

// May this search use the int8 kernel at all?  (mode 2 = forced by the tests)
bool i8_usable(const vsc_index* idx) {
    if (idx->i8_mode == 2) return true;
    if (idx->i8_mode != 1) return false;
    if (idx->i8_loose_cnt <= 0.0) return true;
    return std::sqrt((double)idx->dim) * (idx->i8_loose_sum / idx->i8_loose_cnt) <= idx->i8_max_rel;
}

// Pack the query rows: returns device pointer; buffer holds round_up(nq,128)+128 zero-padded rows.
int pack_queries(vsc_index* idx, const float* q, int64_t nq, int q_mem, float** out, bool with_half) {
    const int64_t rows = round_up64(nq, ROW_PAD) + ROW_PAD;
    VSC_TRY(idx->ws.qbuf.reserve((size_t)rows * idx->dpad * 4));
    HalfImage h;
    if (with_half) {
        // a batch starts at any multiple of 32 rows and reads whole 256-row tiles from there
        h.rows_out = round_up64(nq, ROW_PAD_H) + ROW_PAD_H;
        h.dpadh = idx->dpadh;
        VSC_TRY(idx->ws.qh.reserve((size_t)h.rows_out * idx->dpadh * 2));
        VSC_TRY(idx->ws.qn.reserve((size_t)h.rows_out * 4));
        h.rows = idx->ws.qh.as<_Float16>();
        h.norms = idx->ws.qn.as<float>();
    }
    VSC_TRY(pack_into(q, nq, idx->dim, q_mem, idx->ws.qbuf.as<float>(), rows, idx->dpad, idx->ws, idx->stream, h));
    *out = idx->ws.qbuf.as<float>();
    return VSC_OK;
}

// cap: kept hits (list A, and the compaction target B of the thresholded search); ccap: candidates of ONE
// pre-filter launch (defaults to cap)
int ensure_hit_buffers(vsc_index* idx, int64_t cap, int64_t ccap, bool need_b) {
    if (ccap < 0) ccap = cap;
    for (int c = 0; c < 3; ++c) {
        VSC_TRY(idx->ws.hA[c].reserve((size_t)cap * 4 + 16));  // (+ 16: select_hist_kernel reads whole 16-byte pieces)
        if (need_b) VSC_TRY(idx->ws.hB[c].reserve((size_t)cap * 4 + 16));
    }
    if (idx->prefilter) {
        // candidate list: `ccap` entries in per-wave segments + a shared tail.  The tail is handed out in chunks
        // (cand_list.h): a wave closes a chunk when its next group of <= 64 candidates does not fit, so a chunk is at
        // least half used on average, and every wave leaves one chunk partly filled -- 3 ccap + 128 entries per wave
        // hold any distribution of <= ccap candidates over the waves
        VSC_TRY(idx->ws.ci.reserve((size_t)cand_entries(ccap) * 4));
        VSC_TRY(idx->ws.cj.reserve((size_t)cand_entries(ccap) * 4));
        VSC_TRY(idx->ws.segcnt.reserve(2048 * sizeof(int)));
    }
    VSC_TRY(idx->ws.ctl.reserve(sizeof(SelectCtl)));
    return VSC_OK;
}


extern "C" {

int vsc_index_profile(vsc_index_t* idx, int enable) {
    if (!idx) return VSC_ERR_INVALID;
    idx->prof = enable != 0;
    return VSC_OK;
}

int vsc_index_profile_read_class(vsc_index_t* idx, int cls, double* ms, int64_t* launches, double* work,
                                 int reset) {
    if (!idx || cls < 0 || cls > 6) return VSC_ERR_INVALID;
    if (ms) *ms = idx->prof_ms[cls];
    if (launches) *launches = idx->prof_launches[cls];
    if (work) *work = idx->prof_work[cls];
    if (reset) {
        idx->prof_ms[cls] = 0.0;
        idx->prof_work[cls] = 0.0;
        idx->prof_launches[cls] = 0;
    }
    return VSC_OK;
}

int vsc_index_profile_read(vsc_index_t* idx, double* sim_ms, int64_t* sim_launches, double* sim_flops,
                           int reset) {
    return vsc_index_profile_read_class(idx, 0, sim_ms, sim_launches, sim_flops, reset);
}

int vsc_index_search_stats(vsc_index_t* idx, int64_t* candidates, int64_t* hits) {
    if (!idx) return VSC_ERR_INVALID;
    if (candidates) *candidates = (int64_t)idx->stat_candidates;
    if (hits) *hits = (int64_t)idx->stat_hits;
    return VSC_OK;
}

}  // extern "C"
