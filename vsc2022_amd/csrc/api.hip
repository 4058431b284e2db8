// C ABI of libvscmi.so (declared in include/vscmi.h).  Host orchestration only: handles, HBM
// residency, the stream-ordered batch schedule of the global-threshold search, staging of host
// buffers.  Every arithmetic step runs in the HIP kernels of the sibling translation units.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "kernels.h"

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

static int check_device(int device) {
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

// Workspace shared by the operations of one stream owner.
struct Workspace {
    DevBuf stage;   // host->device staging of raw fp32 rows
    DevBuf qbuf;    // packed query rows
    DevBuf hA[3], hB[3];  // kept hits (i, j, s) + compaction target
    DevBuf ctl;     // SelectCtl
    DevBuf w0, w1, w2, w3, tmp, cnt;  // sort scratch
    DevBuf out[4];  // device-side outputs when the caller wants host results
    DevBuf parts, partj, mat, maps0, maps1;
    DevBuf qh, qn;  // fp16 image + norm bounds of the query rows (pre-filter)
    DevBuf ci, cj, segcnt;  // pre-filter candidates of one batch (per-wave segments + their fill levels)
    DevBuf rowthr;          // per-row thresholds of the pre-filtered k-NN
    DevBuf slices;          // per-panel slice counters of the panel-stationary pre-filter
    DevBuf q8, pstat;       // int8 image + per-panel {1/s, E, N, s} of ONE launch's query rows (sim_i8p.hip)
    DevBuf cs[4], cstmp, csn;  // candidates of a launch compacted + sorted by reference row (keys, values, ping-pong)
    DevBuf tailfill;        // fill levels of the chunks of the candidate list's shared tail (cand_list.h)
    DevBuf rt8c;            // the rows' largest |x| (second sort key of launches with per-row thresholds)
    DevBuf rt8, rt8b;       // ... and its row thresholds in position order (rows sorted by threshold inside a launch);
                            // rt8b: thresholds lowered by the excluded coordinates' contribution, in row order
    void release() {
        stage.release(); qbuf.release();
        qh.release(); qn.release(); ci.release(); cj.release(); segcnt.release(); rowthr.release(); slices.release();
        q8.release(); pstat.release(); rt8.release(); rt8b.release(); rt8c.release(); tailfill.release();
        for (auto& b : cs) b.release();
        cstmp.release(); csn.release();
        for (auto& b : hA) b.release();
        for (auto& b : hB) b.release();
        ctl.release(); w0.release(); w1.release(); w2.release(); w3.release(); tmp.release(); cnt.release();
        for (auto& b : out) b.release();
        parts.release(); partj.release(); mat.release(); maps0.release(); maps1.release();
    }
};

// Bring raw fp32 rows (host or device) into the packed engine layout at dst (rows_out rows are
// written, rows >= n zero).  Host sources are staged in chunks.
// Optional second image for the fp16 pre-filter: rows_out_h rows of dpadh halves + one norm per row.
struct HalfImage {
    _Float16* rows = nullptr;  // natural layout: first row to write; fragment-major: base of the WHOLE image
    float* norms = nullptr;    // first norm to write
    int64_t rows_out = 0;
    int dpadh = 0;
    bool frag = false;         // fragment-major reference image of the panel-stationary pre-filter (sim_f16p.hip)
    int64_t row0 = 0;          // fragment-major: absolute index of the first row written
};

static int pack_half_any(const float* x, int64_t n, int dim, const HalfImage& h, int64_t r0, int64_t rows_out,
                         hipStream_t stream) {
    if (h.frag)
        return launch_pack_half_frag(x, n, dim, h.rows, h.norms + r0, h.row0 + r0, rows_out, h.dpadh, stream);
    return launch_pack_half(x, n, dim, h.rows + r0 * h.dpadh, h.norms + r0, rows_out, h.dpadh, stream);
}

static int pack_into(const float* x, int64_t n, int dim, int mem, float* dst, int64_t rows_out, int dpad,
                     Workspace& ws, hipStream_t stream, const HalfImage& h = HalfImage()) {
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

using namespace vscmi;

struct vsc_index {
    int dim = 0, dpad = 0, metric = 0, device = 0;
    int64_t ntotal = 0, cap_rows = 0;
    DevBuf ref;
    // fp16 image (dpadh halves per row) and row-norm bounds of the references: the pre-filter of the
    // thresholded inner-product searches (sim_f16.hip).  Not kept for L2 indexes.
    DevBuf refh, refn;
    int dpadh = 0;
    bool frag = false;  // refh is fragment-major (dpadh <= 512: panel-stationary pre-filter), else natural
    // int8 image (dpad8 bytes per row, fragment-major) + per-row {1/s, E, N, s}: the pre-filter of the batches
    // whose hits are sparse (sim_i8p.hip).  i8_mode: 0 off, 1 chosen per batch by expected hit density, 2 every
    // pre-filtered batch (tests)
    DevBuf ref8, ref8m;
    int dpad8 = 0, i8_mode = 0;
    double i8_density = 5e-4;
    // sum / count of E_r / N_r over the reference rows: sqrt(dim) x their mean is the references' share of eps / sigma
    // (0.17 for unit-norm Gaussian-like rows); above i8_max_rel the 8-bit bound passes too much and the batches stay
    // on the fp16 kernel (e.g. score-normalised descriptors: one coordinate of every row is 1, the scale follows it)
    double i8_loose_sum = 0.0, i8_loose_cnt = 0.0, i8_max_rel = 0.35;
    // coordinates on which all reference rows agree (order-preserving keys of the per-coordinate min / max over every
    // row added so far), the ones the int8 image currently leaves out, and whether the image lags behind the rows
    // (it is (re)written from the packed fp32 rows: for the new rows at `add` while the excluded set stays the same,
    // for all rows before the next search when it changed)
    std::vector<unsigned> cmin_key, cmax_key;
    ExcludedDims i8_ex;
    bool i8_dirty = false;
    int64_t i8_rows = 0;  // rows [0, i8_rows) of the image are current
    // rows [i8_seen, ntotal) have been added but not yet folded into the per-coordinate min / max nor quantised: `add`
    // only packs rows, the first search afterwards catches up in one go (a dim_minmax pass, two copies to
    // the host and two stream syncs PER ADD made many small adds -- one per video -- slow)
    int64_t i8_seen = 0;
    unsigned long long stat_i8_fallbacks = 0;
    // tuning / A-B switches of the pre-filtered routes, read from the environment when the handle is created
    // (include/vscmi.h lists them)
    bool i8_exclude = true;      // VSC_I8_EXCLUDE=0: keep agreeing coordinates in the images
    int i8p_order = 1;           // VSC_I8P_ORDER: 1 slice-major work items (default), 0 panel-major with stealing
    int i8p_pair = 1;            // VSC_I8P_PAIR: 1 work items of two panels where the launch is large enough (default), 0 never, 2 wherever legal
    int64_t knn_step = 0;        // VSC_KNN_STEP: query rows per launch of a k-NN threshold pass (0: 32768, more over short ranges)
    double knn_step_work = 64.0;  // VSC_KNN_STEP_WORK: x 32768 x 196608 = rows x range a launch should reach
    int64_t knn_step_max = 262144;  // VSC_KNN_STEP_MAX: ... at most this many (131072 / 262144 / 524288: 2125 / 2120 / 2121 ms per configs[3] step)
    int i8p_slice = 0;           // VSC_I8P_SLICE: col-steps per work item (0: 16 slice-major / the plan's panel-major)
    bool i8_sort_rows = true;    // VSC_I8_SORT=0: the rows of a launch keep their order
    int i8_group_shift = 9;      // VSC_I8_GROUP=<log2 rows>: radius searches with per-row thresholds order groups of 2^n rows by scale (0: off)
    bool rescore_by_ref = true;  // VSC_RESCORE_SORT=0: re-score the waves' segments as they are
    bool i8_screen = false;      // VSC_I8_SCREEN=1: fp16 screen between the int8 pre-filter and the exact stage
    bool knn_i8 = true;          // VSC_I8_KNN=0: k-NN passes on the fp16 kernel
    bool knn_two_level = true;   // VSC_KNN_LEVELS=1: one refinement level
    double knn_subset_factor = 300.0;  // VSC_KNN_SUBSET
    int knn_s0_div = 28;         // VSC_KNN_S0DIV
    int knn_s0_min = 1024;       // VSC_KNN_S0MIN: smallest exact subset
    double knn_ratio = 0.0;      // VSC_KNN_RATIO (0: by k)
    int knn_nchunk = 0;          // VSC_KNN_NCHUNK: reference chunks of the exact k-NN kernel (0: by size)
    bool debug_i8 = false, debug_screen = false;  // VSC_DEBUG_I8 / VSC_DEBUG_SCREEN: stderr notes
    bool prefilter = false, prefilter_force = false;
    double prefilter_density = 0.05;  // expected hit density below which a batch goes through the pre-filter (r03: 0.02 -> 0.05 with the cheaper exact stage: -0.8 %)
    unsigned long long stat_candidates = 0, stat_hits = 0;  // last search (vsc_index_profile_read)
    DevBuf cand[3];  // sorted hits of vsc_index_candidates
    hipStream_t stream = nullptr;      // the stream every launch of this handle goes to: own_stream, or the caller's
    hipStream_t own_stream = nullptr;  // (vsc_index_set_stream)
    int64_t cand_budget = (int64_t)1 << 28;  // cand_budget: entries of the candidate list a k-NN threshold pass may ask for
    Workspace ws;
    int64_t hit_cap_user = 0;
    int64_t hit_cap_learned = 0;  // the capacity the last search ended with after overflow reruns (ties keep the radius low)
    // kernel-time accounting (HIP events on the handle's stream), per kernel class:
    // 0 = exact fp32 similarity kernels, 1 = fp16 pre-filter, 2 = exact re-scoring of candidates,
    // 3 = re-threshold (radix select + compaction) kernels, 4 = final ordering of the kept hits
    bool prof = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    std::vector<int> ev_class;
    size_t ev_used = 0;
    // 5 = int8 pre-filter kernel, 6 = its launches' preamble (row thresholds / scales, sorts, quantisation of the panels)
    double prof_ms[7] = {}, prof_work[7] = {}, pending_work[7] = {};
    int64_t prof_launches[7] = {};
};

static int prof_begin(vsc_index* idx, hipEvent_t* stop_out, int cls = 0) {
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
static int prof_end(vsc_index* idx, hipEvent_t stop, double work, int cls = 0) {
    if (!stop) return VSC_OK;
    VSC_HIP(hipEventRecord(stop, idx->stream));
    idx->pending_work[cls] += work;
    return VSC_OK;
}
// call after a stream sync
static int prof_collect(vsc_index* idx) {
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

// Process-wide kernel-time accounting of the entry points that own no index handle (HIP events on the stream
// the kernels run on; read after the call's own stream sync): 0 = vsc_pair_max, 1 = Temporal-Network launches.
struct AuxProf {
    bool on = false;
    std::mutex mu;
    double ms[2] = {}, bytes[2] = {};
    int64_t n[2] = {};
};
static AuxProf g_aux;
struct AuxTimer {
    hipEvent_t a = nullptr, b = nullptr;
    int cls = 0;
    double bytes = 0.0;
    void begin(int c, hipStream_t s) {
        cls = c;
        if (!g_aux.on) return;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
        (void)hipEventRecord(a, s);
    }
    void end(double by, hipStream_t s) {
        if (a && b) { (void)hipEventRecord(b, s); bytes += by; }
    }
    void collect() {  // after the stream has been synchronised
        if (!a || !b) return;
        float t = 0.0f;
        if (hipEventElapsedTime(&t, a, b) == hipSuccess) {
            std::lock_guard<std::mutex> lk(g_aux.mu);
            g_aux.ms[cls] += t;
            g_aux.bytes[cls] += bytes;
            g_aux.n[cls] += 1;
        }
        (void)hipEventDestroy(a);
        (void)hipEventDestroy(b);
        a = b = nullptr;
    }
};

// ------------------------------------------------------------------ options
// One table for the environment switches (read when a handle is created) and vsc_index_set_option / _get_option.
struct OptionName { const char* name; const char* env; };
static const OptionName kOptions[] = {
    {"prefilter", "VSC_PREFILTER"}, {"prefilter_density", "VSC_PREFILTER_DENSITY"}, {"f16_kernel", "VSC_F16_KERNEL"},
    {"i8", "VSC_I8"}, {"i8_density", "VSC_I8_DENSITY"}, {"i8_max_rel", "VSC_I8_MAX_REL"}, {"i8_exclude", "VSC_I8_EXCLUDE"},
    {"i8_sort", "VSC_I8_SORT"}, {"i8_group", "VSC_I8_GROUP"}, {"i8p_order", "VSC_I8P_ORDER"}, {"i8p_slice", "VSC_I8P_SLICE"},
    {"i8p_pair", "VSC_I8P_PAIR"}, {"i8_screen", "VSC_I8_SCREEN"}, {"i8_knn", "VSC_I8_KNN"}, {"knn_step", "VSC_KNN_STEP"},
    {"knn_step_max", "VSC_KNN_STEP_MAX"}, {"knn_step_work", "VSC_KNN_STEP_WORK"}, {"rescore_sort", "VSC_RESCORE_SORT"},
    {"knn_levels", "VSC_KNN_LEVELS"}, {"knn_subset", "VSC_KNN_SUBSET"}, {"knn_s0div", "VSC_KNN_S0DIV"},
    {"knn_s0min", "VSC_KNN_S0MIN"}, {"knn_ratio", "VSC_KNN_RATIO"}, {"knn_nchunk", "VSC_KNN_NCHUNK"},
    {"cand_budget", "VSC_CAND_BUDGET"}, {"debug_i8", "VSC_DEBUG_I8"}, {"debug_screen", "VSC_DEBUG_SCREEN"},
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
    if (is("cand_budget")) { if (!(v >= 1048576.0)) goto bad; idx->cand_budget = (int64_t)v; return VSC_OK; }
    if (is("debug_i8")) { idx->debug_i8 = v != 0.0; return VSC_OK; }
    if (is("debug_screen")) { idx->debug_screen = v != 0.0; return VSC_OK; }
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
    else if (is("cand_budget")) *out = (double)idx->cand_budget;
    else if (is("debug_i8")) *out = idx->debug_i8;
    else if (is("debug_screen")) *out = idx->debug_screen;
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
    (void)hipStreamSynchronize(idx->stream);
    idx->ref.release();
    idx->refh.release();
    idx->refn.release();
    idx->ref8.release();
    idx->ref8m.release();
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
    VSC_HIP(hipStreamSynchronize(idx->stream));  // nothing of this handle is left on the stream it leaves
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
static int i8_quantise(vsc_index* idx, int64_t row0, int64_t rows, int64_t count_rows) {
    if (rows <= 0) return VSC_OK;
    VSC_TRY(launch_quant_ref_frag(idx->ref.as<float>(), idx->dpad, idx->ref8.p, idx->ref8m.as<float4>(), row0, rows,
                                  idx->dpad8, idx->i8_ex, idx->stream));
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
    if (!same && first_new > 0) {
        idx->i8_ex = ex;
        idx->i8_dirty = true;  // the rows quantised so far left other coordinates out
        return VSC_OK;
    }
    idx->i8_ex = ex;
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

// Call before a search that may use the int8 kernel: brings the image up to date when the set of excluded
// coordinates changed since it was written.
static int i8_prepare(vsc_index* idx) {
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
static bool i8_usable(const vsc_index* idx) {
    if (idx->i8_mode == 2) return true;
    if (idx->i8_mode != 1) return false;
    if (idx->i8_loose_cnt <= 0.0) return true;
    return std::sqrt((double)idx->dim) * (idx->i8_loose_sum / idx->i8_loose_cnt) <= idx->i8_max_rel;
}

// Pack the query rows: returns device pointer; buffer holds round_up(nq,128)+128 zero-padded rows.
static int pack_queries(vsc_index* idx, const float* q, int64_t nq, int q_mem, float** out,
                        bool with_half = false) {
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

// entries of each candidate array for a candidate capacity of ccap (segments + chunked tail, see ensure_hit_buffers)
static inline int64_t cand_entries(int64_t ccap) { return 4 * ccap + 2048 * 128; }

// cap: kept hits (list A, and the compaction target B of the thresholded search); ccap: candidates of ONE
// pre-filter launch (defaults to cap)
static int ensure_hit_buffers(vsc_index* idx, int64_t cap, int64_t ccap = -1, bool need_b = true) {
    if (ccap < 0) ccap = cap;
    for (int c = 0; c < 3; ++c) {
        VSC_TRY(idx->ws.hA[c].reserve((size_t)cap * 4));
        if (need_b) VSC_TRY(idx->ws.hB[c].reserve((size_t)cap * 4));
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

}  // extern "C"

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

extern "C" {

// fp16 pre-filter + exact re-scoring of query rows [i0, i1): appends to hit buffer A every (row, ref, score)
// with score > *radius -- or, when `row_thr` (one threshold per query row, padded like the fp16 query
// image) is given, with score >= row_thr[row].
static int enqueue_f16(vsc_index* idx, const float* qpacked, int64_t i0, int64_t i1, int64_t cap,
                       const float* row_thr, int64_t ccap = -1, int64_t nr_limit = -1, bool use_i8 = false,
                       int64_t nr_begin = 0) {
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
            if (idx->i8_ex.n > 0) {
                // coordinates the images leave out (all references agree on them) act through the rows' thresholds:
                // t_row - sum_c q_c v_c, with t_row the row's k-NN threshold or the search radius
                VSC_TRY(idx->ws.rt8b.reserve((size_t)nqb * sizeof(float)));
                VSC_TRY(launch_row_bias_thresholds(qpacked + i0 * idx->dpad, idx->dpad, nqb, thr_src, &ctl->radius,
                                                   idx->i8_ex, idx->ws.rt8b.as<float>(), idx->stream));
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

static int init_ctl(vsc_index* idx, float radius_score_space) {
    SelectCtl h;
    memset(&h, 0, sizeof(h));
    h.radius = radius_score_space;
    VSC_HIP(hipMemcpyAsync(idx->ws.ctl.p, &h, sizeof(h), hipMemcpyHostToDevice, idx->stream));
    VSC_HIP(hipStreamSynchronize(idx->stream));  // h is a stack object
    return VSC_OK;
}

}  // extern "C"

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
            const bool f16 = idx->prefilter_force ||
                             (idx->prefilter && seen > 0 && (double)K < idx->prefilter_density * seen * (double)idx->ntotal);
            // ... and once it is low enough that the int8 kernel's 4-5x candidates cost less than the fp16 kernel's
            // second half (the bound of 8-bit rows is ~16x looser), the batch runs on int8
            const bool i8 = f16 && allow_i8 &&
                            (idx->i8_mode == 2 || (seen > 0 && (double)K < idx->i8_density * seen * (double)idx->ntotal));
            used_i8 |= i8;
            VSC_TRY(enqueue_batch(idx, qp, i0, i1, cap, f16, i8));
            hipEvent_t stop;
            VSC_TRY(prof_begin(idx, &stop, 3));
            VSC_TRY(enqueue_rethreshold(ctl, idx->ws.hA[0].as<int32_t>(), idx->ws.hA[1].as<int32_t>(),
                                        idx->ws.hA[2].as<float>(), idx->ws.hB[0].as<int32_t>(),
                                        idx->ws.hB[1].as<int32_t>(), idx->ws.hB[2].as<float>(),
                                        (unsigned long long)K, idx->stream));
            VSC_TRY(prof_end(idx, stop, 0.0, 3));
            if (!seeded && bs < 20000) bs *= 2;
            i0 = i1;
        }
        VSC_HIP(hipMemcpyAsync(&h, ctl, sizeof(h), hipMemcpyDeviceToHost, idx->stream));
        VSC_HIP(hipStreamSynchronize(idx->stream));
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
    VSC_TRY(sort_hits_topk(idx->ws.hA[0].as<int32_t>(), idx->ws.hA[1].as<int32_t>(), idx->ws.hA[2].as<float>(),
                           n, K, nq, idx->ws.w0, idx->ws.w1, idx->ws.w2, idx->ws.w3, idx->ws.tmp, di, dj, ds,
                           ip ? 0 : 1, &mm, idx->stream));
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

extern "C" {

int vsc_index_global_topk(vsc_index_t* idx, const float* q, int64_t nq, int q_mem, int64_t K,
                          int32_t* out_i, int32_t* out_j, float* out_s, int64_t cap_out, int out_mem,
                          int64_t* n_out, float* final_radius) {
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
                            ws.maps0.as<int32_t>(), ws.maps1.as<int32_t>(), 0, ws.w0, ws.w1, ws.w2, ws.w3, ws.tmp,
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
                 idx->ws.parts.as<float>(), idx->ws.partj.as<int32_t>()};
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
    VSC_HIP(hipStreamSynchronize(c->stream));
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
                            (const int32_t*)dr, 0, ws.w0, ws.w1, ws.w2, ws.w3, ws.tmp, ws.cnt, oq, orr, os, of,
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
    VSC_HIP(hipStreamSynchronize(c->stream));
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
    if (c->stream) (void)hipStreamSynchronize(c->stream);
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
