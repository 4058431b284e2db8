// Internal declarations shared by the translation units of the C ABI (api.hip: handles, options, index upkeep;
// api_search.hip: the thresholded searches; api_knn.hip: k-NN; api_aux.hip: pair-max, row normalisation, Temporal
// Network).  Nothing here is exported: include/vscmi.h is the ABI.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "kernels.h"

namespace vscmi {

int check_device(int device);

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
    DevBuf rt8d;            // centred reference image: x . mu and sum |x mu| of the launch's rows
    DevBuf rt8, rt8b;       // ... and its row thresholds in position order (rows sorted by threshold inside a launch);
                            // rt8b: thresholds lowered by the excluded coordinates' contribution, in row order
    DevBuf sample, sk[3], tk[3];  // proven top-K route (api_search.hip): the row sample, its sorted hits, the K + 1 best of the steady run
    void release() {
        stage.release(); qbuf.release(); sample.release();
        for (auto& b : sk) b.release();
        for (auto& b : tk) b.release();
        qh.release(); qn.release(); ci.release(); cj.release(); segcnt.release(); rowthr.release(); slices.release();
        q8.release(); pstat.release(); rt8.release(); rt8b.release(); rt8c.release(); rt8d.release(); tailfill.release();
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

int pack_half_any(const float* x, int64_t n, int dim, const HalfImage& h, int64_t r0, int64_t rows_out, hipStream_t stream);
int pack_into(const float* x, int64_t n, int dim, int mem, float* dst, int64_t rows_out, int dpad, Workspace& ws,
              hipStream_t stream, const HalfImage& h = HalfImage());

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
    // Centre of the int8 reference image (quant_i8.hip, "CENTRED references"): the mean of the rows present when the image
    // was first written, in the rows' packed order; i8_mu is its device copy with zeros on the excluded coordinates
    // (re-uploaded when that set changes).  i8_center: 0 never, 1 when the mean carries >= 2 % of the rows' energy
    // (default), 2 always (tests).
    int i8_center = 1;
    bool i8_mu_decided = false, i8_mu_on = false;
    double i8_mu_ratio = 0.0;
    std::vector<float> i8_mu_host;
    DevBuf i8_mu;
    ExcludedDims i8_mu_ex;
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
    bool knn_first_tile = true;  // VSC_KNN_FIRST_TILE=0: no per-wave k-th-largest threshold on a run's first tile (k > 1)
    int knn_nchunk = 0;          // VSC_KNN_NCHUNK: reference chunks of the exact k-NN kernel (0: by size)
    bool debug_i8 = false, debug_screen = false;  // VSC_DEBUG_I8 / VSC_DEBUG_SCREEN: stderr notes
    // vsc_index_global_topk, optional route (api_search.hip: global_topk_proven): the exact top-K from a sampled seed radius
    // + steady batches, returned only when it is PROVEN to be the reference schedule's result (no tie on the K cut), else
    // the schedule is replayed.  OFF by default: at BASELINE's sizes a tie on the cut is certain (configs[3]: ~70 pairs per
    // fp32 value at the cut), the proof never succeeds there.  VSC_TOPK_SHORTCUT: 0 (default) the schedule, 1 large query
    // sets, 2 wherever the route is defined (tests)
    int topk_shortcut = 0;
    double density_hint = 0.0;        // option density_hint: expected hit density of the next thresholded searches (0: from K)
    bool sort_hits = true;            // VSC_SORT_HITS=0: thresholded searches return their kept hits in list order (a SET)
    int64_t topk_sample_rows = 4096;  // VSC_TOPK_SAMPLE: rows of the strided sample that seeds the radius
    int last_topk_route = 0;          // get_option("last_topk_route"): 0 schedule, 1 proven route, 2 proven route tried, schedule replayed
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

// kernel-time accounting of a handle (api.hip)
int prof_begin(vsc_index* idx, hipEvent_t* stop_out, int cls = 0);
int prof_end(vsc_index* idx, hipEvent_t stop, double work, int cls = 0);  // `work`: algorithmic flops / bytes of the launch
int prof_collect(vsc_index* idx);                                         // call after a stream sync

// Process-wide kernel-time accounting of the entry points that own no index handle (HIP events on the stream
// the kernels run on; read after the call's own stream sync): 0 = vsc_pair_max, 1 = Temporal-Network launches.
struct AuxProf {
    bool on = false;
    std::mutex mu;
    double ms[2] = {}, bytes[2] = {};
    int64_t n[2] = {};
};
extern AuxProf g_aux;
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

// index upkeep (api.hip)
int i8_prepare(vsc_index* idx);         // before a search that may use the int8 kernel: bring the image up to date
bool i8_usable(const vsc_index* idx);   // may this search use the int8 kernel at all?
int pack_queries(vsc_index* idx, const float* q, int64_t nq, int q_mem, float** out, bool with_half = false);
// entries of the two candidate arrays for a list of `ccap` entries in segments (cand_list.h: + the chunked tail)
inline int64_t cand_entries(int64_t ccap) { return 4 * ccap + 2048 * 128; }
int ensure_hit_buffers(vsc_index* idx, int64_t cap, int64_t ccap = -1, bool need_b = true);

// the batches of the thresholded searches (api_search.hip), shared with the k-NN's threshold passes
int enqueue_f16(vsc_index* idx, const float* qpacked, int64_t i0, int64_t i1, int64_t cap, const float* row_thr,
                int64_t ccap = -1, int64_t nr_limit = -1, bool use_i8 = false, int64_t nr_begin = 0);
int init_ctl(vsc_index* idx, float radius_score_space);
