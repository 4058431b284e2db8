// Argument blocks and launchers of the HIP kernels (shared by the kernel translation units and
// the C-ABI layer in api.hip).
#pragma once
#include "vscmi_common.h"
#include "cand_list.h"

namespace vscmi {

struct SimThreshArgs {
    const float* Q; const float* R; int dpad; int nq; int i0; int nr; int tq; int tr;
    const float* radius; int32_t* out_i; int32_t* out_j; float* out_s;
    unsigned long long* counter; long long cap; int* overflow;
};
// fp16 pre-filter (sim_f16.hip): emits every (row, ref) whose fp16 score PLUS the tile's error bound
// exceeds *radius -- a superset of the exact hits; rescore_candidates then applies the exact test.
struct SimF16Args {
    const _Float16* Q; const _Float16* R;  // fp16 images [rows pad 256][dpadh], natural k order
    const float* qn; const float* rn;      // per-row upper bounds of the L2 norm (+inf: row not representable)
    int dpadh; int nq; int i0; int nr; int tq; int tr;
    float c1, c2, c3;                      // |fp16 score - exact score| <= c1*nq*nr + c2*(nq+nr) + c3
    const float* radius; int32_t* out_i; int32_t* out_j;
    // k-NN mode: per-row thresholds (indexed like qn; the test becomes fp16 score + bound >= row_thr[row]);
    // nullptr = thresholded search against *radius
    const float* row_thr;
    // candidate list = one private segment of seg_cap entries per wave of the launch (8 per workgroup)
    // + a shared tail (atomic counter) for waves whose segment is full
    int seg_cap; int* seg_count; int64_t tail_base; long long tail_cap; int tail_shift; int* tail_fill; unsigned long long* tail_count;
    int* overflow;
};
// panel-stationary fp16 pre-filter (sim_f16p.hip; dpadh <= 512): the query side is the natural fp16 image, the
// reference side the FRAGMENT-MAJOR image written by launch_pack_half_frag (layout.hip)
constexpr int F16P_PANEL_ROWS = 128;  // query rows per LDS-resident panel
constexpr int F16P_COL_STEP = 512;    // reference columns per workgroup step (8 waves x 64); row padding of the image
constexpr int F16P_MAX_DPADH = 512;   // the panel must fit the LDS: 128 rows x dpadh x 2 B <= 128 KiB
struct SimF16PArgs {
    const _Float16* Q; const void* Rf;
    const float* qn; const float* rn;      // per-row upper bounds of the L2 norm (+inf: row not representable)
    int dpadh; int nq; int i0; int nr;
    int npanel; int nsteps; int slice;     // work split (sim_f16p_plan); items = (panel, slice of col-steps)
    int* next_slice;                       // [npanel] slices handed out so far (zeroed by the launcher)
    float c1, c2, c3;
    const float* radius; const float* row_thr;  // as in SimF16Args
    int32_t* out_i; int32_t* out_j;
    int seg_cap; int* seg_count; int64_t tail_base; long long tail_cap; int tail_shift; int* tail_fill; unsigned long long* tail_count;
    int* overflow;
};
// panel-stationary INT8 pre-filter (sim_i8p.hip; dims <= 1024): the query side is the launch's own int8 image with one
// scale per 128-row panel (quant_query_panels), the reference side the fragment-major int8 image + per-row meta of
// launch_quant_ref_frag (quant_i8.hip)
constexpr int I8P_MAX_DPAD8 = 1024;    // the panel must fit the LDS: 128 rows x dpad8 B <= 128 KiB
// coordinates left out of the int8 images because every reference row holds the same value there (quant_i8.hip)
constexpr int I8_MAX_EXCLUDED = 8;
struct ExcludedDims {
    int n = 0;
    int idx[I8_MAX_EXCLUDED] = {};     // logical coordinate
    float val[I8_MAX_EXCLUDED] = {};   // the references' common value
    __host__ __device__ bool holds(int k) const {
        bool h = false;
#pragma unroll
        for (int c = 0; c < I8_MAX_EXCLUDED; ++c) h |= c < n && idx[c] == k;
        return h;
    }
};
struct SimI8PArgs {
    const void* Q; const float4* pstat;   // [npanel * 128][dpad8] int8; per panel {1 / s, max E, max N, max N'}
    const void* Rf; const float4* rmeta;  // fragment-major int8 image; per reference row {1 / s, E, N, N'}
    int dpad8; int nq; int i0; int nr;
    int npanel; int nsteps; int slice;     // work split (sim_f16p_plan)
    int* next_slice;                       // [npanel + 1]: per-panel slice counters + one global item counter
    int order;                             // 0: panel-major with stealing; 1: slice-major (all panels of a slice first)
    int pair;                              // 1: work items of TWO panels, wave tiles of 256 rows x 32 columns (sim_i8p_pairs)
    float c_acc;                           // rounding of the exact fp32 chain per |q||r|
    const float* radius; const float* row_thr;  // as in SimF16Args (row_thr indexed by POSITION inside the launch)
    int32_t* out_i; int32_t* out_j;
    int seg_cap; int* seg_count; int64_t tail_base; long long tail_cap; int tail_shift; int* tail_fill; unsigned long long* tail_count;
    int* overflow;
};
struct RescoreArgs {
    const float* Q; const float* R; int dpad;  // packed fp32 images (exact arithmetic contract)
    const int32_t* cand_i; const int32_t* cand_j; int n_seg; int seg_cap; const int* seg_count;
    int64_t tail_base; long long tail_cap; unsigned long long* tail_count;  // reset to 0 after the pass
    int tail_shift; const int* tail_fill;  // the tail is handed out in chunks of 1 << tail_shift entries, each with a fill level
    // the int8 kernel hands over POSITIONS inside its launch when the launch's rows were permuted (sorted by threshold
    // or by scale): row = perm_i0 + perm[position - perm_i0].  nullptr: the list holds rows
    const int32_t* perm; int perm_i0;
    unsigned long long* n_cand_total;      // statistics
    const float* radius; int32_t* out_i; int32_t* out_j; float* out_s;
    unsigned long long* counter; long long cap; int* overflow;
    const float* row_thr;  // k-NN mode: keep score >= row_thr[global row] (nullptr: score > *radius)
    // the launch searched the reference rows [j0, j0 + nr): the kernels' images were handed over from row j0 on, the
    // candidate list holds refs RELATIVE to it (cand_compact / the segment-wise exact stage add j0 back)
    int j0 = 0;
};
// fp16 screen of the int8 route's candidates (sim_f16.hip): the pair list sorted by reference row in, the pairs whose
// fp16 score + error bound still reaches the threshold out
struct ScreenArgs {
    const _Float16* Qh; const float* qn;  // row-major fp16 query image, norm bounds (absolute rows)
    const _Float16* Rh; const float* rn;  // reference image: fragment-major (frag) or row-major
    int dpadh; int frag;
    float c1, c2, c3;                     // |fp16 score - exact score| <= c1 |q||r| + c2 (|q| + |r|) + c3
    const float* radius; const float* row_thr;
    const uint32_t* sj; const uint32_t* si; long long n;
    uint32_t* out_j; uint32_t* out_i; unsigned long long* n_out;
    unsigned long long* n_cand_total;     // statistics: += n
    const int* overflow;
};
struct SimKnnArgs {
    const float* Q; const float* R; int dpad; int nq; int nr; int tq; int tr; int nchunk; int k;
    float* part_s; int32_t* part_j;
    int no_first_tile_select;  // option knn_first_tile = 0: k > 1 inserts every score of a run's first tile (A/B)
};
struct KnnMergeArgs {
    const float* part_s; const int32_t* part_j; int nq; int nchunk; int k; float* out_s; int64_t* out_j; int l2;
};
struct ScoreMatArgs { const float* Q; const float* R; int dpad; int dim; int nq; int nr; int metric; float* S; };
struct MatThreshArgs {
    const float* S; int nq; int nr; int i0; const float* radius; int32_t* out_i; int32_t* out_j; float* out_s;
    unsigned long long* counter; long long cap; int* overflow;
};
struct MatKnnArgs { const float* S; int nq; int nr; int k; float* part_s; int32_t* part_j; };
struct SelectCtl {
    unsigned long long n; unsigned long long n_tmp; float radius; int overflow; int active;
    unsigned int prefix; unsigned int prefix_mask; unsigned long long rank; unsigned int hist[256];
    unsigned long long n_rethreshold;
    unsigned long long n_cand_total;  // pre-filter candidates of the whole search (statistics)
    unsigned long long n_tail;        // fill level of the candidate list's shared tail (current batch)
};
struct TnPairArgs {
    const float* qfeat; const float* rfeat; const int64_t* q_off; const int64_t* r_off; int dpad;
    const int32_t* pair_q; const int32_t* pair_r; const int32_t* work; int n_work; vsc_tn_params prm;
    float bias; int max_lq; int lds_tile_floats; float* slab; int64_t slab_floats;
    int32_t* out_nbox; int32_t* out_boxes; float* out_boxmax;
    // forward_sim mode (sims_in != nullptr): precomputed matrices, pair p at sims_in + sims_off[p]
    const float* sims_in; const int64_t* sims_off; const int32_t* sims_lq; const int32_t* sims_lr;
    // over-long videos: per-workgroup working state in HBM (state_bytes each) instead of LDS; nullptr = LDS
    char* state; int64_t state_bytes;
};
struct TnSimsArgs { const float* qfeat; const float* rfeat; int64_t qrow0, rrow0; int lq, lr, dpad; float bias; float* out; };

int launch_sim_thresh(const SimThreshArgs&, hipStream_t);
int launch_sim_f16(const SimF16Args&, hipStream_t);
int sim_f16_grid(int tq, int tr);
int launch_rescore(const RescoreArgs&, hipStream_t);
int launch_cand_count(const RescoreArgs&, int, unsigned long long*, hipStream_t);
int launch_cand_compact(const RescoreArgs&, int, uint32_t*, uint32_t*, unsigned long long*, hipStream_t);
int launch_rescore_dense(const RescoreArgs&, const uint32_t*, const uint32_t*, long long, hipStream_t,
                         const unsigned long long* n_dev = nullptr);
int launch_f16_screen(const ScreenArgs&, hipStream_t);
int sort_candidates_by_ref(uint32_t*, uint32_t*, uint32_t*, uint32_t*, int64_t, int64_t, DevBuf&, const uint32_t**,
                           const uint32_t**, hipStream_t);
void sim_f16p_plan(int64_t nq, int64_t nr, int* npanel, int* nsteps, int* slice, int* grid);
int launch_sim_f16p(const SimF16PArgs&, int grid, hipStream_t);
int launch_sim_i8p(const SimI8PArgs&, int grid, hipStream_t);
bool sim_i8p_pairs(int dpad8, int npanel, int nsteps, int slice, bool force);  // does a launch of this size use work items of two panels?
int launch_quant_ref_frag(const float*, int, void*, float4*, int64_t, int64_t, int, const ExcludedDims&, const float*, hipStream_t);
int launch_col_sums(const float*, int64_t, int, double*, double*, hipStream_t);
int launch_row_center(const float*, int, int, const float*, float*, float*, hipStream_t);
int launch_dim_minmax(const float*, int64_t, int, unsigned*, unsigned*, hipStream_t);
int launch_meta_looseness(const float4*, int64_t, double*, hipStream_t);
int launch_quant_query_panels(const float*, int, int, int, void*, int, float4*, const int32_t*, const float*, float*,
                              const ExcludedDims&, hipStream_t);
int launch_row_absmax(const float*, int, int, const ExcludedDims&, float*, hipStream_t);
int launch_row_bias_thresholds(const float*, int, int, const float*, const float*, const ExcludedDims&, const float*, const float*, float*,
                               hipStream_t);
int sort_rows_by_threshold(const float*, int64_t, DevBuf&, DevBuf&, DevBuf&, DevBuf&, DevBuf&, const int32_t**, hipStream_t);
int argsort_scores_desc(const float*, int64_t, DevBuf&, DevBuf&, DevBuf&, DevBuf&, DevBuf&, const int32_t**, hipStream_t);
int launch_ctl_init(SelectCtl*, float, hipStream_t);
int launch_filter_hits(const int32_t*, const int32_t*, const float*, long long, float, int32_t*, int32_t*, float*, unsigned long long*,
                       hipStream_t);
int launch_score_hist(const float*, long long, const long long*, int, long long*, hipStream_t);
int launch_score_pick(const long long*, long long*, int, hipStream_t);
int launch_merge_topk(const float*, const long long*, long long, int, int, float*, long long*, hipStream_t);
int sort_rows_by_threshold_then_scale(const float*, const float*, int64_t, int, DevBuf&, DevBuf&, DevBuf&, DevBuf&, DevBuf&,
                                      const int32_t**, hipStream_t);
int launch_pack_half_frag(const float*, int64_t, int, _Float16*, float*, int64_t, int64_t, int, hipStream_t);
int launch_pack_half(const float*, int64_t, int, _Float16*, float*, int64_t, int, hipStream_t);
int launch_sim_knn(const SimKnnArgs&, hipStream_t);
int launch_knn_merge(const KnnMergeArgs&, hipStream_t);
int knn_from_hits(const int32_t*, const int32_t*, const float*, int64_t, int64_t, int, DevBuf&, DevBuf&, DevBuf&,
                  DevBuf&, DevBuf&, float*, int64_t*, hipStream_t);
int launch_knn_row_thr(const float*, int64_t, int, float*, int64_t, hipStream_t);
int launch_knn_seed_hits(const float*, const int64_t*, int64_t, int, int32_t*, int32_t*, float*, unsigned long long*, hipStream_t);
int launch_score_matrix(const ScoreMatArgs&, hipStream_t);
int launch_matrix_thresh(const MatThreshArgs&, hipStream_t);
int launch_matrix_knn(const MatKnnArgs&, hipStream_t);
int set_thresh_kernel_attrs();
int enqueue_rethreshold(SelectCtl*, int32_t*, int32_t*, float*, int32_t*, int32_t*, float*, unsigned long long, hipStream_t);
int sort_hits_topk(const int32_t*, const int32_t*, const float*, int64_t, int64_t, int64_t, int64_t, DevBuf&, DevBuf&,
                   DevBuf&, DevBuf&, DevBuf&, int32_t*, int32_t*, float*, int, int64_t*, hipStream_t);
int sort_hits_rowcol(const int32_t*, const int32_t*, const float*, int64_t, DevBuf&, DevBuf&, DevBuf&, DevBuf&,
                     DevBuf&, int32_t*, int32_t*, float*, int, hipStream_t);
int pair_max_device(const int32_t*, const int32_t*, const float*, int64_t, const int32_t*, const int32_t*, int64_t, int64_t,
                    DevBuf&, DevBuf&, DevBuf&, DevBuf&, DevBuf&, DevBuf&, int32_t*, int32_t*, float*, int64_t*,
                    int64_t, int64_t*, hipStream_t);
int launch_pack_rows(const float*, int64_t, int, float*, int64_t, int, hipStream_t);
int launch_row_normalize(const float*, int64_t, int, float*, hipStream_t);
size_t tn_state_bytes_host(int, int, int, int idx_bytes = 2);
int launch_tn_pairs(const TnPairArgs&, size_t, hipStream_t);
int launch_tn_sims(const TnSimsArgs&, hipStream_t);


}  // namespace vscmi
