/*
 * vscmi.h -- C ABI of libvscmi.so: the MI355X (gfx950) descriptor-search / candidate /
 * temporal-localisation engine that sits under the vsc.index / vsc.candidates / vcsl.vta call
 * surface of facebookresearch/vsc2022.
 *
 * Conventions
 *   - Plain pointers and sizes only.  Every array is owned by the caller; the library frees
 *     nothing it did not allocate and hands back no memory except opaque handles that have an
 *     explicit *_destroy.
 *   - Each array argument carries a memory kind: VSC_MEM_HOST (pageable host memory, e.g. a
 *     numpy array) or VSC_MEM_DEVICE (HBM of the handle's device, e.g. a torch tensor's
 *     data_ptr).  Scalars written through pointers (counts, radii) are always host memory.
 *   - All feature matrices are fp32, C-contiguous, row-major [n][dim]; frame-row and video
 *     ordinals are int32 (row counts < 2^31).
 *   - Return value: VSC_OK (0) or a negative VSC_ERR_* code; vsc_last_error() returns a
 *     thread-local message.  The Python layer raises RuntimeError / ValueError from these,
 *     matching the reference's exception/assert convention (vsc/index.py:37-40,
 *     vsc/storage.py:49-57, vsc/baseline/localization.py:59,64).
 *   - A handle owns one HIP stream; calls on one handle are serialised by the caller, distinct
 *     handles may be used from distinct threads.  No callbacks into the caller.
 *   - Arithmetic contract: every similarity is the fp32 fma chain in ascending k
 *     (acc = fmaf(q[k], r[k], acc), acc0 = +0), produced on the matrix cores by
 *     v_mfma_f32_32x32x2_f32.  Results are bit-identical to oracle/vsc_oracle.c.
 *
 * Paths below are relative to the reference checkout (/root/reference).
 */
#ifndef VSCMI_H
#define VSCMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSC_OK 0
#define VSC_ERR_INVALID (-1)  /* bad argument */
#define VSC_ERR_HIP (-2)      /* HIP runtime failure */
#define VSC_ERR_NOMEM (-3)    /* device allocation failed */
#define VSC_ERR_CAPACITY (-4) /* caller-provided output capacity too small; *n_out holds the need */
#define VSC_ERR_OVERFLOW (-5) /* internal hit buffer overflowed (pathological score distribution) */
#define VSC_ERR_NODEVICE (-6) /* no gfx950 device */

#define VSC_MEM_HOST 0
#define VSC_MEM_DEVICE 1

/* same numeric values as faiss.METRIC_INNER_PRODUCT / faiss.METRIC_L2 (vsc/index.py:78,145) */
#define VSC_METRIC_INNER_PRODUCT 0
#define VSC_METRIC_L2 1

#define VSC_TN_MAX_BOXES 16

typedef struct vsc_index vsc_index_t;

const char* vsc_last_error(void);
int vsc_version(void);
/* number of visible HIP devices whose arch is gfx950 (0 => the library cannot run) */
int vsc_device_count(void);

/* ---------------------------------------------------------------- flat index
 * Replaces faiss.index_factory(dim, "Flat", metric) + index.add(x) (vsc/index.py:82,94).
 * The reference set stays resident in HBM across searches; add is incremental.
 *
 * Every score the library returns is the fp32 chain acc = fmaf(q[k], r[k], acc), k ascending.  For
 * inner-product indexes the thresholded searches and the k-NN first evaluate the score matrix at reduced precision
 * on the matrix cores -- int8 (v_mfma_i32_16x16x64_i8, dims <= 1024) where hits are sparse, fp16 otherwise -- and
 * hand to that exact stage every pair whose low-precision score plus a rigorous error bound reaches the threshold;
 * the outputs are bit-identical to the all-fp32 route.
 *
 * Options.  Every switch below is an option of the handle: `vsc_index_set_option(idx, "<name>", value)` with the name in
 * lower case and without the VSC_ prefix ("i8_density", "prefilter", "knn_step", ...; "f16_kernel": 1 = ring), and the
 * environment variable of the same name supplies its initial value when a handle is created (read once, then it stays
 * with that handle):
 *   VSC_PREFILTER=0           no pre-filter (every search on the exact fp32 MFMA kernel); =2 forces it onto every
 *                             batch / every k-NN regardless of size (tests)
 *   VSC_PREFILTER_DENSITY=f   expected hit density below which a batch of the thresholded search is pre-filtered
 *                             (default 0.05)
 *   VSC_F16_KERNEL=ring       the 256x256 LDS-ring fp16 pre-filter instead of the panel-stationary one (A/B)
 *   VSC_I8=0                  no int8 image (fp16 pre-filter only); =2 forces the int8 kernel onto every
 *                             pre-filtered batch (tests)
 *   VSC_I8_DENSITY=f          expected hit density below which a pre-filtered batch runs on int8 (default 5e-4)
 *   VSC_I8_MAX_REL=f          sqrt(dim) x mean(E_r / N'_r) of the references above which the index never starts
 *                             on int8 (default 0.35: the 8-bit bound would pass too much of the matrix)
 *   VSC_I8_EXCLUDE=0          keep coordinates on which all references agree inside the int8 images
 *   VSC_I8_CENTER=0|1|2       the int8 reference image holds y - mu, mu = the mean of the rows present at the first search over
 *                             >= 1024 rows (fixed from then on); the rows' x . mu moves their thresholds (exact: x.y = x.(y - mu) + x.mu).  1 (default):
 *                             when the mean carries >= 2 % of the rows' energy (uncentred embeddings; isotropic rows are left
 *                             alone), 0 never, 2 always (tests).  Read-only options "i8_center_on", "i8_center_share"
 *   VSC_I8_SORT=0             int8 launches see their rows in batch order (default: sorted by threshold / scale)
 *   VSC_I8_GROUP=n            radius searches with per-row thresholds (excluded coordinates): inside groups of 2^n
 *                             rows of the threshold order the rows are ordered by scale (default 9; 0: off)
 *   VSC_I8P_ORDER=0           panel-major work items with stealing (default 1: slice-major)
 *   VSC_I8P_SLICE=n           col-steps of 512 reference rows per work item (default 32 slice-major)
 *   VSC_I8P_PAIR=0 / 2        never / always (dims <= 512) use work items of two 128-row panels (256 x 32 wave tiles;
 *                             default 1: where a launch is large enough)
 *   VSC_I8_SCREEN=1           fp16 screen between the int8 pre-filter and the exact stage (measured neutral: off)
 *   VSC_KNN_STEP=n            query rows per launch of a k-NN threshold pass (default: 32768, doubled until rows x range
 *                             reaches VSC_KNN_STEP_WORK (64) x 32768 x 196608 or VSC_KNN_STEP_MAX (262144) rows)
 *   VSC_I8_KNN=0              k-NN threshold passes on the fp16 kernel
 *   VSC_RESCORE_SORT=0        exact stage over the waves' candidate segments as they are (default: compacted and
 *                             sorted by reference row)
 *   VSC_KNN_LEVELS=1          pre-filtered k-NN with one refinement level; VSC_KNN_SUBSET=<factor> (default 300),
 *   VSC_KNN_S0DIV=<n> (28), VSC_KNN_S0MIN=<rows> (1024), VSC_KNN_RATIO=<r> (by k), VSC_KNN_NCHUNK=<n>: sizes of its exact subset pass / levels
 *   VSC_KNN_FIRST_TILE=0      exact k-NN kernel, k > 1: insert every score of a run's first tile (default: only those that reach
 *                             a per-wave lower bound of the row's k-th largest score of that tile; A/B)
 *   VSC_DEBUG_I8 / VSC_DEBUG_SCREEN: notes on stderr when a search falls back from int8 / per screen launch
 *   VSC_TOPK_SHORTCUT=0|1|2   proven top-K route of vsc_index_global_topk (see there); VSC_TOPK_SAMPLE=<rows> (4096)
 *   density_hint=<d>          (option only) expected hit density of the batches of the next thresholded searches; > 0: the
 *                             exact / fp16 / int8 rule uses it instead of K / (rows x references) -- the sharded schedule calls the
 *                             seeded search once per batch with a generous budget K and knows the density the batch will have
 *   VSC_SORT_HITS=0           vsc_index_global_topk / _seeded (inner product) return their hits as a SET, in the kept
 *                             list's order, instead of (score desc, row asc, ref asc): the column-sharded schedule joins
 *                             a batch's hits to a list that is sorted once at the end.  min(n, K) entries come back; K
 *                             of them may be a truncated list
 * Process-wide (first use): VSC_SIM_GRID (persistent grid of the exact similarity kernel), VSC_POISON_ALLOC=1
 * (fresh device buffers filled with 0xFF). */
int vsc_index_create(int dim, int metric, int device, vsc_index_t** out);
int vsc_index_destroy(vsc_index_t* idx);
/* Programmatic form of the switches above (the reference's analogue: faiss.ParameterSpace().set_index_parameter(index,
 * name, value) on the object vsc/index.py:82 creates).  Options that decide which images of the reference rows are kept
 * ("prefilter" 0 <-> non-0, "i8" 0 <-> non-0, "f16_kernel", "i8_exclude", "i8_center") can only change while the index is empty:
 * VSC_ERR_INVALID otherwise, as for an unknown name or an out-of-range value.  "cand_budget": entries of the candidate
 * list a k-NN threshold pass may ask for (default 2^28; the rows per launch are halved until it fits). */
int vsc_index_set_option(vsc_index_t* idx, const char* name, double value);
int vsc_index_get_option(const vsc_index_t* idx, const char* name, double* value);
/* Run every launch and copy of this handle on the caller's HIP stream (a hipStream_t, e.g. torch's current stream) instead
 * of the handle's own non-blocking stream: work the caller queued on that stream before a call -- the kernel that
 * produced the query rows -- is then ordered before the library's reads without a device-wide synchronisation.
 * hip_stream = NULL is HIP's default stream (torch's default); own != 0 goes back to the handle's own stream (hip_stream
 * is ignored then).  Calls still return only when their results are complete.
 * Lifetime: a bound stream must stay alive while a call of this handle runs on it; between calls nothing of the handle
 * is pending on it.  When the handle leaves a caller's stream (another vsc_index_set_stream, vsc_index_destroy) the
 * library does NOT touch that stream again -- it may already be destroyed -- and drains the device instead. */
int vsc_index_set_stream(vsc_index_t* idx, void* hip_stream, int own);
int vsc_index_add(vsc_index_t* idx, const float* x, int64_t n, int x_mem);
int64_t vsc_index_ntotal(const vsc_index_t* idx);
int vsc_index_dim(const vsc_index_t* idx);
int vsc_index_metric(const vsc_index_t* idx);
/* Capacity (entries) of the internal kept-hit buffer used by vsc_index_global_topk; 0 restores the
 * default max(32*ntotal, 2K) + 2K.  A search that overflows it returns VSC_ERR_OVERFLOW. */
int vsc_index_set_hit_capacity(vsc_index_t* idx, int64_t cap);
/* block until all work queued on the handle's stream has finished */
int vsc_index_sync(vsc_index_t* idx);

/* Replaces faiss index.search(x, k) (vsc/index.py:174; vsc/baseline/score_normalization.py:96).
 * Per query row the k best refs ordered by (score desc, ref asc) [L2: (dist asc, ref asc)].
 * out_s[nq*k] fp32, out_j[nq*k] int64 (faiss idx_t); missing slots hold -1 and -/+FLT_MAX.
 * k <= 4096; k <= 64 runs on the MFMA kernels (fp16 pre-filter + exact stage for large problems), larger k on an
 * explicit score matrix (same fp32 chains, same order; O(k * nr) per row). */
int vsc_index_knn(vsc_index_t* idx, const float* q, int64_t nq, int q_mem, int k, float* out_s,
                  int64_t* out_j, int out_mem);

/* Replaces faiss index.range_search(x, radius) (reached through
 * faiss.contrib.exhaustive_search.range_search_max_results at vsc/index.py:147-154): all (row, ref)
 * with score > radius (IP) or dist < radius (L2), STRICT.  Rows ascending, refs ascending within a
 * row.  lims[nq+1] cumulative counts; D/I written when cap >= *n_out, else VSC_ERR_CAPACITY (*n_out
 * = required).  All outputs host memory. */
int vsc_index_range_search(vsc_index_t* idx, const float* q, int64_t nq, int q_mem, float radius,
                           int64_t* lims, float* D, int64_t* I, int64_t cap, int64_t* n_out);

/* Replaces VideoIndex._global_threshold_knn_search (vsc/index.py:142-165): the adaptive
 * global-threshold search of range_search_max_results(exponential_query_iterator(Q), radius=-/+1e10,
 * max_results=2K, min_results=K) followed by the stable sort + truncate to K.  Reproduces the
 * reference's batch schedule (32, 64, ... rows) and strict re-thresholding, ties included.
 * Output: <= K hits ordered by (score desc, row asc, ref asc) [L2: dist asc].
 * out_* capacity `cap` (>= K suffices).  *final_radius receives the last radius.
 *
 * Optional proven route (option "topk_shortcut" / VSC_TOPK_SHORTCUT: 0 = never [default], 1 = inner-product query sets of
 * >= 65536 rows and >= 4e10 pairs, 2 = wherever it is defined [tests]).  What the reference returns is the first K of
 * {s > tau_final} in sorted order, and tau_final -- the (K+1)-th best score of the row PREFIX its last re-threshold event
 * saw -- cannot exceed s_(K+1), the (K+1)-th best score of the whole matrix; so whenever s_K > s_(K+1) the reference's
 * result IS the exact top-K, whatever its batch schedule did on the way.  The route computes that top-K without the
 * doubling batches (the schedule over a strided sample of <= "topk_sample" = 4096 rows seeds a radius just below the cut;
 * all rows then run as steady 32768-row batches from it with the budget K + 1), checks s_K > s_(K+1) on the result and
 * returns it only then; with a tie on the cut, a seed that turned out too high or an overflow it replays the schedule as
 * above.  Results are bit-identical either way (tests/test_gpu_topk_proven.py).  Off by default because the proof cannot
 * succeed at BASELINE's sizes: with 2e12 scores and K = 48 M about 70 pairs share every fp32 value at the cut (4.6e9 pairs
 * per unit of score x 1.5e-8 per ulp), a tie on the cut is certain and the schedule's own final radius decides (DESIGN.md
 * section 8).  On the proven route *final_radius is the steady run's last radius (every returned hit lies above it).
 * get_option("last_topk_route"): 0 schedule, 1 proven, 2 tried + replayed. */
int vsc_index_global_topk(vsc_index_t* idx, const float* q, int64_t nq, int q_mem, int64_t K,
                          int32_t* out_i, int32_t* out_j, float* out_s, int64_t cap, int out_mem,
                          int64_t* n_out, float* final_radius);

/* The same search for a caller that already knows a radius below the K-th best score (the query-sharded pipeline,
 * vsc2022_amd/dist.py: every rank seeds its local search with a radius agreed over a row sample -- what FAISS gives
 * the reference when an index is spread over GPUs, vsc/index.py:153): the rows run as steady 32768-row batches from
 * `radius0` (a score for inner product, a distance for L2) instead of replaying the doubling schedule from -/+1e10.
 * Returns every hit STRICTLY beyond max(radius0, the re-threshold radii) -- the re-threshold rule of
 * range_search_max_results stays active (more than 2K kept: radius <- (K+1)-th best), so a seed that is too low costs
 * time, not memory -- ordered and truncated like vsc_index_global_topk; *final_radius = the last radius: the list is
 * complete beyond it.  Fewer than K hits come back when the seed was too high: the caller falls back to the unseeded
 * search.  radius0 must be finite. */
int vsc_index_global_topk_seeded(vsc_index_t* idx, const float* q, int64_t nq, int q_mem, int64_t K, float radius0,
                                 int32_t* out_i, int32_t* out_j, float* out_s, int64_t cap, int out_mem,
                                 int64_t* n_out, float* final_radius);

/* Fused candidate generation: vsc_index_global_topk followed by vsc_pair_max without the hit list
 * leaving HBM -- the whole of CandidateGeneration.query with MaxScoreAggregation
 * (vsc/candidates.py:36-40 over vsc/index.py:96-165).  row2q[nq] / row2r[ntotal] (host) map frame
 * rows to video ordinals.  Outputs (host, capacity cap; cap >= min(K, nq*ntotal) suffices): pairs in
 * score-descending stable order.  *n_hits receives the number of frame hits found (<= K). */
int vsc_index_candidates(vsc_index_t* idx, const float* q, int64_t nq, int q_mem, int64_t K,
                         const int32_t* row2q, const int32_t* row2r, int32_t* out_q, int32_t* out_r,
                         float* out_s, int64_t cap, int64_t* n_pairs, int64_t* n_hits);

/* ------------------------------------------------------- candidate generation
 * Replaces the regroup loop of VideoIndex.search (vsc/index.py:121-140) fused with
 * MaxScoreAggregation + the stable descending sort of CandidateGeneration.query
 * (vsc/candidates.py:24-40).  hits must be in search order (as produced by
 * vsc_index_global_topk).  row2q[nq_rows] / row2r[nr_rows] map frame rows to video ordinals.
 * Output pairs in first-appearance order (== score-descending, stable): video ordinals, max score
 * and the index of the pair's first hit.  cap >= n suffices.  device selects the GPU. */
int vsc_pair_max(const int32_t* hit_i, const int32_t* hit_j, const float* hit_s, int64_t n,
                 int hits_mem, const int32_t* row2q, int64_t nq_rows, const int32_t* row2r,
                 int64_t nr_rows, int maps_mem, int32_t* out_q, int32_t* out_r, float* out_s,
                 int64_t* out_first, int64_t cap, int out_mem, int64_t* n_pairs, int device);

/* The final ordering of a hit list on its own: (score desc, query row asc, reference row asc), the order in which
 * VideoIndex._global_threshold_knn_search flattens and stably sorts its hits (vsc/index.py:158-165) -- for callers that
 * assemble a hit list themselves (the sharded schedule, vsc2022_amd/engine.py: the kept hits of the column slices arrive at
 * the rank that owns their query rows in no particular order).  Arrays of n entries, host or device; out_* may not alias the
 * inputs.  max_row / max_ref: exclusive upper bounds of the row / reference numbers (<= 0: unknown, all 31 bits are
 * sorted); they only save sort passes. */
int vsc_sort_hits(const int32_t* hit_i, const int32_t* hit_j, const float* hit_s, int64_t n, int hits_mem,
                  int64_t max_row, int64_t max_ref, int32_t* out_i, int32_t* out_j, float* out_s, int out_mem,
                  int device);

/* Exact order statistics over UNSORTED score lists that are spread over ranks -- the re-threshold events of
 * range_search_max_results (vsc/index.py:147-154: "more than 2K kept -> radius = the (K+1)-th best") when the kept list
 * lives on several GPUs (vsc2022_amd/dist.py:kth_best_unsorted), without sorting anything: a 4 x 8-bit radix select whose
 * per-level histograms are summed by the CALLER (one all-reduce of 256 counters per level).
 *   state   int64[4] in device memory: {key prefix found so far, its mask, 1-based rank still wanted among the keys that
 *           match the prefix, scores strictly above the prefix so far}; start with {0, 0, k, 0}.
 *   vsc_score_histogram   hist[256] (device int64) <- per-digit counts, digit = bits [shift, shift+8) of the key, over the
 *           scores whose key matches state's prefix; shift = 24, 16, 8, 0 in that order.  key = order-preserving image of
 *           score + 0.0f (-0.0 and +0.0 share a key, as they compare equal in the reference's float comparisons).
 *   vsc_score_pick        narrows state by the (summed) histogram of that level.
 * After the level with shift 0 state[0] is the key of the k-th best score of the union and state[3] the number of scores
 * strictly above it; the sum of the first level's histogram is the size of the union (k beyond it: state is meaningless).
 * Device pointers only.  On a caller's stream (vsc_set_aux_stream) both calls only ENQUEUE -- the caller's next operation on
 * that stream is ordered behind them and a whole selection runs without a host round trip --; on the library's own
 * stream they return when done. */
int vsc_score_histogram(const float* scores, int64_t n, const int64_t* state, int shift, int64_t* hist, int device);
int vsc_score_pick(const int64_t* hist, int64_t* state, int shift, int device);

/* The other half of a re-threshold event (apply_maxres inside faiss.contrib.exhaustive_search, reached at vsc/index.py:147-154:
 * "re-filter every kept batch with a strict s > radius") for a kept list the caller holds in HBM: out_* <- the (row, ref,
 * score) triples with score > radius (STRICT), in no particular order; *n_out (host) their number.  Device arrays of n entries;
 * out_* of capacity n, not aliasing the inputs. */
int vsc_filter_hits(const int32_t* hit_i, const int32_t* hit_j, const float* hit_s, int64_t n, float radius, int32_t* out_i,
                    int32_t* out_j, float* out_s, int64_t* n_out, int device);

/* perm[n] (int32) <- the stable argsort of a score list, best first: equal scores (-0.0 == +0.0) keep their input order.
 * Merges the ranks' candidate lists (vsc/candidates.py:38-40 sorts with Python's stable sort; the concatenation of the
 * ranks' lists in rank order IS first-appearance order, vsc2022_amd/dist.py:merge_candidates).  Host or device arrays. */
int vsc_argsort_scores(const float* scores, int64_t n, int mem, int32_t* perm, int perm_mem, int device);

/* Per-row merge of k-NN lists from reference shards (BASELINE configs[4]; vsc2022_amd/dist.py:ref_sharded_knn): row x of
 * scores / ids holds m candidates (ids: GLOBAL reference rows, < 0 = empty slot); out_* [nq, k] <- the k best per row
 * under (score desc, id asc) -- what faiss index.search returns on the concatenated reference set (vsc/index.py:174);
 * missing slots hold -1 and -FLT_MAX.  1 <= k <= m <= 1024; device pointers. */
int vsc_merge_topk(const float* scores, const int64_t* ids, int64_t nq, int m, int k, float* out_s, int64_t* out_ids,
                   int device);

/* The stream of the entry points that own no handle on `device` (vsc_pair_max, vsc_sort_hits, vsc_score_histogram, vsc_score_pick,
 * vsc_filter_hits, vsc_argsort_scores, vsc_merge_topk, vsc_row_normalize, vsc_tn_forward_sim):
 * as vsc_index_set_stream. */
int vsc_set_aux_stream(int device, void* hip_stream, int own);

/* Replaces sklearn.preprocessing.normalize(x) (row L2; vsc/baseline/score_normalization.py:84,
 * vsc/baseline/sscd_baseline.py:129-130): out = x / max(||x||, 0 -> 1). */
int vsc_row_normalize(const float* x, int64_t n, int dim, int x_mem, float* out, int out_mem,
                      int device);

/* --------------------------------------------------- temporal localisation
 * Replaces VCSLLocalization over batches of candidates (vsc/baseline/localization.py:28-96):
 * per pair sims = q.feature @ r.feature.T + bias (localization.py:36,52-54), the VCSL
 * Temporal-Network aligner reached through model.forward_sim (localization.py:58; vcsl.vta `tn`,
 * third-party -- see DESIGN.md for the unpinned-parity note), and the MaxSim box score
 * (localization.py:88-91).
 *
 * A context keeps the query and reference descriptors of LocalizationWithMetadata.__init__
 * (localization.py:29-31) resident in HBM: qfeat[total_q_rows][dim] with q_off[n_qvid+1] row
 * offsets per query video; same for refs. */
typedef struct vsc_tn_ctx vsc_tn_ctx_t;

typedef struct vsc_tn_params {
    int32_t tn_max_step; /* VCSL default 10; reference passes 5 (sscd_baseline.py:122,132) */
    int32_t tn_top_k;    /* 5 */
    int32_t max_path;    /* 10 */
    int32_t min_length;  /* VCSL default 5; reference passes 4 */
    float min_sim;       /* 0.2 */
    float max_iou;       /* 0.3 */
} vsc_tn_params;

int vsc_tn_create(const float* qfeat, const int64_t* q_off, int64_t n_qvid, const float* rfeat,
                  const int64_t* r_off, int64_t n_rvid, int dim, int feat_mem, int device,
                  vsc_tn_ctx_t** out);
int vsc_tn_destroy(vsc_tn_ctx_t* ctx);
/* As vsc_index_set_stream, for a localisation context. */
int vsc_tn_set_stream(vsc_tn_ctx_t* ctx, void* hip_stream, int own);
/* Replace the QUERY side of a context (a new query batch against the same, resident references: the per-query-set
 * step of vsc/baseline/sscd_baseline.py:90-176 when many query sets meet one reference set).  The reference rows
 * stay packed in HBM; only the nq query rows are uploaded and packed.  Same argument meaning as vsc_tn_create. */
int vsc_tn_set_queries(vsc_tn_ctx_t* ctx, const float* qfeat, const int64_t* q_off, int64_t n_qvid, int feat_mem);

/* One localize_all batch.  pair_q/pair_r[n_pairs] are video ordinals.  Outputs (host or device):
 *   out_nbox[n_pairs]                       number of boxes (<= VSC_TN_MAX_BOXES)
 *   out_boxes[n_pairs][VSC_TN_MAX_BOXES][4] q_lo, r_lo, q_hi, r_hi (frame indices, inclusive ends)
 *   out_boxmax[n_pairs][VSC_TN_MAX_BOXES]   max(sims[q_lo:q_hi, r_lo:r_hi]) - bias (half-open slice,
 *                                           localization.py:91); -inf for an empty slice */
int vsc_tn_localize(vsc_tn_ctx_t* ctx, const int32_t* pair_q, const int32_t* pair_r, int64_t n_pairs,
                    int pairs_mem, const vsc_tn_params* params, float bias, int32_t* out_nbox,
                    int32_t* out_boxes, float* out_boxmax, int out_mem);

/* vcsl.vta.build_vta_model("TN").forward_sim([(name, sims), ...]) (vsc/baseline/localization.py:58)
 * on caller-supplied similarity matrices (host memory): pair p is the row-major lq[p] x lr[p] fp32
 * matrix at sims + sims_off[p]; sims_off[n_pairs] is the total length.  Outputs (host) as in
 * vsc_tn_localize with bias = 0. */
int vsc_tn_forward_sim(const float* sims, const int64_t* sims_off, const int32_t* lq, const int32_t* lr,
                       int64_t n_pairs, const vsc_tn_params* params, int32_t* out_nbox,
                       int32_t* out_boxes, float* out_boxmax, int device);

/* LocalizationWithMetadata.similarity / VCSLLocalization.similarity (localization.py:33-36,48-54)
 * for one pair: out[lq*lr] (host) = q.feature @ r.feature.T + bias; *lq / *lr receive the shape.
 * cap is the capacity of out in floats (VSC_ERR_CAPACITY if too small). */
int vsc_tn_similarity(vsc_tn_ctx_t* ctx, int32_t q_vid, int32_t r_vid, float bias, float* out,
                      int64_t cap, int32_t* lq, int32_t* lr);

/* ---------------------------------------------------------- instrumentation
 * Kernel-time accounting for bench.py: HIP events recorded on the handle's stream around the
 * dominant similarity kernel.  vsc_index_profile(idx, 1) enables it; vsc_index_profile_read
 * returns accumulated kernel milliseconds, launches and algorithmic flops since the last reset. */
int vsc_index_profile(vsc_index_t* idx, int enable);
int vsc_index_profile_read(vsc_index_t* idx, double* sim_ms, int64_t* sim_launches, double* sim_flops,
                           int reset);
/* Same accounting per kernel class: 0 = exact fp32 similarity kernels (what vsc_index_profile_read
 * reports), 1 = fp16 pre-filter GEMM (work = algorithmic flops 2*nq*nr*dim), 2 = exact re-scoring of
 * the pre-filter's candidates (work unused), 3 = re-threshold kernels of the search schedule (radix select +
 * compaction; work unused), 4 = final ordering of the kept hits (work = bytes of the hit triples read),
 * 5 = int8 pre-filter kernel of the sparse batches (work = 2*nq*nr*dim), 6 = the preamble of its launches (row
 * thresholds / scales, the sort of the launch's rows, quantisation of the query panels; work unused). */
int vsc_index_profile_read_class(vsc_index_t* idx, int cls, double* ms, int64_t* launches, double* work,
                                 int reset);
/* Process-wide accounting of the entry points that own no index handle, same method (HIP events on the
 * stream the kernels run on): cls 0 = vsc_pair_max (vsc/candidates.py:24-40 on device hits), cls 1 = the
 * Temporal-Network launches of vsc_tn_localize / vsc_tn_forward_sim (vcsl.vta TN.forward_sim).
 * bytes = algorithmic bytes of the calls (4*dim*(Lq+Lr) per pair + boxes; 12 B per hit
 * + 20 B per pair).  bench.py only. */
int vsc_aux_profile(int enable);
int vsc_aux_profile_read(int cls, double* ms, int64_t* calls, double* bytes, int reset);
/* Counters of the last thresholded search on this handle: pairs the fp16 pre-filter passed on to
 * the exact stage, and (reserved) hits. */
int vsc_index_search_stats(vsc_index_t* idx, int64_t* candidates, int64_t* hits);

/* ------------------------------------------------- frame inference
 * Epilogue of one convolution of the SSCD trunk with its BatchNorm folded in (what
 * vsc/baseline/inference_impl.py:210-239 runs through TorchScript as conv -> bn -> (+ identity) -> relu):
 * y[r, c] = act(y[r, c] + bias[c] (+ res[r, c])), bf16 device arrays [rows, cols] (NHWC activations, cols =
 * channels, a multiple of 8), fp32 bias, fp32 arithmetic with one rounding; res may be NULL; relu != 0 applies
 * max(., 0).  In place on y, on the HIP stream `hip_stream` (NULL = the default stream).  Device pointers only. */
int vsc_bias_act_bf16(void* y, const void* res, const float* bias, int64_t rows, int64_t cols, int relu,
                      void* hip_stream);
/* A 3x3 (padding 1, stride 1 or 2; taps = 9) or 1x1 (taps = 1, stride 1) convolution of that trunk as an implicit GEMM
 * on the matrix cores with the epilogue inside: out[b, ho, wo, n] = act(sum x[b, hi, wi, c] * w[n, tap, c] + bias[n]
 * (+ res[b, ho, wo, n])); x [B, H, W, C] NHWC, w [N, taps, C] (= the channels-last memory of the PyTorch weight
 * [N, C, kh, kw]), res / out [B, Ho, Wo, N] with Ho = (H - 1) / stride + 1: bf16 device arrays; bias fp32; C and N
 * multiples of 64.  fp32 accumulation, one rounding.  Same stream convention. */
int vsc_conv_bias_act_bf16(const void* x, const void* w, const float* bias, const void* res, void* out, int64_t B,
                           int64_t H, int64_t W, int64_t C, int64_t N, int taps, int stride, int relu, void* hip_stream);
/* The stem's tail in one pass: out = maxpool3x3(stride 2, padding 1)(relu(x + bias)); x [N, H, W, C], out
 * [N, (H-1)/2+1, (W-1)/2+1, C] bf16 NHWC device arrays, C a multiple of 8; bit-identical to the two separate passes. */
int vsc_pool3x3s2_bias_relu_bf16(const void* x, const float* bias, void* out, int64_t N, int64_t H, int64_t W,
                                 int64_t C, void* hip_stream);
/* A 1x1 convolution of that trunk with its epilogue in one kernel:
 * out[m, n] = act(sum_k a[m, k] * w[n, k] + bias[n] (+ res[m, n])); a [M, K] = NHWC activations (M = batch * H * W),
 * w [N, K] = the convolution's weight as stored (Cout x Cin), res / out [M, N]: bf16 device arrays, bias fp32;
 * fp32 accumulation on the matrix cores, one rounding.  N and K multiples of 64; res may be NULL; out must not
 * alias a.  Same stream convention. */
int vsc_gemm_bias_act_bf16(const void* a, const void* w, const float* bias, const void* res, void* out,
                           int64_t M, int64_t N, int64_t K, int relu, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* VSCMI_H */
