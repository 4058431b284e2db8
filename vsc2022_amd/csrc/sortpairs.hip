// Final ordering of the kept hits and (query video, ref video) aggregation (gfx950).
//
//   sort_hits_topk : vsc/index.py:158-165 -- flatten in (row asc, ref asc) order, stable sort by
//                    score descending, truncate to K.  Done as two stable LSD radix sorts:
//                    by the 64-bit (row, ref) key, then by the 32-bit score key descending.
//   pair_max       : vsc/index.py:121-140 + vsc/candidates.py:24-40 -- group hits by
//                    (query video, ref video) in first-appearance order of the score-sorted list;
//                    because the list is score-descending the first hit of a pair carries its max,
//                    and first-appearance order IS the stable descending order of the pair scores.
//
// HBM-bound plumbing on <= 2K hits (12-16 B each) over the stable LSD radix sort of radix.h.
#include <cfloat>
#include <cstring>
#include <utility>

#include "kernels.h"
#include "radix.h"

namespace vscmi {

__global__ __launch_bounds__(256) void pack_hits_kernel(const int32_t* i, const int32_t* j,
                                                        const float* s, int64_t n, uint64_t* key64,
                                                        uint32_t* key32) {
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (x >= n) return;
    key64[x] = ((uint64_t)(uint32_t)i[x] << 32) | (uint32_t)j[x];
    key32[x] = f2key(s[x]);
}

__global__ __launch_bounds__(256) void unpack_hits_kernel(const uint64_t* key64, const uint32_t* key32,
                                                          int64_t n, int32_t* i, int32_t* j, float* s,
                                                          int negate) {
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (x >= n) return;
    i[x] = (int32_t)(key64[x] >> 32);
    j[x] = (int32_t)(key64[x] & 0xffffffffu);
    const float v = key2f(key32[x]);
    s[x] = negate ? -v : v;
}

static inline unsigned grid_for(int64_t n) { return (unsigned)((n + 255) / 256); }

static int bits_for(uint64_t maxval) {
    int b = 1;
    while (b < 64 && (maxval >> b)) ++b;
    return b;
}

// Orders hits (device arrays, arbitrary order) by (score desc, i asc, j asc) and writes the first
// min(n, K) to out_* (device).  `negate`: stored scores are negated distances (L2).
int sort_hits_topk(const int32_t* hi, const int32_t* hj, const float* hs, int64_t n, int64_t K,
                   int64_t max_i, int64_t max_j, DevBuf& w0, DevBuf& w1, DevBuf& w2, DevBuf& w3, DevBuf& tmp,
                   int32_t* out_i, int32_t* out_j, float* out_s, int negate, int64_t* n_out,
                   hipStream_t stream) {
    const int64_t m = n < K ? n : K;
    *n_out = m;
    if (n <= 0) return VSC_OK;
    VSC_TRY(w0.reserve(sizeof(uint64_t) * n));
    VSC_TRY(w1.reserve(sizeof(uint64_t) * n));
    VSC_TRY(w2.reserve(sizeof(uint32_t) * n));
    VSC_TRY(w3.reserve(sizeof(uint32_t) * n));
    uint64_t* k64a = w0.as<uint64_t>();
    uint64_t* k64b = w1.as<uint64_t>();
    uint32_t* k32a = w2.as<uint32_t>();
    uint32_t* k32b = w3.as<uint32_t>();
    hipLaunchKernelGGL(pack_hits_kernel, dim3(grid_for(n)), dim3(256), 0, stream, hi, hj, hs, n, k64a, k32a);
    VSC_HIP(hipGetLastError());
    const int end_bit = 32 + bits_for((uint64_t)(max_i > 0 ? max_i : 1));
    VSC_TRY(tmp.reserve(radix_tmp_bytes(n)));
    // (row, ref) ascending, then -- stably -- score descending
    // (the reference rows occupy `max_j` bits of the low word: the passes over the zero bits above them are skipped)
    int w = radix_sort_pairs<uint64_t, uint32_t>(k64a, k64b, k32a, k32b, n, 0, max_j > 0 ? bits_for((uint64_t)max_j) : 32, false,
                                                 tmp.p, stream, 32, end_bit);
    if (w < 0) { set_error("radix sort launch failed"); return VSC_ERR_HIP; }
    if (w) { std::swap(k64a, k64b); std::swap(k32a, k32b); }  // sorted pairs now in (k64a, k32a)
    w = radix_sort_pairs<uint32_t, uint64_t>(k32a, k32b, k64a, k64b, n, 0, 32, true, tmp.p, stream);
    if (w < 0) { set_error("radix sort launch failed"); return VSC_ERR_HIP; }
    if (w) { std::swap(k64a, k64b); std::swap(k32a, k32b); }
    hipLaunchKernelGGL(unpack_hits_kernel, dim3(grid_for(m)), dim3(256), 0, stream, k64a, k32a, m, out_i, out_j, out_s, negate);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

// ----------------------------------------------------------------------------- rows of a launch by threshold
//
// The int8 pre-filter tests a tile against the SMALLEST row threshold of its 128-row panel, and a 16-row block of a
// passing column block against the smallest of its own (round 3: 32-row blocks), before it looks at single
// accumulators; with an 8-bit error bound that gate only filters when the rows of a block have similar thresholds.
// Inside one launch the row order is free (candidates carry their row index), so the rows are handed to the kernel
// sorted by threshold: perm[position] = row.  All 32 key bits: the kernel tests a whole block against its smallest
// threshold, and at 3.5 sigma a threshold that is 0.04 sigma too low (what 16 key bits leave inside a block) already
// passes 15 % more candidates (measured: 1.02 -> 1.24 G candidates on the score-normalised config-4 search).
__global__ __launch_bounds__(256) void thr_keys_kernel(const float* __restrict__ thr, int n, uint32_t* __restrict__ keys,
                                                       int32_t* __restrict__ vals) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= n) return;
    keys[x] = f2key(thr[x]);
    vals[x] = x;
}

int sort_rows_by_threshold(const float* thr, int64_t n, DevBuf& w0, DevBuf& w1, DevBuf& w2, DevBuf& w3, DevBuf& tmp,
                           const int32_t** perm, hipStream_t stream) {
    *perm = nullptr;
    if (n <= 0) return VSC_OK;
    VSC_TRY(w0.reserve(sizeof(uint32_t) * n));
    VSC_TRY(w1.reserve(sizeof(uint32_t) * n));
    VSC_TRY(w2.reserve(sizeof(int32_t) * n));
    VSC_TRY(w3.reserve(sizeof(int32_t) * n));
    VSC_TRY(tmp.reserve(radix_tmp_bytes(n)));
    uint32_t *ka = w0.as<uint32_t>(), *kb = w1.as<uint32_t>();
    int32_t *va = w2.as<int32_t>(), *vb = w3.as<int32_t>();
    hipLaunchKernelGGL(thr_keys_kernel, dim3(grid_for(n)), dim3(256), 0, stream, thr, (int)n, ka, va);
    VSC_HIP(hipGetLastError());
    const int w = radix_sort_pairs<uint32_t, int32_t>(ka, kb, va, vb, n, 0, 32, false, tmp.p, stream);
    if (w < 0) { set_error("radix sort launch failed"); return VSC_ERR_HIP; }
    *perm = w ? vb : va;
    return VSC_OK;
}

// Stable argsort of a score list, best first (vsc_argsort_scores: the merge of the ranks' candidate lists, vsc2022_amd/dist.py
// merge_candidates -- equal scores keep their input order, which there is rank order = first-appearance order).
__global__ __launch_bounds__(256) void score_keys_kernel(const float* __restrict__ s, long long n, uint32_t* __restrict__ keys,
                                                         int32_t* __restrict__ vals) {
    const long long x = (long long)blockIdx.x * 256 + threadIdx.x;
    if (x >= n) return;
    keys[x] = f2key(s[x] + 0.0f);   // (-0.0 == +0.0, as in the reference's float comparisons)
    vals[x] = (int32_t)x;
}

int argsort_scores_desc(const float* s, int64_t n, DevBuf& w0, DevBuf& w1, DevBuf& w2, DevBuf& w3, DevBuf& tmp,
                        const int32_t** perm, hipStream_t stream) {
    *perm = nullptr;
    if (n <= 0) return VSC_OK;
    VSC_TRY(w0.reserve(sizeof(uint32_t) * n));
    VSC_TRY(w1.reserve(sizeof(uint32_t) * n));
    VSC_TRY(w2.reserve(sizeof(int32_t) * n));
    VSC_TRY(w3.reserve(sizeof(int32_t) * n));
    VSC_TRY(tmp.reserve(radix_tmp_bytes(n)));
    uint32_t *ka = w0.as<uint32_t>(), *kb = w1.as<uint32_t>();
    int32_t *va = w2.as<int32_t>(), *vb = w3.as<int32_t>();
    hipLaunchKernelGGL(score_keys_kernel, dim3(grid_for(n)), dim3(256), 0, stream, s, (long long)n, ka, va);
    VSC_HIP(hipGetLastError());
    const int w = radix_sort_pairs<uint32_t, int32_t>(ka, kb, va, vb, n, 0, 32, true, tmp.p, stream);
    if (w < 0) { set_error("radix sort launch failed"); return VSC_ERR_HIP; }
    *perm = w ? vb : va;
    return VSC_OK;
}

// ... and, inside groups of `group` consecutive positions of that order, by the rows' scale (their largest |x|): the
// panels of the int8 kernel share ONE quantisation scale per 128 rows, so a panel of rows that would have picked
// nearly that scale themselves keeps the panel's error bound E_q near a single row's (the radius search sorts all its
// rows that way); with per-row thresholds the order by threshold comes first -- a block is gated by its smallest
// threshold -- but inside a group of 512 rows of a 32768-row launch the thresholds differ by a few percent of their
// spread, less than what the tighter scale buys (round 4: -6.5 % candidates on configs[3]'s search; api.hip).
__global__ __launch_bounds__(256) void group_scale_keys_kernel(const int32_t* __restrict__ perm, const float* __restrict__ scale,
                                                               int n, int group_shift, uint32_t* __restrict__ keys) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    // 12 bits of group number | the top 20 bits of the order-preserving key of a non-negative float
    keys[p] = ((uint32_t)(p >> group_shift) << 20) | (f2key(scale[perm[p]]) >> 12);
}

// The same order in ONE launch when a group fits a workgroup (<= 1024 positions; the default is 512): the positions are
// already grouped, so the "sort by (group, scale)" is a stable sort by scale INSIDE every group -- rank counting in LDS,
// one workgroup per group: rank(p) = #{p' of the group: key(p') < key(p), or equal and p' < p}.  Same keys (the top 20
// bits of the scale's order-preserving image), same stable order as the four radix passes it replaces -- which were 16
// launches of ~5 us kernels per batch and rank of the column-sharded schedule (profiles/r06_rank_work.md).
__global__ __launch_bounds__(1024) void group_scale_rank_kernel(const int32_t* __restrict__ perm, const float* __restrict__ scale, int n,
                                                                int group_shift, int32_t* __restrict__ out) {
    __shared__ uint32_t key[1024];
    const int G = 1 << group_shift;
    const int base = blockIdx.x << group_shift;
    const int cnt = min(G, n - base);
    const int t = threadIdx.x;
    int32_t row = 0;
    uint32_t mine = 0;
    if (t < cnt) {
        row = perm[base + t];
        mine = f2key(scale[row]) >> 12;
        key[t] = mine;
    }
    __syncthreads();
    if (t >= cnt) return;
    int rank = 0;
    for (int f = 0; f < cnt; ++f) {
        const uint32_t kf = key[f];
        rank += (kf < mine) || (kf == mine && f < t);
    }
    out[base + rank] = row;
}

int sort_rows_by_threshold_then_scale(const float* thr, const float* scale, int64_t n, int group_shift, DevBuf& w0, DevBuf& w1,
                                      DevBuf& w2, DevBuf& w3, DevBuf& tmp, const int32_t** perm, hipStream_t stream) {
    VSC_TRY(sort_rows_by_threshold(thr, n, w0, w1, w2, w3, tmp, perm, stream));
    if (n <= 0 || (n >> group_shift) >= 4096) return VSC_OK;  // (12 bits of group number)
    if (group_shift <= 10) {
        int32_t* va = const_cast<int32_t*>(*perm);
        int32_t* vb = va == w2.as<int32_t>() ? w3.as<int32_t>() : w2.as<int32_t>();
        const int G = 1 << group_shift;
        hipLaunchKernelGGL(group_scale_rank_kernel, dim3((unsigned)((n + G - 1) >> group_shift)), dim3(std::max(64, G)), 0, stream,
                           va, scale, (int)n, group_shift, vb);
        VSC_HIP(hipGetLastError());
        *perm = vb;
        return VSC_OK;
    }
    uint32_t *ka = w0.as<uint32_t>(), *kb = w1.as<uint32_t>();
    int32_t* va = const_cast<int32_t*>(*perm);
    int32_t* vb = va == w2.as<int32_t>() ? w3.as<int32_t>() : w2.as<int32_t>();
    hipLaunchKernelGGL(group_scale_keys_kernel, dim3(grid_for(n)), dim3(256), 0, stream, va, scale, (int)n, group_shift, ka);
    VSC_HIP(hipGetLastError());
    const int w = radix_sort_pairs<uint32_t, int32_t>(ka, kb, va, vb, n, 0, 32, false, tmp.p, stream);
    if (w < 0) { set_error("radix sort launch failed"); return VSC_ERR_HIP; }
    *perm = w ? vb : va;
    return VSC_OK;
}

// candidates (reference row, query row) of one launch ordered by reference row: keys need only the bits of nrefs
int sort_candidates_by_ref(uint32_t* ka, uint32_t* kb, uint32_t* va, uint32_t* vb, int64_t n, int64_t nrefs, DevBuf& tmp,
                           const uint32_t** sorted_j, const uint32_t** sorted_i, hipStream_t stream) {
    *sorted_j = ka;
    *sorted_i = va;
    if (n <= 0) return VSC_OK;
    VSC_TRY(tmp.reserve(radix_tmp_bytes(n)));
    const int w = radix_sort_pairs<uint32_t, uint32_t>(ka, kb, va, vb, n, 0, bits_for((uint64_t)std::max<int64_t>(nrefs, 2) - 1),
                                                       false, tmp.p, stream);
    if (w < 0) { set_error("radix sort launch failed"); return VSC_ERR_HIP; }
    *sorted_j = w ? kb : ka;
    *sorted_i = w ? vb : va;
    return VSC_OK;
}

// ----------------------------------------------------------------------------- pair max

// maximum of the video ordinals of the rows, taken as the unsigned words the pair key holds: how many key bits the
// pair sort has to look at
__global__ __launch_bounds__(256) void max_u32_kernel(const int32_t* x, int64_t n, unsigned* out) {
    unsigned m = 0;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) m = max(m, (unsigned)x[e]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(out, m);
}

__global__ __launch_bounds__(256) void pair_key_kernel(const int32_t* hi, const int32_t* hj, int64_t n,
                                                       const int32_t* row2q, const int32_t* row2r,
                                                       uint64_t* key, uint32_t* rank) {
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (x >= n) return;
    key[x] = ((uint64_t)(uint32_t)row2q[hi[x]] << 32) | (uint32_t)row2r[hj[x]];
    rank[x] = (uint32_t)x;
}

__global__ __launch_bounds__(256) void pair_heads_kernel(const uint64_t* key, const uint32_t* rank,
                                                         int64_t n, uint32_t* head_rank,
                                                         uint64_t* head_key, unsigned long long* count) {
    // one atomic per workgroup on the (single) counter, not one per wavefront
    __shared__ unsigned int wave_cnt[4];
    __shared__ unsigned long long block_base;
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool head = (x < n) && (x == 0 || key[x] != key[x - 1]);
    const unsigned long long m = __ballot(head);
    if (lane == 0) wave_cnt[wave] = (unsigned int)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int tot = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        block_base = tot ? atomicAdd(count, (unsigned long long)tot) : 0ull;
    }
    __syncthreads();
    if (head) {
        unsigned long long pos = block_base + __popcll(m & ((1ull << lane) - 1));
        for (int w = 0; w < wave; ++w) pos += wave_cnt[w];
        head_rank[pos] = rank[x];  // stable sort => smallest rank of the run == first appearance
        head_key[pos] = key[x];
    }
}

__global__ __launch_bounds__(256) void pair_out_kernel(const uint32_t* head_rank, const uint64_t* head_key,
                                                       const float* hs, int64_t np, int32_t* out_q,
                                                       int32_t* out_r, float* out_s, int64_t* out_first) {
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (x >= np) return;
    out_q[x] = (int32_t)(head_key[x] >> 32);
    out_r[x] = (int32_t)(head_key[x] & 0xffffffffu);
    out_s[x] = hs[head_rank[x]];
    if (out_first) out_first[x] = (int64_t)head_rank[x];
}

// All pointers device.  hits are in search order (score-descending).  Synchronises the stream once
// (the pair count sizes the second sort).
int pair_max_device(const int32_t* hi, const int32_t* hj, const float* hs, int64_t n,
                    const int32_t* row2q, const int32_t* row2r, int64_t nq_rows, int64_t nr_rows, DevBuf& w0,
                    DevBuf& w1, DevBuf& w2, DevBuf& w3, DevBuf& tmp, DevBuf& cnt, int32_t* out_q,
                    int32_t* out_r, float* out_s, int64_t* out_first, int64_t cap, int64_t* n_pairs,
                    hipStream_t stream) {
    *n_pairs = 0;
    if (n <= 0) return VSC_OK;
    if (n > 0xffffffffLL) {
        set_error("pair_max: more than 2^32 hits");
        return VSC_ERR_INVALID;
    }
    VSC_TRY(w0.reserve(sizeof(uint64_t) * n));
    VSC_TRY(w1.reserve(sizeof(uint64_t) * n));
    VSC_TRY(w2.reserve(sizeof(uint32_t) * n));
    VSC_TRY(w3.reserve(sizeof(uint32_t) * n));
    VSC_TRY(cnt.reserve(sizeof(unsigned long long)));
    uint64_t* ka = w0.as<uint64_t>();
    uint64_t* kb = w1.as<uint64_t>();
    uint32_t* ra = w2.as<uint32_t>();
    uint32_t* rb = w3.as<uint32_t>();
    hipLaunchKernelGGL(pair_key_kernel, dim3(grid_for(n)), dim3(256), 0, stream, hi, hj, n, row2q, row2r, ka, ra);
    VSC_HIP(hipGetLastError());
    VSC_TRY(tmp.reserve(radix_tmp_bytes(n)));
    // the key is (query video << 32 | ref video): only the bits the largest ordinals occupy are sorted (40 000 videos a
    // side: 2 + 2 passes instead of 8)
    VSC_TRY(cnt.reserve(sizeof(unsigned long long)));
    VSC_HIP(hipMemsetAsync(cnt.p, 0, sizeof(unsigned long long), stream));
    unsigned* mx = reinterpret_cast<unsigned*>(cnt.p);
    hipLaunchKernelGGL(max_u32_kernel, dim3(256), dim3(256), 0, stream, row2q, nq_rows, mx);
    hipLaunchKernelGGL(max_u32_kernel, dim3(256), dim3(256), 0, stream, row2r, nr_rows, mx + 1);
    VSC_HIP(hipGetLastError());
    unsigned mxh[2] = {0, 0};
    VSC_HIP(hipMemcpyAsync(mxh, mx, sizeof(mxh), hipMemcpyDeviceToHost, stream));
    VSC_HIP(hipStreamSynchronize(stream));
    const int bits_q = bits_for((uint64_t)std::max(mxh[0], 1u)), bits_r = bits_for((uint64_t)std::max(mxh[1], 1u));
    int w = radix_sort_pairs<uint64_t, uint32_t>(ka, kb, ra, rb, n, 0, bits_r, false, tmp.p, stream, 32, 32 + bits_q);
    if (w < 0) { set_error("radix sort launch failed"); return VSC_ERR_HIP; }
    if (!w) { std::swap(ka, kb); std::swap(ra, rb); }  // sorted pairs in (kb, rb); (ka, ra) are free
    VSC_HIP(hipMemsetAsync(cnt.p, 0, sizeof(unsigned long long), stream));
    // heads compacted into (ra, ka)
    hipLaunchKernelGGL(pair_heads_kernel, dim3(grid_for(n)), dim3(256), 0, stream, kb, rb, n, ra, ka,
                       cnt.as<unsigned long long>());
    VSC_HIP(hipGetLastError());
    unsigned long long np = 0;
    VSC_HIP(hipMemcpyAsync(&np, cnt.p, sizeof(np), hipMemcpyDeviceToHost, stream));
    VSC_HIP(hipStreamSynchronize(stream));
    *n_pairs = (int64_t)np;
    if ((int64_t)np > cap) {
        set_error("pair_max: output capacity %lld < %llu pairs", (long long)cap, np);
        return VSC_ERR_CAPACITY;
    }
    // first-appearance order = ascending rank of the heads (4 passes: the result is back in (ra, ka))
    w = radix_sort_pairs<uint32_t, uint64_t>(ra, rb, ka, kb, (int64_t)np, 0, 32, false, tmp.p, stream);
    if (w < 0) { set_error("radix sort launch failed"); return VSC_ERR_HIP; }
    if (w) { std::swap(ka, kb); std::swap(ra, rb); }
    hipLaunchKernelGGL(pair_out_kernel, dim3(grid_for((int64_t)np)), dim3(256), 0, stream, ra, ka, hs,
                       (int64_t)np, out_q, out_r, out_s, out_first);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

// ----------------------------------------------------------------------------- k-NN from a hit list
//
// Second half of the pre-filtered k-NN (api.hip, vsc_index_knn): the exact stage delivered, in any order,
// every (row, ref, score) with score >= the row's threshold -- at least k per row.  Order them by
// (row asc, score desc, ref asc) with two stable radix sorts and cut each row at k.

__global__ __launch_bounds__(256) void knn_keys_kernel(const int32_t* hi, const float* hs, int64_t n,
                                                       uint64_t* key64) {
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (x >= n) return;
    key64[x] = ((uint64_t)(uint32_t)hi[x] << 32) | (uint32_t)~f2key(hs[x]);  // ascending = score descending
}

__global__ __launch_bounds__(256) void knn_cut_kernel(const uint64_t* key64, const uint32_t* ref, int64_t n,
                                                      int64_t nq, int k, float* out_s, int64_t* out_j) {
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (x >= nq * k) return;
    const int64_t row = x / k;
    const int slot = (int)(x % k);
    // first element of the row: lower bound of (row << 32)
    const uint64_t want = (uint64_t)row << 32;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (key64[mid] < want) lo = mid + 1;
        else hi = mid;
    }
    const int64_t e = lo + slot;
    const bool have = e < n && (int64_t)(key64[e] >> 32) == row;
    out_s[x] = have ? key2f(~(uint32_t)(key64[e] & 0xffffffffu)) : -FLT_MAX;
    out_j[x] = have ? (int64_t)ref[e] : -1;
}

int knn_from_hits(const int32_t* hi, const int32_t* hj, const float* hs, int64_t n, int64_t nq, int k, DevBuf& w0,
                  DevBuf& w1, DevBuf& w2, DevBuf& w3, DevBuf& tmp, float* out_s, int64_t* out_j,
                  hipStream_t stream) {
    const int64_t nn = n > 0 ? n : 1;
    VSC_TRY(w0.reserve(sizeof(uint64_t) * nn));
    VSC_TRY(w1.reserve(sizeof(uint64_t) * nn));
    VSC_TRY(w2.reserve(sizeof(uint32_t) * nn));
    VSC_TRY(w3.reserve(sizeof(uint32_t) * nn));
    uint64_t* ka = w0.as<uint64_t>();
    uint64_t* kb = w1.as<uint64_t>();
    uint32_t* ra = w2.as<uint32_t>();
    uint32_t* rb = w3.as<uint32_t>();
    if (n > 0) {
        hipLaunchKernelGGL(knn_keys_kernel, dim3(grid_for(n)), dim3(256), 0, stream, hi, hs, n, ka);
        VSC_HIP(hipGetLastError());
        // the sort ping-pongs between its buffers: work on a copy of the refs, the hit list stays intact
        VSC_HIP(hipMemcpyAsync(ra, hj, (size_t)n * 4, hipMemcpyDeviceToDevice, stream));
        const int end_bit = 32 + bits_for((uint64_t)(nq > 0 ? nq : 1));
        VSC_TRY(tmp.reserve(radix_tmp_bytes(n)));
        // 1. refs ascending (payload: the row/score key) ...
        int w = radix_sort_pairs<uint32_t, uint64_t>(ra, rb, ka, kb, n, 0, 32, false, tmp.p, stream);
        if (w < 0) { set_error("radix sort launch failed"); return VSC_ERR_HIP; }
        if (w) { std::swap(ka, kb); std::swap(ra, rb); }
        // 2. ... then, stably, row ascending / score descending
        w = radix_sort_pairs<uint64_t, uint32_t>(ka, kb, ra, rb, n, 0, end_bit, false, tmp.p, stream);
        if (w < 0) { set_error("radix sort launch failed"); return VSC_ERR_HIP; }
        if (w) { std::swap(ka, kb); std::swap(ra, rb); }
    }
    hipLaunchKernelGGL(knn_cut_kernel, dim3(grid_for(nq * k)), dim3(256), 0, stream, ka, ra, n, nq, k, out_s, out_j);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

// The k-NN lists of the reference rows searched so far re-enter the hit list as (row, ref, score) triples, so that the
// next range of references is merged with them by the same sort + cut (knn_from_hits); empty slots (ref -1) are skipped.
__global__ __launch_bounds__(256) void knn_seed_hits_kernel(const float* __restrict__ knn_s, const int64_t* __restrict__ knn_j,
                                                            int64_t n, int k, int32_t* __restrict__ hi, int32_t* __restrict__ hj,
                                                            float* __restrict__ hs, unsigned long long* __restrict__ counter) {
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool have = x < n && knn_j[x] >= 0;
    const unsigned long long m = __ballot(have);
    if (m == 0ull) return;
    const int lane = threadIdx.x & 63;
    unsigned long long base = 0;
    if (lane == __ffsll((long long)m) - 1) base = atomicAdd(counter, (unsigned long long)__popcll(m));
    base = __shfl(base, __ffsll((long long)m) - 1);
    if (have) {
        const unsigned long long p = base + __popcll(m & ((1ull << lane) - 1));
        hi[p] = (int32_t)(x / k);
        hj[p] = (int32_t)knn_j[x];
        hs[p] = knn_s[x];
    }
}

int launch_knn_seed_hits(const float* knn_s, const int64_t* knn_j, int64_t nq, int k, int32_t* hi, int32_t* hj, float* hs,
                         unsigned long long* counter, hipStream_t stream) {
    const int64_t n = nq * k;
    if (n <= 0) return VSC_OK;
    hipLaunchKernelGGL(knn_seed_hits_kernel, dim3(grid_for(n)), dim3(256), 0, stream, knn_s, knn_j, n, k, hi, hj, hs, counter);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

// row_thr[i] = k-th best score of row i (out of a k-NN result), +inf for the padding rows
__global__ __launch_bounds__(256) void knn_row_thr_kernel(const float* knn_s, int64_t nq, int k, float* row_thr,
                                                          int64_t rows) {
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (x >= rows) return;
    row_thr[x] = x < nq ? knn_s[x * k + (k - 1)] : INFINITY;
}

int launch_knn_row_thr(const float* knn_s, int64_t nq, int k, float* row_thr, int64_t rows, hipStream_t stream) {
    if (rows <= 0) return VSC_OK;
    hipLaunchKernelGGL(knn_row_thr_kernel, dim3(grid_for(rows)), dim3(256), 0, stream, knn_s, nq, k, row_thr, rows);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

// faiss range_search order: rows ascending, refs ascending within a row (device in/out).
int sort_hits_rowcol(const int32_t* hi, const int32_t* hj, const float* hs, int64_t n, DevBuf& w0,
                     DevBuf& w1, DevBuf& w2, DevBuf& w3, DevBuf& tmp, int32_t* out_i, int32_t* out_j,
                     float* out_s, int negate, hipStream_t stream) {
    if (n <= 0) return VSC_OK;
    VSC_TRY(w0.reserve(sizeof(uint64_t) * n));
    VSC_TRY(w1.reserve(sizeof(uint64_t) * n));
    VSC_TRY(w2.reserve(sizeof(uint32_t) * n));
    VSC_TRY(w3.reserve(sizeof(uint32_t) * n));
    uint64_t* k64a = w0.as<uint64_t>();
    uint64_t* k64b = w1.as<uint64_t>();
    uint32_t* k32a = w2.as<uint32_t>();
    uint32_t* k32b = w3.as<uint32_t>();
    hipLaunchKernelGGL(pack_hits_kernel, dim3(grid_for(n)), dim3(256), 0, stream, hi, hj, hs, n, k64a, k32a);
    VSC_HIP(hipGetLastError());
    VSC_TRY(tmp.reserve(radix_tmp_bytes(n)));
    const int w = radix_sort_pairs<uint64_t, uint32_t>(k64a, k64b, k32a, k32b, n, 0, 64, false, tmp.p, stream);
    if (w < 0) { set_error("radix sort launch failed"); return VSC_ERR_HIP; }
    if (w) { std::swap(k64a, k64b); std::swap(k32a, k32b); }
    hipLaunchKernelGGL(unpack_hits_kernel, dim3(grid_for(n)), dim3(256), 0, stream, k64a, k32a, n, out_i, out_j, out_s, negate);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

}  // namespace vscmi
