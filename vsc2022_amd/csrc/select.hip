// Device-side re-thresholding of the kept hit list (gfx950).
//
// Restates faiss.contrib.exhaustive_search.apply_maxres as called from
// range_search_max_results (reached at vsc/index.py:147-154): when more than max_results = 2K hits
// are kept, the radius becomes the (K+1)-th best kept score and every kept hit is re-filtered with
// a STRICT comparison.  Everything is stream-ordered and predicated on device memory, so the whole
// batch schedule of a search is enqueued without a host round trip.
//
// HBM-bound: one radix-select = 4 passes over <= ~2K + batch scores (4 B each); one compaction
// pass over 12 B/hit.  Negligible next to the similarity kernel; kept simple.
#include <cfloat>
#include "kernels.h"

namespace vscmi {


__global__ void select_begin_kernel(SelectCtl* c, unsigned long long max_results,
                                    unsigned long long min_results) {
    if (threadIdx.x < 256) c->hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        // After an overflow c->n counts hits that were never stored (it may exceed the buffer):
        // the host reruns the whole schedule with a larger buffer, so every later round is a no-op
        // instead of walking past the end of the hit arrays.
        const bool act = !c->overflow && c->n > max_results;
        c->active = act ? 1 : 0;
        c->prefix = 0;
        c->prefix_mask = 0;
        c->rank = min_results;  // 0-based descending rank of the (K+1)-th best
        c->n_tmp = 0;
        if (act) c->n_rethreshold += 1;
    }
}

// One digit histogram of the kept scores that match the prefix found so far.  Round 5: 16-byte loads, and the FIRST pass
// (shift 24: every kept score of a search shares its sign and exponent byte, so 64 lanes used to queue on one LDS counter)
// counts equal digits inside the wave with ballots and adds once per distinct digit; the later passes see digits that are
// spread (or few matching elements) and keep the plain LDS atomics.
__global__ __launch_bounds__(256) void select_hist_kernel(SelectCtl* c, const float* s, int shift) {
    if (!c->active) return;
    __shared__ unsigned int lh[256];
    lh[threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long n = c->n;
    const unsigned int prefix = c->prefix, pmask = c->prefix_mask;
    const int lane = threadIdx.x & 63;
    const unsigned long long n4 = (n + 3) / 4;  // (the buffer is 16-byte aligned and padded: see ensure_hit_buffers)
    for (unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x; q < ((n4 + 255) & ~255ull);
         q += (unsigned long long)gridDim.x * 256) {
        float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (q < n4) {
            const float4 t = *reinterpret_cast<const float4*>(s + q * 4);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned int key = f2key(v[e]);
            const bool valid = q * 4 + e < n && (key & pmask) == prefix;
            const unsigned int bin = (key >> shift) & 255u;
            if (shift == 24) {
                unsigned long long todo = __ballot(valid);
                while (todo) {
                    const int leader = __ffsll((long long)todo) - 1;
                    const unsigned int lb = __shfl(bin, leader);
                    const unsigned long long same = __ballot(valid && bin == lb);
                    if (lane == leader) atomicAdd(&lh[lb], (unsigned int)__popcll(same));
                    todo &= ~same;
                }
            } else if (valid) {
                atomicAdd(&lh[bin], 1u);
            }
        }
    }
    __syncthreads();
    if (lh[threadIdx.x]) atomicAdd(&c->hist[threadIdx.x], lh[threadIdx.x]);
}

__global__ void select_pick_kernel(SelectCtl* c, int shift) {
    if (!c->active) return;
    __shared__ unsigned int h[256];
    h[threadIdx.x] = c->hist[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long rank = c->rank;
        int d = 255;
        for (; d > 0; --d) {
            if (rank < h[d]) break;
            rank -= h[d];
        }
        c->rank = rank;
        c->prefix |= (unsigned int)d << shift;
        c->prefix_mask |= 255u << shift;
        if (shift == 0) c->radius = key2f(c->prefix);
    }
    __syncthreads();
    c->hist[threadIdx.x] = 0;
}

// copy hits with score > radius from A to B
__global__ __launch_bounds__(256) void select_compact_kernel(SelectCtl* c, const int32_t* ai,
                                                             const int32_t* aj, const float* as,
                                                             int32_t* bi, int32_t* bj, float* bs) {
    if (!c->active) return;
    // One global atomic per 2048-element chunk (not per wavefront): with tens of millions of kept
    // hits the single counter would otherwise serialise the whole pass.
    constexpr int PER_THREAD = 8, CHUNK = 256 * PER_THREAD;
    __shared__ unsigned int wave_cnt[4];
    __shared__ unsigned long long chunk_base;
    const unsigned long long n = c->n;
    const float radius = c->radius;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long n_chunks = (n + CHUNK - 1) / CHUNK;
    for (unsigned long long ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
        const unsigned long long x0 = ch * CHUNK + threadIdx.x;
        float s[PER_THREAD];
        unsigned int keep_bits = 0;
#pragma unroll
        for (int e = 0; e < PER_THREAD; ++e) {
            const unsigned long long x = x0 + (unsigned long long)e * 256;
            s[e] = x < n ? as[x] : 0.0f;
            keep_bits |= (unsigned int)((x < n) && (s[e] > radius)) << e;
        }
        // order inside the chunk: wave, element slot, lane -- any order is fine (sorted later).  The survivors of one
        // element slot of a wave go to CONSECUTIVE positions (ballot ranks), so the three stores of a slot are
        // contiguous runs (round 5; a thread's survivors used to sit next to each other, the wave's stores strided)
        unsigned long long bal[PER_THREAD];
        unsigned int wave_total = 0;
#pragma unroll
        for (int e = 0; e < PER_THREAD; ++e) {
            bal[e] = __ballot((keep_bits >> e) & 1u);
            wave_total += (unsigned int)__popcll(bal[e]);
        }
        if (lane == 0) wave_cnt[wave] = wave_total;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned int tot = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
            chunk_base = tot ? atomicAdd(&c->n_tmp, (unsigned long long)tot) : 0ull;
        }
        __syncthreads();
        unsigned long long pos = chunk_base;
        for (int w = 0; w < wave; ++w) pos += wave_cnt[w];
        const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
        for (int e = 0; e < PER_THREAD; ++e) {
            if ((keep_bits >> e) & 1u) {
                const unsigned long long x = x0 + (unsigned long long)e * 256;
                const unsigned long long p = pos + (unsigned int)__popcll(bal[e] & below);
                bi[p] = ai[x];
                bj[p] = aj[x];
                bs[p] = s[e];
            }
            pos += (unsigned int)__popcll(bal[e]);
        }
        __syncthreads();  // wave_cnt / chunk_base are reused by the next chunk
    }
}

__global__ __launch_bounds__(256) void select_copyback_kernel(SelectCtl* c, const int32_t* bi,
                                                              const int32_t* bj, const float* bs,
                                                              int32_t* ai, int32_t* aj, float* as) {
    if (!c->active) return;
    const unsigned long long n = c->n_tmp;
    for (unsigned long long x = (unsigned long long)blockIdx.x * 256 + threadIdx.x; x < n;
         x += (unsigned long long)gridDim.x * 256) {
        ai[x] = bi[x];
        aj[x] = bj[x];
        as[x] = bs[x];
    }
}

__global__ void select_end_kernel(SelectCtl* c) {
    if (c->active) c->n = c->n_tmp;
}

// A fresh control block for a search: everything zero, the radius set (stream-ordered: no host staging, no sync).
__global__ void ctl_init_kernel(SelectCtl* c, float radius) {
    unsigned int* w = reinterpret_cast<unsigned int*>(c);
    for (unsigned x = threadIdx.x; x < sizeof(SelectCtl) / 4; x += blockDim.x) w[x] = 0u;
    __syncthreads();
    if (threadIdx.x == 0) c->radius = radius;
}
int launch_ctl_init(SelectCtl* ctl, float radius, hipStream_t stream) {
    static_assert(sizeof(SelectCtl) % 4 == 0, "SelectCtl is cleared word by word");
    hipLaunchKernelGGL(ctl_init_kernel, dim3(1), dim3(256), 0, stream, ctl, radius);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

// Enqueue one "if n > 2K: radius <- (K+1)-th best; keep score > radius" round.
int enqueue_rethreshold(SelectCtl* ctl, int32_t* ai, int32_t* aj, float* as, int32_t* bi, int32_t* bj,
                        float* bs, unsigned long long K, hipStream_t stream) {
    constexpr int GRID = 1024;
    hipLaunchKernelGGL(select_begin_kernel, dim3(1), dim3(256), 0, stream, ctl, 2 * K, K);
    for (int shift = 24; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL(select_hist_kernel, dim3(GRID), dim3(256), 0, stream, ctl, as, shift);
        hipLaunchKernelGGL(select_pick_kernel, dim3(1), dim3(256), 0, stream, ctl, shift);
    }
    hipLaunchKernelGGL(select_compact_kernel, dim3(GRID), dim3(256), 0, stream, ctl, ai, aj, as, bi, bj, bs);
    hipLaunchKernelGGL(select_copyback_kernel, dim3(GRID), dim3(256), 0, stream, ctl, bi, bj, bs, ai, aj, as);
    hipLaunchKernelGGL(select_end_kernel, dim3(1), dim3(1), 0, stream, ctl);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Order statistics over UNSORTED score lists that are spread over ranks (vsc_score_histogram / vsc_score_pick; the
// sharded schedule's re-threshold events and the reference-sharded selections, vsc2022_amd/dist.py): the same 4 x 8-bit
// radix select as above, but with the histogram of every level handed OUT -- the caller adds the ranks' histograms (one
// all-reduce of 256 counters) and hands the sum back to `score_pick_kernel`, which narrows the key prefix.  The state
// stays in device memory (int64[4]: key prefix, prefix mask, 1-based rank still wanted among the matching keys, scores
// strictly above the prefix so far), so a whole selection runs without a host round trip.
// Keys: f2key(score + 0.0f) -- -0.0 and +0.0 share a key, as they compare equal in the reference's float comparisons.
__global__ __launch_bounds__(256) void score_hist_kernel(const float* __restrict__ s, long long n, const long long* __restrict__ state,
                                                         int shift, unsigned long long* __restrict__ hist) {
    __shared__ unsigned int lh[256];
    lh[threadIdx.x] = 0;
    __syncthreads();
    const unsigned int prefix = (unsigned int)state[0], pmask = (unsigned int)state[1];
    const int lane = threadIdx.x & 63;
    // (16-byte loads where the list allows them: the head up to the first 16-byte boundary and the tail run scalar)
    const long long head = min(n, (long long)(((16 - ((uintptr_t)s & 15)) & 15) >> 2));
    const long long n4 = (n - head) / 4;
    const float4* s4 = reinterpret_cast<const float4*>(s + head);
    const long long rounds = (n4 + 255) & ~255ll;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < rounds; q += (long long)gridDim.x * 256) {
        float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        const bool in = q < n4;
        if (in) {
            const float4 t = s4[q];
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned int key = f2key(v[e] + 0.0f);
            const bool valid = in && (key & pmask) == prefix;
            const unsigned int bin = (key >> shift) & 255u;
            if (shift == 24) {  // (a kept list shares sign and exponent: count equal digits inside the wave first)
                unsigned long long todo = __ballot(valid);
                while (todo) {
                    const int leader = __ffsll((long long)todo) - 1;
                    const unsigned int lb = __shfl(bin, leader);
                    const unsigned long long same = __ballot(valid && bin == lb);
                    if (lane == leader) atomicAdd(&lh[lb], (unsigned int)__popcll(same));
                    todo &= ~same;
                }
            } else if (valid) {
                atomicAdd(&lh[bin], 1u);
            }
        }
    }
    if (blockIdx.x == 0) {  // the unaligned head and the tail (< 4 + 3 scores)
        const long long tail0 = head + n4 * 4;
        const long long extra = head + (n - tail0);
        if ((long long)threadIdx.x < extra) {
            const long long x = (long long)threadIdx.x < head ? threadIdx.x : tail0 + (threadIdx.x - head);
            const unsigned int key = f2key(s[x] + 0.0f);
            if ((key & pmask) == prefix) atomicAdd(&lh[(key >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    if (lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)lh[threadIdx.x]);
}

__global__ void score_pick_kernel(const long long* __restrict__ hist, long long* __restrict__ state, int shift) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    long long need = state[2], above = state[3];
    int d = 255;
    for (; d > 0; --d) {
        if (need <= hist[d]) break;
        need -= hist[d];
        above += hist[d];
    }
    // (fewer than `need` scores in the union: d ends at 0 and the caller reads total < k from the first level's sum)
    state[0] = (long long)((unsigned int)state[0] | ((unsigned int)d << shift));
    state[1] = (long long)((unsigned int)state[1] | (255u << shift));
    state[2] = need;
    state[3] = above;
}

// Stand-alone form of the compaction above for hit lists a caller keeps itself (vsc_filter_hits; the sharded schedule's events,
// vsc2022_amd/dist.py: "everything at or below the new radius goes"): (i, j, s) with s > radius (STRICT) from A to B, any order.
__global__ __launch_bounds__(256) void filter_hits_kernel(const int32_t* __restrict__ ai, const int32_t* __restrict__ aj,
                                                          const float* __restrict__ as, long long n, float radius,
                                                          int32_t* __restrict__ bi, int32_t* __restrict__ bj, float* __restrict__ bs,
                                                          unsigned long long* __restrict__ n_out) {
    constexpr int PER_THREAD = 8, CHUNK = 256 * PER_THREAD;
    __shared__ unsigned int wave_cnt[4];
    __shared__ unsigned long long chunk_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long n_chunks = (n + CHUNK - 1) / CHUNK;
    for (long long ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
        const long long x0 = ch * CHUNK + threadIdx.x;
        float s[PER_THREAD];
        unsigned int keep_bits = 0;
#pragma unroll
        for (int e = 0; e < PER_THREAD; ++e) {
            const long long x = x0 + (long long)e * 256;
            s[e] = x < n ? as[x] : 0.0f;
            keep_bits |= (unsigned int)((x < n) && (s[e] > radius)) << e;
        }
        unsigned long long bal[PER_THREAD];
        unsigned int wave_total = 0;
#pragma unroll
        for (int e = 0; e < PER_THREAD; ++e) {
            bal[e] = __ballot((keep_bits >> e) & 1u);
            wave_total += (unsigned int)__popcll(bal[e]);
        }
        if (lane == 0) wave_cnt[wave] = wave_total;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned int tot = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
            chunk_base = tot ? atomicAdd(n_out, (unsigned long long)tot) : 0ull;
        }
        __syncthreads();
        unsigned long long pos = chunk_base;
        for (int w = 0; w < wave; ++w) pos += wave_cnt[w];
        const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
        for (int e = 0; e < PER_THREAD; ++e) {
            if ((keep_bits >> e) & 1u) {
                const long long x = x0 + (long long)e * 256;
                const unsigned long long p = pos + (unsigned int)__popcll(bal[e] & below);
                bi[p] = ai[x];
                bj[p] = aj[x];
                bs[p] = s[e];
            }
            pos += (unsigned int)__popcll(bal[e]);
        }
        __syncthreads();
    }
}

int launch_filter_hits(const int32_t* ai, const int32_t* aj, const float* as, long long n, float radius, int32_t* bi, int32_t* bj,
                       float* bs, unsigned long long* n_out, hipStream_t stream) {
    VSC_HIP(hipMemsetAsync(n_out, 0, sizeof(unsigned long long), stream));
    if (n <= 0) return VSC_OK;
    const int grid = (int)std::min<long long>(1024, (n + 2047) / 2048);
    hipLaunchKernelGGL(filter_hits_kernel, dim3(grid), dim3(256), 0, stream, ai, aj, as, n, radius, bi, bj, bs, n_out);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

int launch_score_hist(const float* s, long long n, const long long* state, int shift, long long* hist, hipStream_t stream) {
    VSC_HIP(hipMemsetAsync(hist, 0, 256 * sizeof(long long), stream));
    if (n > 0) {
        const int grid = (int)std::min<long long>(1024, (n / 4 + 255) / 256 + 1);
        hipLaunchKernelGGL(score_hist_kernel, dim3(grid), dim3(256), 0, stream, s, n, state, shift,
                           reinterpret_cast<unsigned long long*>(hist));
        VSC_HIP(hipGetLastError());
    }
    return VSC_OK;
}

int launch_score_pick(const long long* hist, long long* state, int shift, hipStream_t stream) {
    hipLaunchKernelGGL(score_pick_kernel, dim3(1), dim3(64), 0, stream, hist, state, shift);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

// Per-row merge of reference shards' k-NN lists (vsc_merge_topk; BASELINE configs[4], vsc2022_amd/dist.py:ref_sharded_knn): row x
// holds m = shards x k candidates (score, global reference id; id < 0 = empty slot); the k best under (score desc, id asc)
// -- the order one index over the concatenated reference set produces -- by rank counting: every candidate counts the
// candidates that beat it (m <= 1024: <= 16 per lane, the row in LDS), and the ones with rank < k go to slot `rank`.
// One wavefront per row; ids are unique across shards, empty slots order by position.
__global__ __launch_bounds__(256) void merge_topk_kernel(const float* __restrict__ s, const long long* __restrict__ ids, long long nq,
                                                         int m, int k, float* __restrict__ out_s, long long* __restrict__ out_ids) {
    extern __shared__ unsigned char lds_raw[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* ls = reinterpret_cast<float*>(lds_raw) + (size_t)wave * m;
    long long* li = reinterpret_cast<long long*>(lds_raw + (size_t)4 * m * 4) + (size_t)wave * m;
    const long long row = (long long)blockIdx.x * 4 + wave;
    if (row >= nq) return;
    for (int e = lane; e < m; e += 64) {
        const long long id = ids[row * m + e];
        ls[e] = id < 0 ? -INFINITY : s[row * m + e];
        li[e] = id < 0 ? 0x7fffffffffffffffLL : id;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0);  // (the row's LDS writes of this wave are complete)
    for (int e = lane; e < m; e += 64) {
        const float se = ls[e];
        const long long ie = li[e];
        int rank = 0;
        for (int f = 0; f < m; ++f) {
            const float sf = ls[f];
            const long long jf = li[f];
            rank += (sf > se) || (sf == se && (jf < ie || (jf == ie && f < e)));
        }
        if (rank < k) {
            const bool empty = ie == 0x7fffffffffffffffLL;
            out_s[row * k + rank] = empty ? -FLT_MAX : se;
            out_ids[row * k + rank] = empty ? -1 : ie;
        }
    }
}

int launch_merge_topk(const float* s, const long long* ids, long long nq, int m, int k, float* out_s, long long* out_ids,
                      hipStream_t stream) {
    if (nq <= 0) return VSC_OK;
    const size_t lds = (size_t)4 * m * (4 + 8);
    hipLaunchKernelGGL(merge_topk_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), lds, stream, s, ids, nq, m, k, out_s, out_ids);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

}  // namespace vscmi
