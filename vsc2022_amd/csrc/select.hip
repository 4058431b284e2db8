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

}  // namespace vscmi
