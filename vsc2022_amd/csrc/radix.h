// Device-wide stable LSD radix sort of (key, value) pairs, 8 bits per pass (gfx950).
//
// Used for the final ordering of the kept hits, the (query video, ref video) grouping and the k-NN cut
// (sortpairs.hip): <= a few 10^7 pairs, HBM-bound plumbing behind the search.  Three kernels per pass:
//
//   radix_hist     every workgroup counts the digits of its tile of 256 x RADIX_ITEMS keys (LDS atomics) and
//                  writes one column of the digit-major table hist[digit][tile];
//   radix_scan_*   exclusive prefix sum of the table in digit-major order = the first output position of every
//                  (digit, tile) -- one workgroup per digit row, then the 256 row totals;
//   radix_scatter  every workgroup re-reads its tile and places each pair at position
//                  base[digit][tile] + (pairs of the same digit earlier in the tile).  Stability inside the tile:
//                  elements are visited in tile order (wave, slot, lane); the lanes of a wave that hold the same
//                  digit in a slot find each other with 8 ballots (one per digit bit) and take consecutive ranks
//                  behind a per-wave LDS counter of that digit; the waves' counters are prefix-summed at the end.
//                  The tile is first sorted by digit through LDS, so that the global writes are coalesced runs.
//
// Cost per pass and pair: key read twice, key + value written once -- (2 sizeof(K) + sizeof(V)) read +
// (sizeof(K) + sizeof(V)) written; passes = ceil((end_bit - begin_bit) / 8).
#pragma once
#include "vscmi_common.h"

namespace vscmi {

constexpr int RADIX_ITEMS = 16;                   // keys per thread
constexpr int RADIX_TILE = 256 * RADIX_ITEMS;     // keys per workgroup

template <class K>
__device__ __forceinline__ unsigned radix_digit(K key, int shift, bool desc) {
    if (desc) key = ~key;
    return (unsigned)(key >> shift) & 255u;
}

template <class K>
__global__ __launch_bounds__(256) void radix_hist_kernel(const K* __restrict__ keys, int64_t n, int shift, bool desc,
                                                         unsigned* __restrict__ hist, int ntiles) {
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RADIX_TILE;
#pragma unroll 4
    for (int k = 0; k < RADIX_ITEMS; ++k) {
        const int64_t e = base + k * 256 + threadIdx.x;
        if (e < n) atomicAdd(&h[radix_digit(keys[e], shift, desc)], 1u);
    }
    __syncthreads();
    hist[(int64_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

// one workgroup per digit row: exclusive scan of its ntiles counts in place; row total -> totals[digit]
__global__ __launch_bounds__(256) void radix_scan_rows_kernel(unsigned* __restrict__ hist, int ntiles,
                                                              unsigned* __restrict__ totals) {
    __shared__ unsigned part[256];
    unsigned* row = hist + (int64_t)blockIdx.x * ntiles;
    const int per = (ntiles + 255) / 256;  // consecutive entries per thread
    const int t0 = threadIdx.x * per;
    unsigned s = 0;
    for (int t = t0; t < min(ntiles, t0 + per); ++t) s += row[t];
    part[threadIdx.x] = s;
    __syncthreads();
    // Hillis-Steele inclusive scan of the 256 partial sums
    for (int off = 1; off < 256; off <<= 1) {
        const unsigned v = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned run = threadIdx.x ? part[threadIdx.x - 1] : 0u;
    for (int t = t0; t < min(ntiles, t0 + per); ++t) {
        const unsigned c = row[t];
        row[t] = run;
        run += c;
    }
    if (threadIdx.x == 255) totals[blockIdx.x] = part[255];
}

__global__ __launch_bounds__(256) void radix_scan_totals_kernel(unsigned* __restrict__ totals) {
    __shared__ unsigned part[256];
    part[threadIdx.x] = totals[threadIdx.x];
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const unsigned v = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    totals[threadIdx.x] = threadIdx.x ? part[threadIdx.x - 1] : 0u;  // exclusive
}

template <class K, class V>
__global__ __launch_bounds__(256) void radix_scatter_kernel(const K* __restrict__ keys_in, const V* __restrict__ vals_in,
                                                            K* __restrict__ keys_out, V* __restrict__ vals_out, int64_t n,
                                                            int shift, bool desc, const unsigned* __restrict__ hist,
                                                            const unsigned* __restrict__ totals, int ntiles) {
    __shared__ unsigned wcount[4][256];  // per wave: pairs of each digit seen so far, then its offset inside the digit
    __shared__ unsigned dstart[256];     // first slot of each digit in the tile's digit-major order
    __shared__ unsigned gbase[256];      // first output position of each digit of this tile
    __shared__ K skey[RADIX_TILE];
    __shared__ V sval[RADIX_TILE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int x = threadIdx.x; x < 4 * 256; x += 256) (&wcount[0][0])[x] = 0;
    gbase[threadIdx.x] = totals[threadIdx.x] + hist[(int64_t)threadIdx.x * ntiles + blockIdx.x];
    __syncthreads();
    // tile order: (wave, slot, lane)
    const int64_t tile0 = (int64_t)blockIdx.x * RADIX_TILE;
    const int64_t base = tile0 + (int64_t)wave * (RADIX_ITEMS * 64);
    K key[RADIX_ITEMS];
    unsigned rank[RADIX_ITEMS];
#pragma unroll
    for (int k = 0; k < RADIX_ITEMS; ++k) {
        const int64_t e = base + k * 64 + lane;
        const bool valid = e < n;
        key[k] = valid ? keys_in[e] : (K)0;
        const unsigned d = radix_digit(key[k], shift, desc);
        // lanes of this slot with the same digit
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const unsigned before = (unsigned)__popcll(peers & ((1ull << lane) - 1));
        unsigned c = 0;
        if (valid) c = wcount[wave][d];  // same value for all peers (read before any of them writes)
        __builtin_amdgcn_wave_barrier();
        if (valid && before == 0) wcount[wave][d] = c + (unsigned)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
        rank[k] = c + before;
    }
    __syncthreads();
    // per digit: the waves' counts become exclusive offsets inside the digit; digit totals -> digit-major starts
    {
        const int d = threadIdx.x;
        unsigned run = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const unsigned c = wcount[w][d];
            wcount[w][d] = run;
            run += c;
        }
        dstart[d] = run;
    }
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {  // inclusive scan of the digit totals
        const unsigned v = threadIdx.x >= off ? dstart[threadIdx.x - off] : 0u;
        __syncthreads();
        dstart[threadIdx.x] += v;
        __syncthreads();
    }
    const unsigned mine_incl = dstart[threadIdx.x];
    const unsigned prev_incl = threadIdx.x ? dstart[threadIdx.x - 1] : 0u;
    __syncthreads();
    dstart[threadIdx.x] = prev_incl;  // exclusive
    (void)mine_incl;
    __syncthreads();
    // local sort through LDS: every digit's pairs become one run of consecutive slots (stable), so that the
    // global writes below are coalesced per run instead of 12 scattered bytes per lane
#pragma unroll
    for (int k = 0; k < RADIX_ITEMS; ++k) {
        const int64_t e = base + k * 64 + lane;
        if (e < n) {
            const unsigned d = radix_digit(key[k], shift, desc);
            const unsigned slot = dstart[d] + wcount[wave][d] + rank[k];
            skey[slot] = key[k];
            sval[slot] = vals_in[e];
        }
    }
    __syncthreads();
    const int cnt = (int)min((int64_t)RADIX_TILE, n - tile0);
#pragma unroll 4
    for (int k = 0; k < RADIX_ITEMS; ++k) {
        const int slot = k * 256 + threadIdx.x;
        if (slot < cnt) {
            const K kk = skey[slot];
            const unsigned d = radix_digit(kk, shift, desc);
            const int64_t pos = (int64_t)gbase[d] + (slot - dstart[d]);
            keys_out[pos] = kk;
            vals_out[pos] = sval[slot];
        }
    }
}

inline size_t radix_tmp_bytes(int64_t n) {
    const int64_t ntiles = (n + RADIX_TILE - 1) / RADIX_TILE;
    return (size_t)(256 * ntiles + 256) * sizeof(unsigned);
}

// Stable sort of n pairs by key bits [begin_bit, end_bit), ascending (or descending).  The buffers ping-pong; the
// result is left in (keys_b, vals_b) when the number of passes is odd, else in (keys_a, vals_a): the return
// value tells (0 = a, 1 = b), < 0 = error.  `tmp` holds radix_tmp_bytes(n).  n < 2^32.
// (begin2, end2): a second bit range sorted after the first -- keys made of two fields with unused bits between them
// (row << 32 | ref, query video << 32 | ref video) skip the passes over the gap (round 5: the pair-max sorted 64 bits for
// two 16-bit fields, 8 passes instead of 4)
template <class K, class V>
int radix_sort_pairs(K* keys_a, K* keys_b, V* vals_a, V* vals_b, int64_t n, int begin_bit, int end_bit, bool desc,
                     void* tmp, hipStream_t stream, int begin2 = 0, int end2 = 0) {
    if (n <= 0) return 0;
    if (n > 0xffffffffLL) {  // histograms, totals and scatter offsets are 32-bit
        set_error("radix_sort_pairs: %lld entries exceed the 2^32 limit of the sort", (long long)n);
        return -1;
    }
    const int ntiles = (int)((n + RADIX_TILE - 1) / RADIX_TILE);
    unsigned* hist = reinterpret_cast<unsigned*>(tmp);
    unsigned* totals = hist + (size_t)256 * ntiles;
    int where = 0;
    for (int pass = 0;; ++pass) {
        const int n1 = (end_bit - begin_bit + 7) / 8;
        const int shift = pass < n1 ? begin_bit + 8 * pass : begin2 + 8 * (pass - n1);
        if (pass >= n1 && shift >= end2) break;
        const K* ki = where ? keys_b : keys_a;
        const V* vi = where ? vals_b : vals_a;
        K* ko = where ? keys_a : keys_b;
        V* vo = where ? vals_a : vals_b;
        hipLaunchKernelGGL((radix_hist_kernel<K>), dim3((unsigned)ntiles), dim3(256), 0, stream, ki, n, shift, desc, hist, ntiles);
        hipLaunchKernelGGL(radix_scan_rows_kernel, dim3(256), dim3(256), 0, stream, hist, ntiles, totals);
        hipLaunchKernelGGL(radix_scan_totals_kernel, dim3(1), dim3(256), 0, stream, totals);
        hipLaunchKernelGGL((radix_scatter_kernel<K, V>), dim3((unsigned)ntiles), dim3(256), 0, stream, ki, vi, ko, vo, n, shift,
                           desc, hist, totals, ntiles);
        if (hipGetLastError() != hipSuccess) return -1;
        where ^= 1;
    }
    return where;
}

}  // namespace vscmi
