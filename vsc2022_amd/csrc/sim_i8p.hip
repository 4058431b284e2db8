// Panel-stationary INT8 pre-filter of the thresholded searches (gfx950), dims <= 1024.
//
// Same contract as sim_f16p.hip -- candidates = a superset of the pairs whose exact fp32 score reaches the
// threshold; vsc/index.py:142-165 semantics are restored by the exact stage (rescore_kernel) -- and the same data
// flow (128-row query panel resident in LDS, reference fragments streamed straight into registers from a
// fragment-major image, 8 waves x (4 x 2) blocks of 32x32, work items behind one atomic counter per panel), on
// v_mfma_i32_32x32x32_i8: twice the k per instruction at the same issue cadence.  Measured on the power-capped
// parts of this pool (scripts/ubench/i8_panel.hip, profiles/r03_i8_ubench.md): the skeleton sustains 2930 TOP/s
// against 1440 TFLOP/s of the fp16 one on the same box.
//
// Why this is still exact: the integer accumulators are EXACT dot products of the quantised rows, so the only
// error is quantisation, and quant_i8.hip records per row what was actually lost:
//     x = s (q + e),  E >= ||x - s q||,  N >= ||x||   =>   | x.y - s_x s_y (q_x . q_y) | <= E_x N_y + (N_x + E_x) E_y
// plus c_acc N_x N_y for the rounding of the exact fp32 chain itself.  (Coordinates on which all references agree are
// kept out of the images and enter through per-row thresholds instead: quant_i8.hip, "EXCLUDED coordinates".)  A pair whose exact score exceeds the radius
// therefore has   q_x . q_y  >  (radius - eps_xy) / (s_x s_y);   the kernel tests the integer accumulator against
// the floor of a lower bound of that quotient, per reference column (one scale per reference ROW, one scale and the
// largest E / N per query PANEL).  With 8 bits the bound is ~16x looser than the fp16 one (eps ~ 0.018 for unit
// 512-d rows: 4-5x as many candidates), which is why api.hip uses this kernel only where hits are sparse.
#include "kernels.h"

namespace vscmi {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

namespace i8p {

constexpr int PR = F16P_PANEL_ROWS;  // 128
constexpr int CSW = F16P_COL_STEP;   // 512
#ifndef VSC_I8P_PF
#define VSC_I8P_PF 4
#endif
constexpr int PF = VSC_I8P_PF;       // register ring: k-steps (PF - 1 in flight, 2 KiB each per wave).  Measured on the
                                     // bench: 8 loses 3 % (2391 vs 2474 TOP/s); delaying the second wave of every
                                     // SIMD by 1000-4000 cycles after an item's barrier changes nothing (2431-2435)

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, int voff, char* lds) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
}
__device__ __forceinline__ i32x4 bload(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}
#ifndef VSC_I8P_ASMLOAD
#define VSC_I8P_ASMLOAD 1  // r03: +5 % without candidates (2601 -> 2735 TOP/s), +1 % on the bench, the k-NN passes and config 4
#endif
// The reference stream with hand-placed waits: the loads are opaque to the compiler's s_waitcnt insertion, which
// otherwise (a) drains the whole stream with vmcnt(0) at the start of every tile -- it loses count of the outstanding
// loads across the emission branches -- and (b) waits vmcnt(3) where the ring allows vmcnt(4).  A load's destination
// must not be touched before ring_wait has named it (the "+v" ties keep every use behind the wait).
__device__ __forceinline__ void bload_asm0(i32x4& dst, __amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&v"(dst) : "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ void bload_asm1(i32x4& dst, __amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:1024" : "=&v"(dst) : "v"(voff), "s"(rs), "s"(soff) : "memory");
}
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gload_asm(f32x4v& dst, const float4* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void meta_wait(f32x4v& m0, f32x4v& m1) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(m0), "+v"(m1) : "n"(N));
}
template <int N>
__device__ __forceinline__ void ring_wait(i32x4& b0, i32x4& b1) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(b0), "+v"(b1) : "n"(N));
}
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ float uniform_f(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}

// One output tile (128 panel rows x the wave's 64 columns, K = NKC x 256): A fragments from the LDS panel one
// k-step ahead, B fragments from the ring, refilled PF-1 k-steps ahead; the stream continues into the wave's next
// tile at `so_next`.  Straight-line code, every LDS address = base register + immediate; issue order pinned.
template <int NKC>
__device__ __forceinline__ void tile_mma(const char* smem, const int (&abase)[8], i32x4 (&a)[4], i32x4 (&ring)[PF][2],
                                         __amdgpu_buffer_rsrc_t rs, int so_tile, int so_next, int lane16,
                                         i32x16 (&acc)[4][2]) {
    const i32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    constexpr int NKS = NKC * 8;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int t = ks + PF - 1;  // k-step that goes into the ring slot freed by k-step ks - 1
        const int so = (t < NKS) ? so_tile + t * 2048 : so_next + (t - NKS) * 2048;
        const int kn = (ks + 1) % NKS;  // A fragments of the next k-step (the next tile starts at 0 again)
        const char* anext = smem + (kn >> 3) * 32768 + abase[kn & 7];
#if VSC_I8P_ASMLOAD
        // both fragments of this k-step have landed once at most the 2 (PF - 2) loads of the younger k-steps are out
        ring_wait<2 * (PF - 2)>(ring[ks % PF][0], ring[ks % PF][1]);
#endif
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            acc[m][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m], ring[ks % PF][0], ks == 0 ? zero : acc[m][0], 0, 0, 0);
            acc[m][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m], ring[ks % PF][1], ks == 0 ? zero : acc[m][1], 0, 0, 0);
            a[m] = *reinterpret_cast<const i32x4*>(anext + m * 8192);
#if VSC_I8P_ASMLOAD
            if (m == 0) bload_asm0(ring[(ks + PF - 1) % PF][0], rs, lane16, so);
            if (m == 1) bload_asm1(ring[(ks + PF - 1) % PF][1], rs, lane16, so);
#else
            if (m < 2) ring[(ks + PF - 1) % PF][m] = bload(rs, lane16 + m * 1024, so);
#endif
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#if !VSC_I8P_ASMLOAD
            if (m < 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#endif
        }
    }
}

// lower edge of the candidate test for an exact threshold t (as in sim_f16p.hip); t = +inf (the rows of a panel that
// lie past the batch) stays +inf instead of turning into inf - inf
__device__ __forceinline__ float candidate_edge(float t, float eps) {
    return t == INFINITY ? t : (t - eps) - 2.4e-7f * (fabsf(t) + eps);
}
// A lower bound of edge / (s_q s_r) from inv = (1 / s_q) (1 / s_r): the few roundings of the quotient are covered by
// 2e-6 relative; the absolute term keeps a quotient that underflowed to +-0 on the safe side of the integers.
__device__ __forceinline__ float quotient_low(float edge, float inv) {
    const float t = edge * inv;
    if (!(fabsf(t) < INFINITY)) return t;  // +-inf thresholds (rows past the batch: +inf) stay what they are; NaN too
    return t - fabsf(t) * 2e-6f - 1e-3f;
}
// integer accumulator > t  <=>  accumulator > floor(t); out-of-range thresholds saturate to never / always
__device__ __forceinline__ int floor_sat(float t) { return (int)floorf(fminf(fmaxf(t, -2.1e9f), 2.1e9f)); }

__device__ __forceinline__ int lane_now() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// Candidates of one wave tile -> the wave's private segment (no atomics; the wave's chunk of the shared tail when the
// segment is full).  ti[m][n] = integer threshold of 32-row block m x the lane's column of block column n: an
// accumulator above it is a candidate.  Radius search: the same threshold for all four row blocks.  Per-row
// thresholds (k-NN, excluded coordinates): the threshold of the block's SMALLEST row threshold -- the rows of a
// launch arrive sorted by threshold (sortpairs.hip), so a block's 32 thresholds are next to equal and testing all
// of its rows against the smallest one passes a few candidates more instead of costing a float comparison per
// accumulator.  General form: edge tiles, full segments, pass-everything columns.
__device__ __forceinline__ void emit_candidates(const SimI8PArgs& a, const bool (&all)[2], const int (&ti)[4][2], int row0,
                                                int col0, bool interior, const i32x16 (&acc)[4][2],
                                                const int (&bm)[4][2], int64_t seg_base, int& count, TailExt* ext) {
    // C layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            if (!__any(all[n] || bm[m][n] > ti[m][n])) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rb = m * 32 + (r & 3) + 8 * (r >> 2);  // + 4 * (lane >> 5) = row inside the panel
                const bool hit = all[n] || acc[m][n][r] > ti[m][n];
                const unsigned long long hits = __ballot(hit);
                if (hits == 0ull) continue;
                const int ln = lane_now();
                const int i = row0 + rb + 4 * (ln >> 5);
                const int j = col0 + n * 32 + (ln & 31);
                const bool mine = hit && (interior || (i < a.nq && j < a.nr));
                const unsigned long long ok = interior ? hits : __ballot(mine);
                if (ok == 0ull) continue;
                const int total = __popcll(ok);
                int64_t pos;
                if (count + total <= a.seg_cap) {
                    pos = seg_base + count;
                    count += total;
                } else {
                    // segment full (candidates are not spread evenly): the wave's chunk of the shared tail
                    if (!tail_take(a.tail_count, a.tail_cap, a.tail_base, a.tail_shift, a.tail_fill, a.overflow, total, ln, ext, pos))
                        continue;
                }
                if (mine) {
                    pos += __builtin_amdgcn_mbcnt_hi((unsigned)(ok >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ok, 0u));
                    a.out_i[pos] = a.i0 + i;  // (a position when the launch's rows are permuted: the exact stage maps it back)
                    a.out_j[pos] = j;
                }
            }
        }
}

// The fast path: interior tile, room for a whole tile (8192 entries) left in the wave's segment: position = count +
// (lanes of this register's ballot below me), two buffer stores with a 32-bit offset.
__device__ __forceinline__ void emit_candidates_seg(const SimI8PArgs& a, const int (&ti)[4][2], int row0, int col0,
                                                    const i32x16 (&acc)[4][2], const int (&bm)[4][2],
                                                    __amdgpu_buffer_rsrc_t rs_i, __amdgpu_buffer_rsrc_t rs_j, int& count) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            if (!__any(bm[m][n] > ti[m][n])) continue;
            const int ln = lane_now();  // (live inside the block only: the register file is full)
            const int pbase = row0 + m * 32 + 4 * (ln >> 5);
            const int j = col0 + n * 32 + (ln & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool cand = acc[m][n][r] > ti[m][n];
                const unsigned long long hits = __ballot(cand);
                if (hits == 0ull) continue;
                if (cand) {
                    const int off = (count + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(hits >> 32),
                                                                            __builtin_amdgcn_mbcnt_lo((unsigned)hits, 0u)))
                                    << 2;
                    const int p = pbase + (r & 3) + 8 * (r >> 2);
                    __builtin_amdgcn_raw_buffer_store_b32(a.i0 + p, rs_i, off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(j, rs_j, off, 0, 0);
                }
                count += __popcll(hits);
            }
        }
}

}  // namespace i8p

template <int NKC, bool ROWTHR>
__global__ __launch_bounds__(512) void sim_i8p_kernel(SimI8PArgs a) {
    using namespace i8p;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // the panel: NKC x 32 KiB
    __shared__ float rt_sh[ROWTHR ? PR : 1];
    __shared__ int item_sh[2];
    __shared__ TailExt tail_sh[8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NKS = NKC * 8;
    constexpr int ROWB = NKC * 256;    // bytes per int8 row
    constexpr int TILEB = NKS * 2048;  // bytes per 64-row wave tile of the fragment-major image
    const int lane16 = lane * 16;
    int abase[8];
    {
        const int hi = lane >> 5, r15 = lane & 15, rl = lane & 31;
#pragma unroll
        for (int u = 0; u < 8; ++u) abase[u] = rl * 256 + ((((2 * u) | hi) ^ r15) << 4);
    }
    const float radius = ROWTHR ? 0.0f : *a.radius;
    const int seg = blockIdx.x * 8 + wave;  // this wave's private segment of the candidate list
    const int64_t seg_base = (int64_t)seg * a.seg_cap;
    int count = 0;
    tail_init(&tail_sh[wave], lane);  // the wave's chunk of the shared tail once its segment is full (cand_list.h)
    const __amdgpu_buffer_rsrc_t rs_ci = __builtin_amdgcn_make_buffer_rsrc(
        (void*)uniform_ptr(reinterpret_cast<const char*>(a.out_i + seg_base)), 0, a.seg_cap * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_cj = __builtin_amdgcn_make_buffer_rsrc(
        (void*)uniform_ptr(reinterpret_cast<const char*>(a.out_j + seg_base)), 0, a.seg_cap * 4, 0x00020000);
    const int nslice = (a.nsteps + a.slice - 1) / a.slice;
    int cur_panel = -1;
    // per panel (wave-uniform): 1 / s_q and the coefficients of N'_r, E_r and N_r in eps
    float inv_sq = 1.0f, coef_k = 0.0f, coef_e = 0.0f, coef_n = 0.0f;
    float rtmin[4] = {0.f, 0.f, 0.f, 0.f};
    int panel = blockIdx.x % a.npanel;
    for (;;) {
        // ---- next work item: a slice of this workgroup's panel, else of the panel with the most left
        __syncthreads();
        if (wave == 0) {
            int p = panel, s = 0;
            // a launch whose candidate list has overflowed is lost (the host reruns the batch on the fp16 kernel or
            // with larger buffers): stop taking work instead of pushing billions of candidates through the tail's
            // one atomic counter (an 8-bit bound that is too loose for the data can pass most of the matrix)
#ifndef VSC_NO_LOST_CHECK
            const bool lost = __hip_atomic_load(a.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
#else
            const bool lost = false;
#endif
            if (a.order == 1 && !lost) {
                // slice-major: items (slice, panel) in one global order -- every workgroup of the chip is inside
                // the same few MB of the reference image, each XCD fetches a slice once; the price is a panel load
                // (64 KiB at 512-d) per item
                int t = 0;
                if (lane == 0) t = atomicAdd(&a.next_slice[a.npanel], 1);
                t = __shfl(t, 0);
                if (t >= nslice * a.npanel) { p = -1; } else { s = t / a.npanel; p = t - s * a.npanel; }
            } else
            for (;;) {
                if (lost) { p = -1; break; }
                if (lane == 0) s = atomicAdd(&a.next_slice[p], 1);
                s = __shfl(s, 0);
                if (s < nslice) break;
                int best = 0x7fffffff, bp = 0x7fffffff;
                for (int q0 = 0; q0 < a.npanel; q0 += 64) {
                    const int q = q0 + lane;
                    const int pp = (q + (int)blockIdx.x) % a.npanel;  // ties: nearest after the workgroup's own
                    const int v = q < a.npanel ? __hip_atomic_load(&a.next_slice[pp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                               : 0x7fffffff;
                    if (v < best) { best = v; bp = pp; }
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const int ob = __shfl_xor(best, off), op = __shfl_xor(bp, off);
                    if (ob < best || (ob == best && op < bp)) { best = ob; bp = op; }
                }
                if (best >= nslice) { p = -1; break; }
                p = bp;
            }
            if (lane == 0) { item_sh[0] = p; item_sh[1] = s; }
        }
        __syncthreads();
        panel = item_sh[0];
        const int sl = item_sh[1];
        if (panel < 0) break;
        const int cs0 = sl * a.slice, cs1 = min(a.nsteps, cs0 + a.slice);
        if (panel != cur_panel) {
            __syncthreads();  // (item_sh is read; nobody reads the old panel any more)
            // query panel -> LDS [k chunk of 256 B][row][slot ^ (row & 15)]: the swizzle is applied to the SOURCE
            // address, an LDS-DMA instruction writes its 64 x 16 bytes to consecutive LDS addresses
            const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(
                (void*)uniform_ptr(reinterpret_cast<const char*>(a.Q) + (int64_t)panel * PR * ROWB), 0, PR * ROWB, 0x00020000);
#pragma unroll
            for (int n = 0; n < NKC * 4; ++n) {
                const int p = n * 512 + tid;
                const int kc = p >> 11, row = (p >> 4) & 127, slot = p & 15;
                const int c = kc * 16 + (slot ^ (row & 15));
                dma16(qrs, row * ROWB + c * 16, smem + (n * 512 + wave * 64) * 16);
            }
            const float4 ps = a.pstat[panel];  // {1 / s_q, max E_q, max N_q, max N'_q}
            inv_sq = uniform_f(ps.x);
            // eps_j = E_q N'_r + (N'_q + E_q) E_r + c_acc N_q N_r   (N' = norm over the coordinates the images hold)
            coef_k = uniform_f(ps.y);
            coef_e = uniform_f(ps.w + ps.y);
            coef_n = uniform_f(a.c_acc * ps.z);
            if (ROWTHR && tid < PR) rt_sh[tid] = panel * PR + tid < a.nq ? a.row_thr[(int64_t)panel * PR + tid] : INFINITY;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces have landed ...
            __syncthreads();                                  // ... and so have everybody else's
            if (ROWTHR) {
                float v0 = rt_sh[lane], v1 = rt_sh[64 + lane];
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) {
                    v0 = fminf(v0, __shfl_xor(v0, off));
                    v1 = fminf(v1, __shfl_xor(v1, off));
                }
                rtmin[0] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v0), 0));
                rtmin[1] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v0), 32));
                rtmin[2] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v1), 0));
                rtmin[3] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v1), 32));
            }
            cur_panel = panel;
        }
        // ---- this wave's stream: tiles (cs * 8 + wave), cs = cs0 .. cs1-1, TILEB contiguous bytes each
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)uniform_ptr(reinterpret_cast<const char*>(a.Rf) + (int64_t)cs0 * 8 * TILEB), 0, (cs1 - cs0) * 8 * TILEB,
            0x00020000);
        int so_tile = wave * TILEB;
        i32x4 ring[PF][2];
#pragma unroll
        for (int dd = 0; dd < PF - 1; ++dd) {
#if VSC_I8P_ASMLOAD
            bload_asm0(ring[dd][0], rs, lane16, so_tile + dd * 2048);
            bload_asm1(ring[dd][1], rs, lane16, so_tile + dd * 2048);
#else
            ring[dd][0] = bload(rs, lane16, so_tile + dd * 2048);
            ring[dd][1] = bload(rs, lane16 + 1024, so_tile + dd * 2048);
#endif
        }
        i32x4 afr[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) afr[m] = *reinterpret_cast<const i32x4*>(smem + abase[0] + m * 8192);
        for (int cs = cs0; cs < cs1; ++cs) {
            const int col0 = cs * CSW + wave * 64;
            // {1 / s_r, E_r, N_r, N'_r} of the lane's two columns (the table is padded to whole col-steps)
#if VSC_I8P_ASMLOAD
            // (issued behind the stream loads of the previous tile, ahead of this tile's: after the K loop only the
            // 2 (PF - 1) stream loads of the next tile are younger)
            f32x4v m0, m1;
            gload_asm(m0, a.rmeta + col0 + (lane & 31));
            gload_asm(m1, a.rmeta + col0 + 32 + (lane & 31));
#else
            const float4 m0 = a.rmeta[col0 + (lane & 31)], m1 = a.rmeta[col0 + 32 + (lane & 31)];
#endif
            i32x16 acc[4][2];
            tile_mma<NKC>(smem, abase, afr, ring, rs, so_tile, so_tile + 8 * TILEB, lane16, acc);
            so_tile += 8 * TILEB;
#if VSC_I8P_ASMLOAD
            meta_wait<2 * (PF - 1)>(m0, m1);
#endif
            const float eps[2] = {(coef_k * m0.w + coef_e * m0.y + coef_n * m0.z) * 1.001f,
                                  (coef_k * m1.w + coef_e * m1.y + coef_n * m1.z) * 1.001f};
            const float inv[2] = {inv_sq * m0.x, inv_sq * m1.x};
            // integer thresholds per (row block, column).  Radius search: strict, floor of a lower bound of the
            // quotient.  Row thresholds: the block's smallest one, non-strict (accumulator >= T  <=  accumulator >
            // floor(T) - 1), so that pairs tied with a row's threshold survive.
            float tl[4][2];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float t = ROWTHR ? rtmin[m] : radius;
                tl[m][0] = (ROWTHR || m == 0) ? quotient_low(candidate_edge(t, eps[0]), inv[0]) : tl[0][0];
                tl[m][1] = (ROWTHR || m == 0) ? quotient_low(candidate_edge(t, eps[1]), inv[1]) : tl[0][1];
            }
            // pass everything where the arithmetic above says nothing: +inf / NaN bounds (unrepresentable rows), scale
            // products outside the range where their inverse is a normal number, NaN quotients (0 x inf, NaN thresholds)
            bool all[2] = {!(eps[0] < INFINITY) || !(inv[0] >= 1e-30f && inv[0] < INFINITY),
                           !(eps[1] < INFINITY) || !(inv[1] >= 1e-30f && inv[1] < INFINITY)};
            int ti[4][2];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    all[n] |= !(tl[m][n] == tl[m][n]);
                    ti[m][n] = floor_sat(tl[m][n]) - (ROWTHR ? 1 : 0);
                }
            // candidates are rare: one max per 32x32 block first, one compare for the whole wave tile
            int bm[4][2];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    int x = max(max(acc[m][n][0], acc[m][n][1]), acc[m][n][2]);
#pragma unroll
                    for (int r = 3; r < 15; r += 2) x = max(max(x, acc[m][n][r]), acc[m][n][r + 1]);
                    bm[m][n] = max(x, acc[m][n][15]);
                }
            bool any_blk = all[0] || all[1];
#pragma unroll
            for (int m = 0; m < 4; ++m) any_blk |= bm[m][0] > ti[m][0] || bm[m][1] > ti[m][1];
            if (__any(any_blk)) {
                const bool interior = panel * PR + PR <= a.nq && col0 + 64 <= a.nr;
                if (interior && !all[0] && !all[1] && count + 8192 <= a.seg_cap) {
                    emit_candidates_seg(a, ti, panel * PR, col0, acc, bm, rs_ci, rs_cj, count);
                } else {
                    emit_candidates(a, all, ti, panel * PR, col0, interior, acc, bm, seg_base, count, &tail_sh[wave]);
                }
            }
        }
    }
#if VSC_I8P_ASMLOAD
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the stream ran a few k-steps past its end)
#endif
    tail_close(a.tail_base, a.tail_shift, a.tail_fill, lane, &tail_sh[wave]);
    if (lane == 0) a.seg_count[seg] = count;
}

template <int NKC>
static int launch_nkc(const SimI8PArgs& a, int grid, hipStream_t stream) {
    const int lds = NKC * 32768;
    static PerDeviceOnce once;
    if (once.first()) {
        VSC_HIP(hipFuncSetAttribute((const void*)sim_i8p_kernel<NKC, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        VSC_HIP(hipFuncSetAttribute((const void*)sim_i8p_kernel<NKC, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        once.commit();
    }
    if (a.row_thr)
        hipLaunchKernelGGL((sim_i8p_kernel<NKC, true>), dim3((unsigned)grid), dim3(512), lds, stream, a);
    else
        hipLaunchKernelGGL((sim_i8p_kernel<NKC, false>), dim3((unsigned)grid), dim3(512), lds, stream, a);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

// (the work split is sim_f16p_plan's: same panel and col-step geometry)
int launch_sim_i8p(const SimI8PArgs& a, int grid, hipStream_t stream) {
    if (grid <= 0 || a.npanel <= 0 || a.nsteps <= 0) {
        // nothing to search: the caller's exact stage must see empty segments, not stale fill levels
        if (grid > 0) VSC_HIP(hipMemsetAsync(a.seg_count, 0, (size_t)grid * 8 * sizeof(int), stream));
        return VSC_OK;
    }
    VSC_HIP(hipMemsetAsync(a.next_slice, 0, ((size_t)a.npanel + 1) * sizeof(int), stream));
    switch (a.dpad8) {
        case 256: return launch_nkc<1>(a, grid, stream);
        case 512: return launch_nkc<2>(a, grid, stream);
        case 768: return launch_nkc<3>(a, grid, stream);
        case 1024: return launch_nkc<4>(a, grid, stream);
    }
    set_error("sim_i8p: dpad8 %d is not one of 256, 512, 768, 1024", a.dpad8);
    return VSC_ERR_INVALID;
}

}  // namespace vscmi
