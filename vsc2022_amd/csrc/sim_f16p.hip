// Panel-stationary fp16 pre-filter of the thresholded searches (gfx950), dims <= 512.
//
// Same contract as sim_f16.hip (candidates = every pair whose fp16 score + rigorous error bound reaches the
// threshold; vsc/index.py:142-165 semantics are restored by the exact stage), different data flow:
//
//   * a 128-row QUERY PANEL (all of K: <= 128 KiB of fp16) sits in LDS for as long as a workgroup works on
//     it -- [k chunk of 128][row][16-byte slot ^ (row & 15)], conflict-free for the 32x32x16 A-fragment
//     reads (16 lanes of a ds_read_b128 phase = 16 rows with distinct row & 15 = all 64 banks);
//   * the REFERENCE rows stream straight into registers from a FRAGMENT-MAJOR fp16 image (layout.hip): the B
//     fragment of one MFMA is one fully coalesced 1 KiB buffer load, fetched PF-1 k-steps ahead into a
//     register ring.  No LDS-DMA, no barrier and no LDS write in the steady state;
//   * 8 waves; wave w owns all 128 panel rows x columns [64 w, 64 w + 64) of a 512-column step: 4 x 2 blocks
//     of v_mfma_f32_32x32x16_f16, 128 accumulator registers, 2 waves per SIMD;
//   * work = (panel, slice of reference col-steps) items behind one atomic counter per panel.  A workgroup
//     stays on its panel while slices are left (the panel is loaded once), then helps the panel with the
//     most slices left: the XCDs of one part run at speeds 10 % apart under the power cap, a static split
//     leaves the fast ones idle at the end of every launch.
//
// Against the 256x256 LDS-ring kernel: the same bytes per flop out of L2, but none of them crosses the LDS
// twice, a third less LDS read traffic, no barriers -- MFMA busy 84 % instead of 64 % of the cycles, and the
// workgroups of an XCD walk the same reference stream in step, so that every line is fetched once per XCD.
// MFMA-bound (power-capped): 2 * 128 * 512 * dpadh flop per workgroup col-step.
#include "kernels.h"

namespace vscmi {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace f16p {

constexpr int PR = F16P_PANEL_ROWS;  // 128
constexpr int CSW = F16P_COL_STEP;   // 512
#ifndef VSC_F16P_PF
#define VSC_F16P_PF 4
#endif
constexpr int PF = VSC_F16P_PF;      // register ring: k-steps (PF - 1 in flight, 2 KiB each per wave).  Measured in the
                                     // product build: 4 beats 8 (1314 vs 1284 TFLOP/s thresholded, 1178 vs 1031 k-NN):
                                     // with 8 the register file is full and the compiler shortens the LDS read pipeline

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, int voff, char* lds) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
}
// cache policy of the reference stream (aux bits of the buffer load: 1 = sc0, 2 = nt, 16 = sc1)
#ifndef VSC_F16P_AUX
#define VSC_F16P_AUX 0
#endif
__device__ __forceinline__ f16x8 bload(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, VSC_F16P_AUX));
}
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}

// One output tile (128 panel rows x the wave's 64 columns, K = NKC x 128): A fragments from the LDS panel one
// k-step ahead, B fragments from the ring, refilled PF-1 k-steps ahead (voffset = 16 * lane + {0, 1024},
// soffset = position in the item's slice); the stream continues into the wave's next tile at `so_next`.
// Straight-line code, every LDS address = base register + immediate; the issue order is pinned (left alone
// the scheduler sinks every read to just before its use).  acc = (not +=) the product.
template <int NKC>
__device__ __forceinline__ void tile_mma(const char* smem, const int (&abase)[8], f16x8 (&a)[4], f16x8 (&ring)[PF][2],
                                         __amdgpu_buffer_rsrc_t rs, int so_tile, int so_next, int lane16,
                                         f32x16 (&acc)[4][2]) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    constexpr int NKS = NKC * 8;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int t = ks + PF - 1;  // k-step that goes into the ring slot freed by k-step ks - 1
        const int so = (t < NKS) ? so_tile + t * 2048 : so_next + (t - NKS) * 2048;
        const int kn = (ks + 1) % NKS;  // A fragments of the next k-step (the next tile starts at 0 again)
        const char* anext = smem + (kn >> 3) * 32768 + abase[kn & 7];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m], ring[ks % PF][0], ks == 0 ? zero : acc[m][0], 0, 0, 0);
            acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m], ring[ks % PF][1], ks == 0 ? zero : acc[m][1], 0, 0, 0);
            a[m] = *reinterpret_cast<const f16x8*>(anext + m * 8192);
            if (m < 2) ring[(ks + PF - 1) % PF][m] = bload(rs, lane16 + m * 1024, so);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (m < 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
    }
}

// lower edge of the candidate test for an exact threshold t: a pair with exact score >(=) t has fp16 score
// >= t - eps; the subtraction's own rounding (< 2^-23 relative to the larger operand) is subtracted again
__device__ __forceinline__ float candidate_edge(float t, float eps) { return (t - eps) - 2.4e-7f * (fabsf(t) + eps); }

// Candidates of one wave tile -> the wave's PRIVATE segment of the candidate list (no atomics, no scans; a
// shared atomic tail only when the segment is full).  Per 32x32 block one question first (its per-lane
// maximum is known), then one ballot per accumulator register of the few blocks that hold a candidate.
//   ROWTHR = false: thr[n] = edge of *radius for the lane's column of block column n (strict test)
//   ROWTHR = true : rt = the panel's 128 row thresholds, rtmin[m] = smallest of row block m, eps[n] = the
//                   lane's column bounds (non-strict test: k-NN ties must survive)
// The lane id, recomputed where it is used (two VALU instructions).  The emission code must not keep per-lane
// values (row / column offsets of the lane) alive across a tile: the register file is full, the compiler spills
// them, and a scratch reload in the hit path costs an s_waitcnt vmcnt(0) that also waits for the acknowledgement
// of every candidate store issued before it (~2 us per hit: half of a tile's time in the early, dense batches).
__device__ __forceinline__ int lane_now() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

template <bool ROWTHR>
__device__ __forceinline__ void emit_candidates(const SimF16PArgs& a, const bool (&all)[2], const float (&thr)[2],
                                                const float (&eps)[2], const float* rt, const float (&rtmin)[4],
                                                int row0, int col0, bool interior, const f32x16 (&acc)[4][2],
                                                const float (&bm)[4][2], int64_t seg_base, int& count, TailExt* ext) {
    // C layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const bool blk = all[n] || (ROWTHR ? bm[m][n] >= candidate_edge(rtmin[m], eps[n]) : bm[m][n] > thr[n]);
            if (!__any(blk)) continue;
            // k-NN: the row thresholds of the lane's 16 rows of this block are looked up by row index: the lane id
            // is taken once per flagged block (live inside the block only)
            const int ln_blk = ROWTHR ? lane_now() : 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rb = m * 32 + (r & 3) + 8 * (r >> 2);  // + 4 * (lane >> 5) = row inside the panel
                bool cand;
                if (ROWTHR)
                    cand = acc[m][n][r] >= candidate_edge(rt[rb + 4 * (ln_blk >> 5)], eps[n]);
                else
                    cand = acc[m][n][r] > thr[n];
                const bool hit = all[n] || cand;
                const unsigned long long hits = __ballot(hit);
                if (hits == 0ull) continue;
                const int ln = ROWTHR ? ln_blk : lane_now();
                const int i = row0 + rb + 4 * (ln >> 5);
                const int j = col0 + n * 32 + (ln & 31);
                // (tiles that reach past the batch or the references drop their padding rows / columns)
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
                    a.out_i[pos] = a.i0 + i;
                    a.out_j[pos] = j;
                }
            }
        }
}

// The same, block at a time: every lane gathers the candidates among its 16 values of a flagged 32x32 block in a bit
// mask, the wave takes an exclusive prefix sum of the per-lane counts (ballots of the counts' bit planes) and every
// lane writes its own candidates behind it.  ~90 instructions per flagged block whatever it holds, against
// 6 + ~35 per accumulator register with a candidate: 3-4x fewer in the early, dense batches of the schedule (a dozen
// candidates per block), slightly fewer in the sparse steady state.
template <bool ROWTHR>
__device__ __forceinline__ void emit_candidates_blk(const SimF16PArgs& a, const bool (&all)[2], const float (&thr)[2],
                                                    const float (&eps)[2], const float* rt, const float (&rtmin)[4],
                                                    int row0, int col0, bool interior, const f32x16 (&acc)[4][2],
                                                    const float (&bm)[4][2], int64_t seg_base, int& count, TailExt* ext) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const bool blk = all[n] || (ROWTHR ? bm[m][n] >= candidate_edge(rtmin[m], eps[n]) : bm[m][n] > thr[n]);
            if (!__any(blk)) continue;
            const int ln = lane_now();  // (live inside the block only, see above)
            const int hi4 = 4 * (ln >> 5);
            const int j = col0 + n * 32 + (ln & 31);
            unsigned mask = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rb = m * 32 + (r & 3) + 8 * (r >> 2);  // + hi4 = row inside the panel
                bool cand;
                if (ROWTHR)
                    cand = acc[m][n][r] >= candidate_edge(rt[rb + hi4], eps[n]);
                else
                    cand = acc[m][n][r] > thr[n];
                cand |= all[n];
                // (tiles that reach past the batch or the references drop their padding rows / columns)
                if (!interior) cand &= row0 + rb + hi4 < a.nq;
                mask |= cand ? (1u << r) : 0u;
            }
            if (!interior && j >= a.nr) mask = 0;
            const int cnt = __popc(mask);
            unsigned long long pl = __ballot(cnt & 1);
            int before = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(pl >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)pl, 0u));
            int total = __popcll(pl);
            if (__any(cnt > 1)) {
#pragma unroll
                for (int b = 1; b < 5; ++b) {
                    pl = __ballot((cnt >> b) & 1);
                    before += (int)__builtin_amdgcn_mbcnt_hi((unsigned)(pl >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)pl, 0u)) << b;
                    total += __popcll(pl) << b;
                }
            }
            if (total == 0) continue;
            int64_t pos;
            if (count + total <= a.seg_cap) {
                pos = seg_base + count;
                count += total;
            } else {
                // segment full (candidates are not spread evenly): the wave's chunk of the shared tail
                if (!tail_take(a.tail_count, a.tail_cap, a.tail_base, a.tail_shift, a.tail_fill, a.overflow, total, ln, ext, pos))
                    continue;
            }
            pos += before;
            const int ibase = a.i0 + row0 + m * 32 + hi4;
            while (mask) {
                const int r = __ffs(mask) - 1;
                mask &= mask - 1;
                a.out_i[pos] = ibase + (r & 3) + 8 * (r >> 2);
                a.out_j[pos] = j;
                ++pos;
            }
        }
}

// The radius search's fast path: interior tile, room for a whole tile (8192 entries) left in the wave's segment.  Then
// nothing needs checking and the per-candidate code shrinks to: position = count + (lanes of this register's ballot
// below me), two buffer stores with a 32-bit offset into the wave's own segment (rs_i / rs_j: buffer resources on the
// segment, uniform per wave).  Per flagged block: 16 compares + lane id and column (5 VALU); per register that holds a
// candidate: 4 VALU + 2 stores -- a third of the instructions of the other two forms.  Everything else (edge tiles,
// segment nearly full, +inf error bounds, the k-NN thresholds) goes through emit_candidates.
__device__ __forceinline__ void emit_candidates_seg(const SimF16PArgs& a, const float (&thr)[2], int row0, int col0,
                                                    const f32x16 (&acc)[4][2], const float (&bm)[4][2],
                                                    __amdgpu_buffer_rsrc_t rs_i, __amdgpu_buffer_rsrc_t rs_j, int& count) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            if (!__any(bm[m][n] > thr[n])) continue;
            const int ln = lane_now();  // (live inside the block only, see above)
            const int ibase = a.i0 + row0 + m * 32 + 4 * (ln >> 5);
            const int j = col0 + n * 32 + (ln & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool cand = acc[m][n][r] > thr[n];
                const unsigned long long hits = __ballot(cand);
                if (hits == 0ull) continue;
                if (cand) {
                    const int off = (count + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(hits >> 32),
                                                                            __builtin_amdgcn_mbcnt_lo((unsigned)hits, 0u)))
                                    << 2;
                    __builtin_amdgcn_raw_buffer_store_b32(ibase + (r & 3) + 8 * (r >> 2), rs_i, off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(j, rs_j, off, 0, 0);
                }
                count += __popcll(hits);
            }
        }
}

// 1: one ballot per accumulator register everywhere (emit_candidates); 2: block at a time everywhere; 0: block at a
// time for the radius search, per register for the k-NN thresholds; 3 (default): emit_candidates_seg on the fast path
// of the radius search, per register elsewhere.  Pre-filter time of a bench step: 317 ms (1), 310 (0), 304 (3); the
// k-NN (k = 1) 348 ms (1, 3) against 354 (2).
#ifndef VSC_F16P_EMIT
#define VSC_F16P_EMIT 3
#endif

}  // namespace f16p

template <int NKC, bool ROWTHR>
__global__ __launch_bounds__(512) void sim_f16p_kernel(SimF16PArgs a) {
    using namespace f16p;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // the panel: NKC x 32 KiB
    __shared__ float qn_max_w[8];
    __shared__ float rt_sh[ROWTHR ? PR : 1];
    __shared__ int item_sh[2];
    __shared__ TailExt tail_sh[8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NKS = NKC * 8;
    constexpr int ROWB = NKC * 256;    // bytes per fp16 row
    constexpr int TILEB = NKS * 2048;  // bytes per 64-row wave tile of the fragment-major image
    const int lane16 = lane * 16;
    // LDS byte offset of the lane's A fragment at k-step u of a k chunk (+ chunk * 32768 + block m * 8192)
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
    // the wave's segment of the two candidate arrays as buffer resources (emit_candidates_seg)
    const __amdgpu_buffer_rsrc_t rs_ci = __builtin_amdgcn_make_buffer_rsrc(
        (void*)uniform_ptr(reinterpret_cast<const char*>(a.out_i + seg_base)), 0, a.seg_cap * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_cj = __builtin_amdgcn_make_buffer_rsrc(
        (void*)uniform_ptr(reinterpret_cast<const char*>(a.out_j + seg_base)), 0, a.seg_cap * 4, 0x00020000);
    const int nslice = (a.nsteps + a.slice - 1) / a.slice;
    int cur_panel = -1;
    float nq_max = 0.f;
    float rtmin[4] = {0.f, 0.f, 0.f, 0.f};
    int panel = blockIdx.x % a.npanel;
    for (;;) {
        // ---- next work item: a slice of this workgroup's panel, else of the panel with the most left
        __syncthreads();
        if (wave == 0) {
            int p = panel, s = 0;
            // a launch whose candidate list has overflowed is lost (the host reruns it with larger buffers): stop
            // taking work
#ifndef VSC_NO_LOST_CHECK
            const bool lost = __hip_atomic_load(a.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
#else
            const bool lost = false;
#endif
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
            // query panel -> LDS [k chunk][row][slot ^ (row & 15)]: the swizzle is applied to the SOURCE address,
            // an LDS-DMA instruction writes its 64 x 16 bytes to consecutive LDS addresses
            const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(
                (void*)uniform_ptr(reinterpret_cast<const char*>(a.Q) + (int64_t)panel * PR * ROWB), 0, PR * ROWB, 0x00020000);
#pragma unroll
            for (int n = 0; n < NKC * 4; ++n) {
                const int p = n * 512 + tid;
                const int kc = p >> 11, row = (p >> 4) & 127, slot = p & 15;
                const int c = kc * 16 + (slot ^ (row & 15));
                dma16(qrs, row * ROWB + c * 16, smem + (n * 512 + wave * 64) * 16);
            }
            // largest row norm of the panel (rows past the batch do not count); k-NN: the row thresholds
            const bool in_batch = tid < PR && panel * PR + tid < a.nq;
            float nv = in_batch ? a.qn[(int64_t)panel * PR + tid] : 0.f;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) nv = fmaxf(nv, __shfl_xor(nv, off));
            if (lane == 0) qn_max_w[wave] = nv;
            if (ROWTHR && tid < PR) rt_sh[tid] = in_batch ? a.row_thr[(int64_t)panel * PR + tid] : INFINITY;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces have landed ...
            __syncthreads();                                  // ... and so have everybody else's
            nq_max = fmaxf(qn_max_w[0], qn_max_w[1]);
            if (ROWTHR) {
                // smallest row threshold of each 32-row block
                float v0 = rt_sh[lane], v1 = rt_sh[64 + lane];
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) {
                    v0 = fminf(v0, __shfl_xor(v0, off));
                    v1 = fminf(v1, __shfl_xor(v1, off));
                }
                // (wave-uniform: kept in scalar registers, the vector file is full)
                rtmin[0] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v0), 0));
                rtmin[1] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v0), 32));
                rtmin[2] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v1), 0));
                rtmin[3] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v1), 32));
            }
            cur_panel = panel;
        }
        // ---- this wave's stream: tiles (cs * 8 + wave), cs = cs0 .. cs1-1, TILEB contiguous bytes each.
        // Buffer resource = the item's slice of the fragment-major image (reads past its end return 0).
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)uniform_ptr(reinterpret_cast<const char*>(a.Rf) + (int64_t)cs0 * 8 * TILEB), 0, (cs1 - cs0) * 8 * TILEB,
            0x00020000);
        int so_tile = wave * TILEB;
        f16x8 ring[PF][2];
#pragma unroll
        for (int dd = 0; dd < PF - 1; ++dd) {
            // (NKS >= 8 > PF - 1: the ring never reaches into the next tile here)
            ring[dd][0] = bload(rs, lane16, so_tile + dd * 2048);
            ring[dd][1] = bload(rs, lane16 + 1024, so_tile + dd * 2048);
        }
        f16x8 afr[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) afr[m] = *reinterpret_cast<const f16x8*>(smem + abase[0] + m * 8192);
        for (int cs = cs0; cs < cs1; ++cs) {
            const int col0 = cs * CSW + wave * 64;
            // norm bounds of the lane's two columns (the image is padded to whole col-steps)
            const float rn0 = a.rn[col0 + (lane & 31)], rn1 = a.rn[col0 + 32 + (lane & 31)];
            f32x16 acc[4][2];
            tile_mma<NKC>(smem, abase, afr, ring, rs, so_tile, so_tile + 8 * TILEB, lane16, acc);
            so_tile += 8 * TILEB;
            const float eps[2] = {(a.c1 * nq_max * rn0 + a.c2 * (nq_max + rn0) + a.c3) * 1.001f,
                                  (a.c1 * nq_max * rn1 + a.c2 * (nq_max + rn1) + a.c3) * 1.001f};
            const bool all[2] = {!(eps[0] < INFINITY), !(eps[1] < INFINITY)};  // also catches NaN (inf * 0)
            const float thr[2] = {candidate_edge(radius, eps[0]), candidate_edge(radius, eps[1])};
            // candidates are rare: one max per 32x32 block first, one compare for the whole wave tile
            float bm[4][2];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    float x = fmaxf(fmaxf(acc[m][n][0], acc[m][n][1]), acc[m][n][2]);
#pragma unroll
                    for (int r = 3; r < 15; r += 2) x = fmaxf(fmaxf(x, acc[m][n][r]), acc[m][n][r + 1]);
                    bm[m][n] = fmaxf(x, acc[m][n][15]);
                }
            bool any_blk = all[0] || all[1];
            if (ROWTHR) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    any_blk |= bm[m][0] >= candidate_edge(rtmin[m], eps[0]) || bm[m][1] >= candidate_edge(rtmin[m], eps[1]);
            } else {
                const float x0 = fmaxf(fmaxf(bm[0][0], bm[1][0]), fmaxf(bm[2][0], bm[3][0]));
                const float x1 = fmaxf(fmaxf(bm[0][1], bm[1][1]), fmaxf(bm[2][1], bm[3][1]));
                any_blk |= x0 > thr[0] || x1 > thr[1];
            }
            if (__any(any_blk)) {
                const bool interior = panel * PR + PR <= a.nq && col0 + 64 <= a.nr;
                if (VSC_F16P_EMIT == 3 && !ROWTHR && interior && !all[0] && !all[1] && count + 8192 <= a.seg_cap)
                    emit_candidates_seg(a, thr, panel * PR, col0, acc, bm, rs_ci, rs_cj, count);
                else if (VSC_F16P_EMIT == 2 || (VSC_F16P_EMIT == 0 && !ROWTHR))
                    emit_candidates_blk<ROWTHR>(a, all, thr, eps, rt_sh, rtmin, panel * PR, col0, interior, acc, bm,
                                                seg_base, count, &tail_sh[wave]);
                else
                    emit_candidates<ROWTHR>(a, all, thr, eps, rt_sh, rtmin, panel * PR, col0, interior, acc, bm,
                                            seg_base, count, &tail_sh[wave]);
            }
        }
    }
    tail_close(a.tail_base, a.tail_shift, a.tail_fill, lane, &tail_sh[wave]);
    if (lane == 0) a.seg_count[seg] = count;
}

// Work split of one launch: slices (in col-steps) such that every workgroup sees >= ~16 items when the
// problem allows it (the stealing balances to within one item), between 4 and 64 col-steps each
void sim_f16p_plan(int64_t nq, int64_t nr, int* npanel, int* nsteps, int* slice, int* grid) {
    using namespace f16p;
    const int64_t P = (nq + PR - 1) / PR, S = (nr + CSW - 1) / CSW;
    int64_t sl = (S * P) / (256 * 16);
    sl = std::max<int64_t>(4, std::min<int64_t>(64, sl));
    sl = std::min(sl, std::max<int64_t>(S, 1));
    const int64_t items = P * ((S + sl - 1) / sl);
    *npanel = (int)P;
    *nsteps = (int)S;
    *slice = (int)sl;
    *grid = (int)std::max<int64_t>(1, std::min<int64_t>(256, items));
}

template <int NKC>
static int launch_nkc(const SimF16PArgs& a, int grid, hipStream_t stream) {
    const int lds = NKC * 32768;
    static PerDeviceOnce once;
    if (once.first()) {
        VSC_HIP(hipFuncSetAttribute((const void*)sim_f16p_kernel<NKC, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        VSC_HIP(hipFuncSetAttribute((const void*)sim_f16p_kernel<NKC, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        once.commit();
    }
    if (a.row_thr)
        hipLaunchKernelGGL((sim_f16p_kernel<NKC, true>), dim3((unsigned)grid), dim3(512), lds, stream, a);
    else
        hipLaunchKernelGGL((sim_f16p_kernel<NKC, false>), dim3((unsigned)grid), dim3(512), lds, stream, a);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

int launch_sim_f16p(const SimF16PArgs& a, int grid, hipStream_t stream) {
    if (grid <= 0 || a.npanel <= 0 || a.nsteps <= 0) {
        // nothing to search: the caller's exact stage must see empty segments, not stale fill levels
        if (grid > 0) VSC_HIP(hipMemsetAsync(a.seg_count, 0, (size_t)grid * 8 * sizeof(int), stream));
        return VSC_OK;
    }
    VSC_HIP(hipMemsetAsync(a.next_slice, 0, (size_t)a.npanel * sizeof(int), stream));
    switch (a.dpadh) {
        case 128: return launch_nkc<1>(a, grid, stream);
        case 256: return launch_nkc<2>(a, grid, stream);
        case 384: return launch_nkc<3>(a, grid, stream);
        case 512: return launch_nkc<4>(a, grid, stream);
    }
    set_error("sim_f16p: dpadh %d is not one of 128, 256, 384, 512", a.dpadh);
    return VSC_ERR_INVALID;
}

}  // namespace vscmi
