// Panel-stationary INT8 pre-filter of the thresholded searches (gfx950), dims <= 1024.
//
// Same contract as sim_f16p.hip -- candidates = a superset of the pairs whose exact fp32 score reaches the
// threshold; vsc/index.py:142-165 semantics are restored by the exact stage (rescore_kernel) -- and the same data
// flow (128-row query panel resident in LDS, reference fragments streamed straight into registers from a
// fragment-major image, 8 waves x (128 rows x 64 columns), work items behind one atomic counter), on
// v_mfma_i32_16x16x64_i8.
//
// Why 16x16x64 and not 32x32x32 (round 4, scripts/ubench/i8_tiles.hip, profiles/r04_i8_tiles_ubench.md): on this
// pool the kernel is POWER-bound, not issue-bound -- the 32x32x32 skeleton keeps the matrix pipe busy 92 % of the
// cycles and the clock drops to 1.50 GHz (MFMA-only loop: 1.74 GHz).  The 16x16x64 form reads and writes a quarter
// of the accumulator registers per instruction (K = 64 per pass over a 16x16 block): the MFMA-only loop holds
// 2.08 GHz (4300 vs 3550 TOP/s), the same skeleton 3100-3200 vs 2900 TOP/s.  Operand bytes per MFMA cycle are
// unchanged (per 64 k: 8 ds_read_b128 + 4 buffer_load_dwordx4 for 32 MFMAs of 16 cycles).
//
// Why this is still exact: the integer accumulators are EXACT dot products of the quantised rows, so the only
// error is quantisation, and quant_i8.hip records per row what was actually lost:
//     x = s (q + e),  E >= ||x - s q||,  N >= ||x||   =>   | x.y - s_x s_y (q_x . q_y) | <= E_x N_y + (N_x + E_x) E_y
// plus c_acc N_x N_y for the rounding of the exact fp32 chain itself.  (Coordinates on which all references agree are
// kept out of the images and enter through per-row thresholds instead: quant_i8.hip, "EXCLUDED coordinates".)  A pair
// whose exact score exceeds the radius therefore has   q_x . q_y  >  (radius - eps_xy) / (s_x s_y);   the kernel tests
// the integer accumulator against the floor of a lower bound of that quotient, per reference column (one scale per
// reference ROW, one scale and the largest E / N per query PANEL).  With 8 bits the bound is ~16x looser than the
// fp16 one (eps ~ 0.018 for unit 512-d rows: 4-5x as many candidates), which is why api.hip uses this kernel only
// where hits are sparse.
//
// Two shapes (round 4, profiles/r04_i8_tiles_ubench.md V10).  HALVES = 1: the geometry above.  HALVES = 2 (512-d and below,
// launches with enough panels): a work item holds TWO consecutive 128-row panels (128 KiB of LDS) and a wave tile is
// 256 rows x 32 columns -- the same 128 accumulators, but every reference fragment a wave fetches from the L2 now
// feeds 16 MFMAs instead of 8, and the panel reads (LDS) double instead.  The reference stream is the expensive operand
// on this power-bound kernel (skeleton: no stream +19 %, no panel reads +6 %), the trade measured +5 % on the skeleton.
// The two halves keep their own scale, bound coefficients and thresholds (quantisation stays per 128 rows), so the
// candidate set is the same as HALVES = 1's, bit for bit.
//
// Epilogue.  C layout of the 16x16 MFMA: lane l holds column l & 15, rows 4 (l >> 4) + r of the block, r = 0..3.  A
// wave tile is 8 row blocks x 4 column blocks; lane l owns the threshold arithmetic of ONE column, col0 + l (one
// coalesced 16-byte meta load per lane and tile), and the four columns it holds accumulators of get their integer
// threshold by ds_bpermute.  The common path is 64 v_max3 (one maximum per lane and column block) + 4 compares.
// Per-row thresholds (k-NN, excluded coordinates): the rows of a launch arrive sorted by threshold (sortpairs.hip);
// the common path tests against the PANEL's smallest threshold, and only a column block that passes is re-tested
// per 16-row block against that block's smallest threshold (integer compares only, the arithmetic of 8 thresholds
// per column block on the rare path instead of 8 per tile on the common one).
#include "kernels.h"

namespace vscmi {

typedef int i32x4 __attribute__((ext_vector_type(4)));

namespace i8p {

#ifndef VSC_I8P_ABLATE  // timing experiments only (scripts/experiments): 1 = no candidate emission, 2 = no block maxima
#define VSC_I8P_ABLATE 0  // and no emission either -- both give WRONG results
#endif

constexpr int PR = F16P_PANEL_ROWS;  // 128
static_assert(F16P_COL_STEP == 512, "a col-step is 8 waves x 64 columns (or 2 steps of 8 x 32)");
#ifndef VSC_I8P_PF
#define VSC_I8P_PF 2
#endif
// Register ring of the reference stream in 64-k steps (PF - 1 in flight, 4 KiB each per wave).  PF must divide the
// steps of a tile (4 per 256 bytes of row): the ring is carried from tile to tile in REGISTERS that asynchronous loads
// are still writing, so the slot a step lands in must not depend on the tile -- otherwise the compiler rotates the
// slots with v_mov at the loop's back edge and copies registers whose loads have not landed (seen with PF = 3).
#ifndef VSC_I8P_PF2
#define VSC_I8P_PF2 4  // ... of the paired shape (2 KiB per step and wave there; ring of 2 / 4: 1366 / 1341 ms of int8 kernel per configs[3] step)
#endif
// (256-d paired: the deeper ring drives the compiler into hundreds of spills -- and a spilled ring register is read
// before its load has landed; every variant's "VGPRs Spill" is checked after a change here)
template <int HALVES, int NKC> struct RingDepth { static constexpr int v = HALVES == 2 && NKC == 2 ? VSC_I8P_PF2 : VSC_I8P_PF; };
static_assert((VSC_I8P_PF == 2 || VSC_I8P_PF == 4) && (VSC_I8P_PF2 == 2 || VSC_I8P_PF2 == 4),
              "the ring must divide the 4 steps of a 256-byte chunk");
constexpr int HB = 8;           // 16-row blocks of a 128-row panel (= of a half of a paired wave tile)
constexpr int AW = 4;           // A operands in registers: a rolling window of AW row blocks (the next AW blocks of the m-major order)

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, int voff, char* lds) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
}
// The reference stream with hand-placed waits (round 3: the loads are opaque to the compiler's s_waitcnt insertion,
// which otherwise drains the whole stream with vmcnt(0) at the start of every tile -- it loses count of the
// outstanding loads across the emission branches).  A load's destination must not be touched before ring_wait has
// named it (the "+v" ties keep every use behind the wait).
// s_nop 4: a VMEM instruction that reads an SGPR written by a VALU instruction (v_readlane / v_readfirstlane: the
// compiler keeps spilled scalars in VGPR lanes and reloads `soff` with v_readlane right in front of the load) needs 5
// wait states, and the compiler's hazard recognizer does not look inside inline assembly.  Found in round 4 with a ring
// of 4 on the paired shape (three prologue loads, the 2nd and 3rd with a freshly reloaded soffset): 16 % of the runs of
// one top-K case lost a hit in an item's first tile (scripts/experiments/dbg_pair_ring.py); 0 of 200 with the nops.
template <int OFF>
__device__ __forceinline__ void bload_asm(i32x4& dst, __amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4"
                 : "=&v"(dst) : "v"(voff), "s"(rs), "s"(soff), "n"(OFF) : "memory");
}
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gload_asm(f32x4v& dst, const float4* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void meta_wait(f32x4v& m) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(m) : "n"(N));
}
template <int N>
__device__ __forceinline__ void ring_wait(i32x4 (&b)[4]) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N));
}
template <int N>
__device__ __forceinline__ void ring_wait(i32x4 (&b)[2]) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(b[0]), "+v"(b[1]) : "n"(N));
}
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ float uniform_f(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}
__device__ __forceinline__ float lane_f(float x, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l));
}

// One output tile (128 panel rows x the wave's 64 columns, K = NKC x 256) in steps of 64 k: the A operand of row
// block m (rows 16 m + (lane & 15), k bytes 16 (lane >> 4) .. + 15 of the step) comes from the LDS panel through a
// rolling window of AW registers quadruples -- the operand of the block AW places further on in the (step, m) order is
// loaded into a slot right behind the slot's last use, 4 AW MFMAs ahead of its own --; the B operands of a step are
// 4 KiB of consecutive image (column block n: 1 KiB at n * 1024), refilled PF - 1 steps ahead; the stream continues
// into the wave's next tile at `so_next`.  Straight-line code, every LDS address = base register + immediate; issue
// order pinned.
template <int NKC, int MB, int CB, int PF>
__device__ __forceinline__ void tile_mma(const char* smem, const int (&abase)[4], i32x4 (&a)[AW], i32x4 (&ring)[PF][CB],
                                         __amdgpu_buffer_rsrc_t rs, int so_tile, int so_next, int lane16,
                                         i32x4 (&acc)[MB][CB]) {
    const i32x4 zero = {0, 0, 0, 0};
    constexpr int NK4 = NKC * 4;
    constexpr int KCB = MB * 4096;  // bytes per 256-byte k chunk of the LDS panel (MB * 16 rows x 256 B)
#pragma unroll
    for (int k4 = 0; k4 < NK4; ++k4) {
        const int t = k4 + PF - 1;  // step that goes into the ring slot freed by step k4 - 1
        const int so = (t < NK4) ? so_tile + t * 4096 : so_next + (t - NK4) * 4096;
        // all four fragments of this step have landed once at most the 4 (PF - 2) loads of the younger steps are out
        ring_wait<CB * (PF - 2)>(ring[k4 % PF]);
#pragma unroll
        for (int m = 0; m < MB; ++m) {
#pragma unroll
            for (int n = 0; n < CB; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[m % AW], ring[k4 % PF][n], k4 == 0 ? zero : acc[m][n], 0, 0, 0);
            {
                // the block AW places on (the next tile starts at step 0 again)
                const int kn = (k4 + (m + AW) / MB) % NK4, mn = (m + AW) % MB;
                a[m % AW] = *reinterpret_cast<const i32x4*>(smem + (kn >> 2) * KCB + abase[kn & 3] + mn * 4096);
            }
            if (m == 0) bload_asm<0>(ring[(k4 + PF - 1) % PF][0], rs, lane16, so);
            if (m == 1) bload_asm<1024>(ring[(k4 + PF - 1) % PF][1], rs, lane16, so);
            if (CB > 2) {
                if (m == 2) bload_asm<2048>(ring[(k4 + PF - 1) % PF][2 % CB], rs, lane16, so);
                if (m == 3) bload_asm<3072>(ring[(k4 + PF - 1) % PF][3 % CB], rs, lane16, so);
            }
        }
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, CB, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
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
// integer accumulator > t  <=>  accumulator > floor(t); out-of-range thresholds saturate to never / always (|acc| <=
// 1024 x 127^2 < 2^25), and so does NaN (fmaxf returns its other operand): a NaN threshold passes everything
__device__ __forceinline__ int floor_sat(float t) { return (int)floorf(fminf(fmaxf(t, -2.1e9f), 2.1e9f)); }
constexpr int PASS_ALL = -2100000000;

// The integer threshold of one reference column against an exact threshold t: strict for the radius search (floor of
// a lower bound of the quotient), non-strict for row thresholds (accumulator >= T  <=  accumulator > floor(T) - 1, so
// that pairs tied with a row's threshold survive).  eps = +inf marks a column whose arithmetic says nothing -- +inf /
// NaN bounds (unrepresentable rows), scale products outside the range where their inverse is a normal number -- and
// passes everything (t = +inf, a row past the batch, wins over it: nothing of such a row is ever a candidate).
template <bool ROWTHR>
__device__ __forceinline__ int column_threshold(float t, float eps, float inv) {
    if (!(eps < INFINITY)) return t == INFINITY ? 2100000000 : PASS_ALL;
    return floor_sat(quotient_low(candidate_edge(t, eps), inv)) - (ROWTHR ? 1 : 0);
}

__device__ __forceinline__ int lane_now() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
__device__ __forceinline__ int bperm(int src_lane, int v) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }

// Candidates of one wave tile -> the wave's private segment (no atomics; the wave's chunk of the shared tail when the
// segment is full).  tc[n] = integer threshold of the lane's column of column block n at the tile's common threshold
// (the radius; the panel's smallest row threshold).  ROWTHR: a column block that passes is re-tested per 16-row
// block against rt16[m], the block's smallest row threshold, with the column's own (eps, inv) fetched from the lane
// that owns it.  General form: edge tiles, full segments, pass-everything columns.
// (M0: the first 16-row block of the half this call is about -- its HB blocks share tc / cm / eps_own / inv_own)
template <bool ROWTHR, int MB, int CB, int M0>
__device__ __forceinline__ void emit_candidates(const SimI8PArgs& a, const int (&tc)[CB], const int (&cm)[CB],
                                                const float (&rt16)[MB], float eps_own, float inv_own, int row0, int col0,
                                                bool interior, const i32x4 (&acc)[MB][CB], int64_t seg_base, int& count,
                                                TailExt* ext) {
#pragma unroll
    for (int n = 0; n < CB; ++n) {
        if (!__any(cm[n] > tc[n])) continue;
        float eps_c = 0.f, inv_c = 0.f;
        if (ROWTHR) {
            const int src = n * 16 + (lane_now() & 15);
            eps_c = __builtin_bit_cast(float, bperm(src, __builtin_bit_cast(int, eps_own)));
            inv_c = __builtin_bit_cast(float, bperm(src, __builtin_bit_cast(int, inv_own)));
        }
#pragma unroll
        for (int m = M0; m < M0 + HB; ++m) {
            // a block is tested against the tile's common threshold first: a block's own threshold is never below it
            // (rt16[m] >= the panel's smallest row threshold, and column_threshold is monotone in t), so the
            // arithmetic of the block's threshold is spent only on blocks that can hold a candidate (PMC, round 4: with
            // it on every block of a passing column block the kernel issued 316 VALU per tile against the skeleton's 67)
            const int bmx = max(max(max(acc[m][n][0], acc[m][n][1]), acc[m][n][2]), acc[m][n][3]);
            if (!__any(bmx > tc[n])) continue;
            const int tb = ROWTHR ? column_threshold<true>(rt16[m], eps_c, inv_c) : tc[n];
            if (ROWTHR && !__any(bmx > tb)) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool hit = acc[m][n][r] > tb;
                const unsigned long long hits = __ballot(hit);
                if (hits == 0ull) continue;
                const int ln = lane_now();
                const int i = row0 + m * 16 + 4 * (ln >> 4) + r;
                const int j = col0 + n * 16 + (ln & 15);
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
}

// The fast path: interior tile, room for a whole tile (8192 entries) left in the wave's segment: position = count +
// (lanes of this register's ballot below me), two buffer stores with a 32-bit offset.
template <bool ROWTHR, int MB, int CB, int M0>
__device__ __forceinline__ void emit_candidates_seg(const SimI8PArgs& a, const int (&tc)[CB], const int (&cm)[CB],
                                                    const float (&rt16)[MB], float eps_own, float inv_own, int row0,
                                                    int col0, const i32x4 (&acc)[MB][CB], __amdgpu_buffer_rsrc_t rs_i,
                                                    __amdgpu_buffer_rsrc_t rs_j, int& count) {
#pragma unroll
    for (int n = 0; n < CB; ++n) {
        if (!__any(cm[n] > tc[n])) continue;
        const int ln = lane_now();  // (live inside the column block only: the register file is full)
        const int pbase = row0 + 4 * (ln >> 4);
        const int j = col0 + n * 16 + (ln & 15);
        float eps_c = 0.f, inv_c = 0.f;
        if (ROWTHR) {
            const int src = n * 16 + (ln & 15);
            eps_c = __builtin_bit_cast(float, bperm(src, __builtin_bit_cast(int, eps_own)));
            inv_c = __builtin_bit_cast(float, bperm(src, __builtin_bit_cast(int, inv_own)));
        }
#pragma unroll
        for (int m = M0; m < M0 + HB; ++m) {
            // a block is tested against the tile's common threshold first: a block's own threshold is never below it
            // (rt16[m] >= the panel's smallest row threshold, and column_threshold is monotone in t), so the
            // arithmetic of the block's threshold is spent only on blocks that can hold a candidate (PMC, round 4: with
            // it on every block of a passing column block the kernel issued 316 VALU per tile against the skeleton's 67)
            const int bmx = max(max(max(acc[m][n][0], acc[m][n][1]), acc[m][n][2]), acc[m][n][3]);
            if (!__any(bmx > tc[n])) continue;
            const int tb = ROWTHR ? column_threshold<true>(rt16[m], eps_c, inv_c) : tc[n];
            if (ROWTHR && !__any(bmx > tb)) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool cand = acc[m][n][r] > tb;
                const unsigned long long hits = __ballot(cand);
                if (hits == 0ull) continue;
                if (cand) {
                    const int off = (count + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(hits >> 32),
                                                                            __builtin_amdgcn_mbcnt_lo((unsigned)hits, 0u)))
                                    << 2;
                    __builtin_amdgcn_raw_buffer_store_b32(a.i0 + pbase + m * 16 + r, rs_i, off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(j, rs_j, off, 0, 0);
                }
                count += __popcll(hits);
            }
        }
    }
}

}  // namespace i8p

template <int NKC, bool ROWTHR, int HALVES>
__global__ __launch_bounds__(512) void sim_i8p_kernel(SimI8PArgs a) {
    using namespace i8p;
    constexpr int MB = HB * HALVES, CB = 4 / HALVES;  // 16-row / 16-column blocks of a wave tile: 8 x 4 or 16 x 2
    constexpr int PRW = PR * HALVES;                  // rows of a work item's panel (HALVES consecutive 128-row panels)
    constexpr int WCOLS = 16 * CB;                    // columns of a wave tile; 8 waves side by side = one step
    constexpr int TPS = 8 / HALVES;                   // 64-column image tiles per step
    constexpr int PF = RingDepth<HALVES, NKC>::v;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // the panel: NKC x 32 KiB x HALVES
    __shared__ float rt_sh[ROWTHR ? PRW : 1];
    __shared__ int item_sh[2];
    __shared__ TailExt tail_sh[8];
#if VSC_I8P_ABLATE
    __shared__ volatile int ablate_sink;
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NK4 = NKC * 4;
    constexpr int ROWB = NKC * 256;    // bytes per int8 row
    constexpr int TILEB = NK4 * 4096;  // bytes per 64-row wave tile of the fragment-major image
    // byte offset of this lane's piece inside a 4 KiB step of an image tile (paired shape: the odd waves take the
    // tile's column blocks 2 and 3)
    const int lane16 = lane * 16 + (HALVES == 2 ? (wave & 1) * 2048 : 0);
    // A operand of row block m at step k4: row 16 m + (lane & 15), 16-byte piece 4 (k4 & 3) + (lane >> 4) of chunk
    // k4 >> 2; LDS image [k chunk of 256 B][row][piece ^ (row & 15)] (conflict-free: the 16 lanes of a ds_read_b128
    // phase hold 16 different pieces ^ rows)
    int abase[4];
    {
        const int kp = lane >> 4, r15 = lane & 15;
#pragma unroll
        for (int u = 0; u < 4; ++u) abase[u] = r15 * 256 + ((((4 * u) | kp) ^ r15) << 4);
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
    // work items: (slice of a.slice 512-column col-steps, panel of PRW rows); a.npanel counts 128-row panels
    const int npan = (a.npanel + HALVES - 1) / HALVES;
    const int nslice = (a.nsteps + a.slice - 1) / a.slice;
    int cur_panel = -1;
    // per 128-row panel (wave-uniform): 1 / s_q and the coefficients of N'_r, E_r and N_r in eps
    float inv_sq[HALVES], coef_k[HALVES], coef_e[HALVES], coef_n[HALVES];
    float rt16[MB];         // smallest row threshold per 16-row block
    float rt_half[HALVES];  // ... and per 128-row panel
#pragma unroll
    for (int h = 0; h < HALVES; ++h) { inv_sq[h] = 1.0f; coef_k[h] = coef_e[h] = coef_n[h] = 0.0f; rt_half[h] = 0.0f; }
#pragma unroll
    for (int m = 0; m < MB; ++m) rt16[m] = 0.0f;
    int panel = blockIdx.x % npan;
    for (;;) {
        // ---- next work item: a slice of this workgroup's panel, else of the panel with the most left
        __syncthreads();
        if (wave == 0) {
            int p = panel, s = 0;
            // a launch whose candidate list has overflowed is lost (the host reruns the batch on the fp16 kernel or
            // with larger buffers): stop taking work instead of pushing billions of candidates through the tail's
            // one atomic counter (an 8-bit bound that is too loose for the data can pass most of the matrix)
            const bool lost = __hip_atomic_load(a.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            if (a.order == 1 && !lost) {
                // slice-major: items (slice, panel) in one global order -- every workgroup of the chip is inside
                // the same few MB of the reference image, each XCD fetches a slice once; the price is a panel load
                // (64 KiB per 128 rows at 512-d) per item
                int t = 0;
                if (lane == 0) t = atomicAdd(&a.next_slice[a.npanel], 1);
                t = __shfl(t, 0);
                if (t >= nslice * npan) { p = -1; } else { s = t / npan; p = t - s * npan; }
            } else
            for (;;) {
                if (lost) { p = -1; break; }
                if (lane == 0) s = atomicAdd(&a.next_slice[p], 1);
                s = __shfl(s, 0);
                if (s < nslice) break;
                int best = 0x7fffffff, bp = 0x7fffffff;
                for (int q0 = 0; q0 < npan; q0 += 64) {
                    const int q = q0 + lane;
                    const int pp = (q + (int)blockIdx.x) % npan;  // ties: nearest after the workgroup's own
                    const int v = q < npan ? __hip_atomic_load(&a.next_slice[pp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
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
        // steps of 8 x WCOLS columns (HALVES per 512-column col-step)
        const int cs0 = sl * a.slice * HALVES, cs1 = min(a.nsteps, (sl + 1) * a.slice) * HALVES;
        if (panel != cur_panel) {
            __syncthreads();  // (item_sh is read; nobody reads the old panel any more)
            // query panel -> LDS [k chunk of 256 B][row][slot ^ (row & 15)]: the swizzle is applied to the SOURCE
            // address, an LDS-DMA instruction writes its 64 x 16 bytes to consecutive LDS addresses.  (Paired shape:
            // the host quantises an even number of 128-row panels, the last one possibly all rows past the batch.)
            const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(
                (void*)uniform_ptr(reinterpret_cast<const char*>(a.Q) + (int64_t)panel * PRW * ROWB), 0, PRW * ROWB, 0x00020000);
#pragma unroll
            for (int n = 0; n < NKC * 4 * HALVES; ++n) {
                const int p = n * 512 + tid;
                const int kc = p / (PRW * 16), row = (p >> 4) % PRW, slot = p & 15;
                const int c = kc * 16 + (slot ^ (row & 15));
                dma16(qrs, row * ROWB + c * 16, smem + (n * 512 + wave * 64) * 16);
            }
#pragma unroll
            for (int h = 0; h < HALVES; ++h) {
                const float4 ps = a.pstat[panel * HALVES + h];  // {1 / s_q, max E_q, max N_q, max N'_q}
                inv_sq[h] = uniform_f(ps.x);
                // eps_j = E_q N'_r + (N'_q + E_q) E_r + c_acc N_q N_r   (N' = norm over the coordinates the images hold)
                coef_k[h] = uniform_f(ps.y);
                coef_e[h] = uniform_f(ps.w + ps.y);
                coef_n[h] = uniform_f(a.c_acc * ps.z);
            }
            if (ROWTHR && tid < PRW) rt_sh[tid] = panel * PRW + tid < a.nq ? a.row_thr[(int64_t)panel * PRW + tid] : INFINITY;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces have landed ...
            __syncthreads();                                  // ... and so have everybody else's
            if (ROWTHR) {
#pragma unroll
                for (int g = 0; g < PRW / 64; ++g) {
                    float v = rt_sh[64 * g + lane];
#pragma unroll
                    for (int off = 8; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off));
#pragma unroll
                    for (int m = 0; m < 4; ++m) rt16[4 * g + m] = lane_f(v, 16 * m);
                }
#pragma unroll
                for (int h = 0; h < HALVES; ++h)
                    rt_half[h] = fminf(fminf(fminf(rt16[8 * h], rt16[8 * h + 1]), fminf(rt16[8 * h + 2], rt16[8 * h + 3])),
                                       fminf(fminf(rt16[8 * h + 4], rt16[8 * h + 5]), fminf(rt16[8 * h + 6], rt16[8 * h + 7])));
                // (NaN row thresholds: fminf drops them unless a whole block holds nothing else -- that block then passes
                // everything on the re-test; the exact stage keeps nothing of a row whose threshold is NaN either way)
            }
            cur_panel = panel;
        }
        // ---- this wave's stream: image tiles cs * TPS + wave / HALVES, cs = cs0 .. cs1-1, TILEB contiguous bytes each
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)uniform_ptr(reinterpret_cast<const char*>(a.Rf) + (int64_t)cs0 * TPS * TILEB), 0, (cs1 - cs0) * TPS * TILEB,
            0x00020000);
        int so_tile = (wave / HALVES) * TILEB;
        i32x4 ring[PF][CB];
#pragma unroll
        for (int dd = 0; dd < PF - 1; ++dd) {
            bload_asm<0>(ring[dd][0], rs, lane16, so_tile + dd * 4096);
            bload_asm<1024>(ring[dd][1], rs, lane16, so_tile + dd * 4096);
            if (CB > 2) {
                bload_asm<2048>(ring[dd][2 % CB], rs, lane16, so_tile + dd * 4096);
                bload_asm<3072>(ring[dd][3 % CB], rs, lane16, so_tile + dd * 4096);
            }
        }
        i32x4 afr[AW];
#pragma unroll
        for (int m = 0; m < AW; ++m) afr[m] = *reinterpret_cast<const i32x4*>(smem + abase[0] + m * 4096);
        for (int cs = cs0; cs < cs1; ++cs) {
            const int col0 = cs * (8 * WCOLS) + wave * WCOLS;
            // {1 / s_r, E_r, N_r, N'_r} of the column this lane does the threshold arithmetic of (the table is padded to
            // whole col-steps): issued behind the stream loads of the previous tile, ahead of this tile's -- after the K
            // loop only the CB (PF - 1) stream loads of the next tile are younger
            f32x4v mt;
            gload_asm(mt, a.rmeta + col0 + (lane & (WCOLS - 1)));
            i32x4 acc[MB][CB];
            tile_mma<NKC, MB, CB, PF>(smem, abase, afr, ring, rs, so_tile, so_tile + TPS * TILEB, lane16, acc);
            so_tile += TPS * TILEB;
            meta_wait<CB * (PF - 1)>(mt);
            // this lane's column: eps, 1 / (s_q s_r) per 128-row panel; +inf eps = "pass everything" (see column_threshold)
            float eps_own[HALVES], inv_own[HALVES];
            int tc[HALVES][CB], cm[HALVES][CB];
            bool any_col = false;
#pragma unroll
            for (int h = 0; h < HALVES; ++h) {
                eps_own[h] = (coef_k[h] * mt.w + coef_e[h] * mt.y + coef_n[h] * mt.z) * 1.001f;
                inv_own[h] = inv_sq[h] * mt.x;
                if (!(eps_own[h] < INFINITY) || !(inv_own[h] >= 1e-30f && inv_own[h] < INFINITY)) eps_own[h] = INFINITY;
                const int t_own = column_threshold<ROWTHR>(ROWTHR ? rt_half[h] : radius, eps_own[h], inv_own[h]);
                // ... and the thresholds of the columns this lane holds accumulators of
#if VSC_I8P_ABLATE == 2
#pragma unroll
                for (int n = 0; n < CB; ++n) tc[h][n] = t_own;
#else
#pragma unroll
                for (int n = 0; n < CB; ++n) tc[h][n] = bperm(n * 16 + (lane & 15), t_own);
#endif
            }
#if VSC_I8P_ABLATE == 2
#pragma unroll
            for (int h = 0; h < HALVES; ++h)
#pragma unroll
                for (int n = 0; n < CB; ++n) {
                    // no maxima: one accumulator of every block stands in for the block (keeps the MFMAs alive)
                    int y = acc[HB * h][n][0];
                    for (int m = 1; m < HB; ++m) y |= acc[HB * h + m][n][m & 3];
                    cm[h][n] = y;
                    any_col |= y == 0x12345678;
                }
#else
            // the lane's 32 accumulators of each (128-row panel, column block) (8 row blocks x 4 registers): 16 v_max3
            // each, the four chains advanced together (a chain of dependent v_max3 issues every ~8 cycles, four
            // independent ones keep the VALU fed).  Measured (round 4, -DVSC_I8P_ABLATE): these maxima + the threshold
            // distribution cost 9 % of the kernel, the emission of this workload's candidates 1 %; starting the second
            // wave of every SIMD 2000-4000 cycles late so that the two waves' epilogues do not coincide: +-0
            {
                int x[HALVES][CB];
#pragma unroll
                for (int h = 0; h < HALVES; ++h)
#pragma unroll
                    for (int n = 0; n < CB; ++n) x[h][n] = max(max(acc[HB * h][n][0], acc[HB * h][n][1]), acc[HB * h][n][2]);
#pragma unroll
                for (int k = 3; k < 31; k += 2)
#pragma unroll
                    for (int h = 0; h < HALVES; ++h)
#pragma unroll
                        for (int n = 0; n < CB; ++n)
                            x[h][n] = max(max(x[h][n], acc[HB * h + (k >> 2)][n][k & 3]), acc[HB * h + ((k + 1) >> 2)][n][(k + 1) & 3]);
#pragma unroll
                for (int h = 0; h < HALVES; ++h)
#pragma unroll
                    for (int n = 0; n < CB; ++n) {
                        cm[h][n] = max(x[h][n], acc[HB * h + 7][n][3]);
                        any_col |= cm[h][n] > tc[h][n];
                    }
            }
#endif
#if VSC_I8P_ABLATE
            if (__any(any_col) && lane == 0) ablate_sink = col0;  // (keeps the accumulators alive)
#endif
            if (VSC_I8P_ABLATE == 0 && __any(any_col)) {
                const bool interior = panel * PRW + PRW <= a.nq && col0 + WCOLS <= a.nr;
                if (interior && count + 8192 <= a.seg_cap) {
                    emit_candidates_seg<ROWTHR, MB, CB, 0>(a, tc[0], cm[0], rt16, eps_own[0], inv_own[0], panel * PRW, col0, acc,
                                                           rs_ci, rs_cj, count);
                    if (HALVES == 2)
                        emit_candidates_seg<ROWTHR, MB, CB, HB*(HALVES - 1)>(a, tc[HALVES - 1], cm[HALVES - 1], rt16, eps_own[HALVES - 1],
                                                                              inv_own[HALVES - 1], panel * PRW, col0, acc, rs_ci, rs_cj,
                                                                              count);
                } else {
                    emit_candidates<ROWTHR, MB, CB, 0>(a, tc[0], cm[0], rt16, eps_own[0], inv_own[0], panel * PRW, col0, interior, acc,
                                                       seg_base, count, &tail_sh[wave]);
                    if (HALVES == 2)
                        emit_candidates<ROWTHR, MB, CB, HB*(HALVES - 1)>(a, tc[HALVES - 1], cm[HALVES - 1], rt16, eps_own[HALVES - 1],
                                                                          inv_own[HALVES - 1], panel * PRW, col0, interior, acc, seg_base,
                                                                          count, &tail_sh[wave]);
                }
            }
        }
        // The stream ran PF - 1 steps past the item's end (out-of-range loads: zeros, no memory traffic) and those loads
        // still target ring registers: they must have landed before the hand-over code reuses the registers.  (Found with
        // the ring of 4 on the paired shape: the item index was computed in registers that late loads then overwrote;
        // with a ring of 2 the barrier at the top of the loop happened to outlast the single step in flight.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    tail_close(a.tail_base, a.tail_shift, a.tail_fill, lane, &tail_sh[wave]);
#if VSC_I8P_ABLATE
    if (ablate_sink == 0x7fffffff) count = -1;
#endif
    if (lane == 0) a.seg_count[seg] = count;
}

template <int NKC, int HALVES>
static int launch_nkc(const SimI8PArgs& a, int grid, hipStream_t stream) {
    const int lds = NKC * 32768 * HALVES;
    static PerDeviceOnce once;
    if (once.first()) {
        VSC_HIP(hipFuncSetAttribute((const void*)sim_i8p_kernel<NKC, false, HALVES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        VSC_HIP(hipFuncSetAttribute((const void*)sim_i8p_kernel<NKC, true, HALVES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        once.commit();
    }
    if (a.row_thr)
        hipLaunchKernelGGL((sim_i8p_kernel<NKC, true, HALVES>), dim3((unsigned)grid), dim3(512), lds, stream, a);
    else
        hipLaunchKernelGGL((sim_i8p_kernel<NKC, false, HALVES>), dim3((unsigned)grid), dim3(512), lds, stream, a);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

// The paired shape needs 2 x NKC x 32 KiB of LDS (512-d and below) and enough items to fill the chip; an odd number of
// 128-row panels leaves half of the last item empty.  api.hip asks before it sizes the quantised image.
bool sim_i8p_pairs(int dpad8, int npanel, int nsteps, int slice, bool force) {
    if (dpad8 > 512 || npanel < 1) return false;
    if (force) return true;  // (tests: every launch the shape can take)
    if (npanel < 2) return false;
    const long long items = (long long)((npanel + 1) / 2) * ((nsteps + slice - 1) / slice);
    return items >= 512 && (npanel % 2 == 0 || npanel >= 25);
}

// (the work split is sim_f16p_plan's: same panel and col-step geometry; a.pair: work items of two panels)
int launch_sim_i8p(const SimI8PArgs& a, int grid, hipStream_t stream) {
    if (grid <= 0 || a.npanel <= 0 || a.nsteps <= 0) {
        // nothing to search: the caller's exact stage must see empty segments, not stale fill levels
        if (grid > 0) VSC_HIP(hipMemsetAsync(a.seg_count, 0, (size_t)grid * 8 * sizeof(int), stream));
        return VSC_OK;
    }
    VSC_HIP(hipMemsetAsync(a.next_slice, 0, ((size_t)a.npanel + 1) * sizeof(int), stream));
    if (a.pair) {
        switch (a.dpad8) {
            case 256: return launch_nkc<1, 2>(a, grid, stream);
            case 512: return launch_nkc<2, 2>(a, grid, stream);
        }
        set_error("sim_i8p: the paired shape takes dpad8 256 or 512, not %d", a.dpad8);
        return VSC_ERR_INVALID;
    }
    switch (a.dpad8) {
        case 256: return launch_nkc<1, 1>(a, grid, stream);
        case 512: return launch_nkc<2, 1>(a, grid, stream);
        case 768: return launch_nkc<3, 1>(a, grid, stream);
        case 1024: return launch_nkc<4, 1>(a, grid, stream);
    }
    set_error("sim_i8p: dpad8 %d is not one of 256, 512, 768, 1024", a.dpad8);
    return VSC_ERR_INVALID;
}

}  // namespace vscmi
