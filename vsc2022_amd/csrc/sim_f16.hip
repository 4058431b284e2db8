// fp16 pre-filter of the thresholded search + exact re-scoring of its candidates (gfx950).
//
// The exact similarity (fp32 fma chain, sim_mfma.hip) runs at the fp32 matrix rate, 1/16 of the fp16
// rate.  Almost every (query row, ref row) pair is far below the search radius, so the bulk of the
// score matrix is first evaluated in fp16 with fp32 accumulation (v_mfma_f32_32x32x16_f16) and a pair
// goes on to the exact stage only if its fp16 score PLUS a rigorous error bound exceeds the radius:
//
//     |fp16 score - exact score| <= c1 * |q| * |r| + c2 * (|q| + |r|)            (api.hip derives c1, c2)
//
// so the candidate set is a superset of the exact hit set and the final result is bit-identical to
// the all-fp32 path (vsc/index.py:142-165 semantics unchanged; tests/test_gpu_prefilter.py).
//
//   sim_f16_kernel   : 256x256 output tile per workgroup step, 8 waves (2 x 4), wave tile 128x64 =
//                      4x2 blocks of 32x32x16 MFMAs; operands by LDS-DMA into an XOR-swizzled image
//                      (2 stages x 64 KiB); persistent workgroups (1 per CU) walk an XCD-aware raster.
//                      MFMA-bound: 2*256*256*dpadh flop per tile.
//   rescore_kernel   : four lanes per candidate run the exact ascending-k fp32 fma chain on the packed fp32
//                      rows, the running sum hopping between them (one 128-byte line per row and load).
//                      HBM/L2-bound: 8*dpad bytes per candidate.
#include <algorithm>

#include "kernels.h"

namespace vscmi {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace f16 {

// Ablation switches for timing studies only (results are WRONG when set): 1 = no barrier / DMA drain,
// 2 = no LDS-DMA in the loop, 4 = no fragment reads.  scripts/ablate_f16.sh builds the variants.
#ifndef VSC_F16_ABLATE
#define VSC_F16_ABLATE 0
#endif

constexpr int BM = 256, BN = 256;
constexpr int BK = 64;                       // fp16 elements per K-tile
constexpr int ROWB = BK * 2;                 // bytes per row per K-tile (128)
constexpr int TILE_BYTES = BM * ROWB;        // 32 KiB per operand
constexpr int STAGE_BYTES = 2 * TILE_BYTES;  // A + B
constexpr int LDS_BYTES = 2 * STAGE_BYTES;   // 128 KiB

// cache policy of the LDS-DMA loads (aux bits of the buffer instruction: 1 = sc0, 2 = nt, 16 = sc1), per operand.
// Measured on one box (fp16 kernel ms per step): default 366; nt on the ref side 438, on the query side 438, on
// both 495 (the panels ARE reused out of L2 by the workgroups of an XCD); sc0 / sc1 366.
#ifndef VSC_F16_AUX_Q
#define VSC_F16_AUX_Q 0
#endif
#ifndef VSC_F16_AUX_R
#define VSC_F16_AUX_R 0
#endif
template <int AUX = 0>
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, char* lds) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff,
                                             soff, 0, AUX);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const void* base, int row_bytes) {
    const uint64_t p = reinterpret_cast<uint64_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
    void* up = reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(up, 0, BM * row_bytes, 0x00020000);
}

// Per-thread constants.  LDS image of one operand K-tile: 128 lines of 256 B; line l holds rows 2l
// and 2l+1 (128 B each = 8 pieces of 16 B); piece c of a row sits at slot c ^ (l & 7) of its half
// line, so the 16 lanes of one ds_read_b128 phase (16 consecutive rows, same piece) cover all 64 banks.
struct TileThread {
    int src_off[4];  // byte offset (inside the tile's rows) of the 4 DMA pieces per operand
    int dst_off[4];  // wave-uniform LDS byte offset of those pieces
    int rdA[4];      // LDS byte offset of A block m at k-step 0 (k-step ks: ^ (ks << 5))
    int rdB[2];
};

__device__ __forceinline__ void tile_thread_init(TileThread& t, int tid, int row_bytes) {
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int p = n * 512 + tid;
        const int line = p >> 4, s = p & 15;
        const int row = 2 * line + (s >> 3);
        const int chunk = (s & 7) ^ (line & 7);
        t.src_off[n] = row * row_bytes + chunk * 16;
        t.dst_off[n] = (n * 512 + __builtin_amdgcn_readfirstlane(wave) * 64) * 16;
    }
    const int hi = lane >> 5;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int row = wr * 128 + m * 32 + (lane & 31);
        const int line = row >> 1;
        t.rdA[m] = line * 256 + ((((row & 1) << 3) | (hi ^ (line & 7))) << 4);
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int row = wc * 64 + n * 32 + (lane & 31);
        const int line = row >> 1;
        t.rdB[n] = line * 256 + ((((row & 1) << 3) | (hi ^ (line & 7))) << 4);
    }
}

struct Frags {
    f16x8 a[4], b[2];
};

__device__ __forceinline__ Frags read_frags(const char* stage, const TileThread& t, int ks) {
    Frags f;
#pragma unroll
    for (int m = 0; m < 4; ++m) f.a[m] = *reinterpret_cast<const f16x8*>(stage + (t.rdA[m] ^ (ks << 5)));
#pragma unroll
    for (int n = 0; n < 2; ++n)
        f.b[n] = *reinterpret_cast<const f16x8*>(stage + TILE_BYTES + (t.rdB[n] ^ (ks << 5)));
    return f;
}

__device__ __forceinline__ void mfma8(const Frags& f, f32x16 (&acc)[4][2]) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[m], f.b[n], acc[m][n], 0, 0, 0);
}

__device__ __forceinline__ void stage_tiles(__amdgpu_buffer_rsrc_t ars, __amdgpu_buffer_rsrc_t brs, int kt,
                                            char* stage, const TileThread& t) {
#pragma unroll
    for (int n = 0; n < 4; ++n) dma16<VSC_F16_AUX_Q>(ars, t.src_off[n], kt * ROWB, stage + t.dst_off[n]);
#pragma unroll
    for (int n = 0; n < 4; ++n) dma16<VSC_F16_AUX_R>(brs, t.src_off[n], kt * ROWB, stage + TILE_BYTES + t.dst_off[n]);
}

// The tile stream.  A workgroup is persistent: the K-tiles of the output tiles it walks form ONE stream
// through the 2-stage LDS ring (K-tile n of the stream lives in stage n & 1), so only the first tile of a
// workgroup pays a load prologue: while the last two K-tiles of tile T are multiplied, the first two of
// tile T+1 are already in flight, and they land during T's epilogue.
struct TileStream {
    const _Float16* q;  // operand panels of the current tile (wave-uniform)
    const _Float16* r;
    Frags cur;          // fragments (K-tile 0, k-step 0) of the current tile
    int sp;             // LDS stage holding K-tile 0 of the current tile
};

__device__ __forceinline__ void stream_begin(TileStream& st, const _Float16* qrows, const _Float16* rrows,
                                             int dpadh, char* smem, const TileThread& t) {
    st.q = qrows;
    st.r = rrows;
    st.sp = 0;
    const __amdgpu_buffer_rsrc_t ars = tile_rsrc(qrows, dpadh * 2), brs = tile_rsrc(rrows, dpadh * 2);
    stage_tiles(ars, brs, 0, smem, t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // K-tile 1 (dpadh >= 128: it always exists): query side only, the ref side is issued by K-tile 0's
    // first k-step like in the steady state
#pragma unroll
    for (int n = 0; n < 4; ++n) dma16<VSC_F16_AUX_Q>(ars, t.src_off[n], ROWB, smem + STAGE_BYTES + t.dst_off[n]);
    st.cur = read_frags(smem, t, 0);
}

// acc += Q[256 rows] . R[256 rows]^T of the stream's current tile; the stream is left on the tile whose
// panels start at (nq, nr).  For the last tile of a workgroup pass the current panels again: the loop
// body has no conditionals (the two K-tiles it then fetches past the end are never read).
//
// Per K-tile (4 k-steps of 8 MFMAs): the barrier sits before the LAST k-step, whose operands are already
// in registers; after it the stage is free, and the DMA of the K-tile two ahead goes into it: the query
// side interleaved with those 8 MFMAs, the ref side with the first 8 of the next K-tile.  Register budget: 2 waves per SIMD => 256 VGPRs per lane, 128 of them accumulators; the A
// fragment of the NEXT k-step is read right after the two MFMAs that consumed the current one (same
// registers), only the two B fragments are double-buffered.
//
// FIRST = first K-tile of an output tile: its first k-step multiplies onto a literal zero (srcC is an
// inline constant), so the 128 accumulators are never cleared with VALU moves.
template <bool FIRST>
__device__ __forceinline__ void stream_ktile(TileStream& st, const _Float16* nq, const _Float16* nr, int dpadh,
                                             int kt, int nkt, Frags& cur, int& sp, char* smem,
                                             const TileThread& t, f32x16 (&acc)[4][2]) {
    const f32x16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    const char* stage = smem + sp * STAGE_BYTES;
    // The 8 LDS-DMA issues a wave owes per K-tile are split so that none of them competes with more than
    // two MFMAs (an issue occupies the wave ~60 cycles, an MFMA the matrix pipe 32): the query side of
    // K-tile kt+2 goes out in this K-tile's last k-step (below), the ref side of K-tile kt+1 here, in the
    // first k-step -- its stage was freed by the previous K-tile's barrier.
    const int k1 = kt + 1;
    const bool in_cur1 = k1 < nkt;
    const __amdgpu_buffer_rsrc_t brs1 = tile_rsrc(in_cur1 ? st.r : nr, dpadh * 2);
    const int soff1 = (in_cur1 ? k1 : k1 - nkt) * ROWB;
    char* wstage1 = smem + (sp ^ 1) * STAGE_BYTES + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
        f16x8 nb[2];
#pragma unroll
        for (int n = 0; n < 2; ++n)
            nb[n] = (VSC_F16_ABLATE & 4) ? cur.b[n]
                                         : *reinterpret_cast<const f16x8*>(stage + TILE_BYTES + (t.rdB[n] ^ ((ks + 1) << 5)));
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.a[m], cur.b[0], (FIRST && ks == 0) ? zero : acc[m][0], 0, 0, 0);
            acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.a[m], cur.b[1], (FIRST && ks == 0) ? zero : acc[m][1], 0, 0, 0);
            if (!(VSC_F16_ABLATE & 4)) cur.a[m] = *reinterpret_cast<const f16x8*>(stage + (t.rdA[m] ^ ((ks + 1) << 5)));
            if (ks == 0 && !(VSC_F16_ABLATE & 2)) dma16<VSC_F16_AUX_R>(brs1, t.src_off[m], soff1, wstage1 + t.dst_off[m]);
        }
        // pin the order: left alone, the scheduler sinks every fragment read to just before its use
        // and exposes the LDS latency once per k-step
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (ks == 0 && !(VSC_F16_ABLATE & 2)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        cur.b[0] = nb[0];
        cur.b[1] = nb[1];
    }
    // last k-step of the K-tile: every fragment of this stage is in registers
    const int k2 = kt + 2;
    const bool in_cur = k2 < nkt;
    const __amdgpu_buffer_rsrc_t ars = tile_rsrc(in_cur ? st.q : nq, dpadh * 2);
    const int soff = (in_cur ? k2 : k2 - nkt) * ROWB;
    const char* nstage = smem + (sp ^ 1) * STAGE_BYTES;
    char* wstage = smem + sp * STAGE_BYTES;
    // The next K-tile of the stream has landed once EVERY wave's own LDS-DMA pieces have: drain this
    // wave's count before the barrier (the compiler does not always do it for the builtin -- without
    // the wait a wave could read pieces another wave's DMA has not delivered yet).
    if (!(VSC_F16_ABLATE & 1)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // ... and nobody reads this stage any more
    }
    f16x8 nb[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
        nb[n] = (VSC_F16_ABLATE & 4) ? cur.b[n] : *reinterpret_cast<const f16x8*>(nstage + TILE_BYTES + t.rdB[n]);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.a[m], cur.b[0], acc[m][0], 0, 0, 0);
        acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.a[m], cur.b[1], acc[m][1], 0, 0, 0);
        if (!(VSC_F16_ABLATE & 4)) cur.a[m] = *reinterpret_cast<const f16x8*>(nstage + t.rdA[m]);
        if (!(VSC_F16_ABLATE & 2)) dma16<VSC_F16_AUX_Q>(ars, t.src_off[m], soff, wstage + t.dst_off[m]);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (!(VSC_F16_ABLATE & 2)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    cur.b[0] = nb[0];
    cur.b[1] = nb[1];
    sp ^= 1;
}

// acc = (not +=) the tile's product; acc needs no initialisation.
__device__ __forceinline__ void stream_tile(TileStream& st, const _Float16* nq, const _Float16* nr, int dpadh,
                                            char* smem, const TileThread& t, f32x16 (&acc)[4][2]) {
    const int nkt = dpadh / BK;  // >= 2
    Frags cur = st.cur;
    int sp = st.sp;
    stream_ktile<true>(st, nq, nr, dpadh, 0, nkt, cur, sp, smem, t, acc);
    for (int kt = 1; kt < nkt; ++kt) stream_ktile<false>(st, nq, nr, dpadh, kt, nkt, cur, sp, smem, t, acc);
    st.cur = cur;
    st.sp = sp;
    st.q = nq;
    st.r = nr;
}

// XCD-aware raster (workgroup b runs on XCD b % 8): every XCD owns a contiguous run of tiles and walks
// it in bands of GQ query tiles, ref tile fastest inside... query tile fastest inside a band column, so
// the tiles in flight on one XCD share GQ query panels and a short run of ref panels in its L2.
// 32-bit arithmetic on purpose (the host guarantees tq * tr < 2^31): every wave evaluates this once per
// tile, and 64-bit divisions cost more instructions than the whole epilogue.
__device__ __forceinline__ bool raster(int xcd, int64_t local64, int tq, int64_t tr64, int& tqi, int64_t& tri) {
    const unsigned local = (unsigned)local64, tr = (unsigned)tr64;
    const unsigned nblk = (unsigned)tq * tr;
    const unsigned per_xcd = (nblk + 7u) >> 3;
    const unsigned logical = (unsigned)xcd * per_xcd + local;
    if (local >= per_xcd || logical >= nblk) return false;
#ifndef VSC_F16_GQ
#define VSC_F16_GQ 8
#endif
    constexpr unsigned GQ = VSC_F16_GQ;  // band height in query tiles (swept 4..32 on one box: see DESIGN.md)
    const unsigned band_sz = GQ * tr;
    const unsigned band = logical / band_sz, rem = logical - band * band_sz;
    const unsigned q0 = band * GQ;
    const unsigned gq = ((unsigned)tq - q0) < GQ ? ((unsigned)tq - q0) : GQ;
    const unsigned t = gq == GQ ? rem / GQ : rem / gq;
    tri = (int64_t)t;
    tqi = (int)(q0 + rem - t * gq);
    return true;
}

// Candidates of one wave tile -> the wave's PRIVATE segment of the candidate list: no atomics, no
// scans.  Per accumulator register one ballot; the (rare) non-empty ones are ranked with mbcnt.
// `count` is the wave-uniform fill level of the segment.
// lower edge of the candidate test for an exact threshold t: a pair with exact score >(=) t has fp16 score
// >= t - eps; the subtraction's own rounding (< 2^-23 relative to the larger operand) is subtracted again
__device__ __forceinline__ float candidate_edge(float t, float eps) {
    return (t - eps) - 2.4e-7f * (fabsf(t) + eps);
}

// ROWTHR = false: one threshold for the launch (`thr` = edge of *radius, strict test: thresholded search).
// ROWTHR = true : one threshold per query row (`rt` = the tile's 256 row thresholds in LDS, `thrb[m]` = edge
//                 of the smallest one of each 32-row block; non-strict test: k-NN, where ties must survive).
template <bool ROWTHR>
__device__ __forceinline__ void emit_candidates(const SimF16Args& a, bool all, float thr, const float (&thrb)[4],
                                                const float* rt, float eps, int row0, int row0_tile, int64_t col0,
                                                const f32x16 (&acc)[4][2], const float (&bm)[4][2], int lane,
                                                int64_t seg_base, int seg_cap, int& count, TailExt* ext) {
    // C layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int row_base = row0 + 4 * (lane >> 5);
    const int col_base = (int)col0 + (lane & 31);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            // first ask per 32x32 block (its max is already known), then per accumulator register
            if (!all && !__any(ROWTHR ? bm[m][n] >= thrb[m] : bm[m][n] > thr)) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                bool cand;
                if (ROWTHR) {
                    const int rl = row0_tile + 4 * (lane >> 5) + m * 32 + (r & 3) + 8 * (r >> 2);
                    cand = acc[m][n][r] >= candidate_edge(rt[rl], eps);
                } else {
                    cand = acc[m][n][r] > thr;
                }
                const unsigned long long hits = __ballot(all || cand);
                if (hits == 0ull) continue;
                const int i = row_base + m * 32 + (r & 3) + 8 * (r >> 2);
                const int j = col_base + n * 32;
                const unsigned long long ok = __ballot(((hits >> lane) & 1ull) && i < a.nq && j < a.nr);
                if (ok == 0ull) continue;
                const int total = __popcll(ok);
                int64_t pos;
                if (count + total <= seg_cap) {
                    pos = seg_base + count;
                    count += total;
                } else {
                    // segment full (candidates are not spread evenly): the wave's chunk of the shared tail
                    if (!tail_take(a.tail_count, a.tail_cap, a.tail_base, a.tail_shift, a.tail_fill, a.overflow, total, lane, ext, pos))
                        continue;
                }
                if ((ok >> lane) & 1ull) {
                    pos += __popcll(ok & ((1ull << lane) - 1));
                    a.out_i[pos] = a.i0 + i;
                    a.out_j[pos] = j;
                }
            }
        }
}

}  // namespace f16

template <bool ROWTHR>
__global__ __launch_bounds__(512, 1) void sim_f16_kernel(SimF16Args a) {
    using namespace f16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float norm_max_buf[2][8];  // double-buffered by tile parity (no barrier between tiles)
    __shared__ float row_thr_buf[2][ROWTHR ? 256 : 1];
    __shared__ TailExt tail_sh[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    TileThread t;
    tile_thread_init(t, tid, a.dpadh * 2);
    const float radius = ROWTHR ? 0.0f : *a.radius;
    const int xcd = blockIdx.x & 7;
    const int64_t lstride = gridDim.x >> 3;  // gridDim.x is a multiple of 8
    // this wave's private segment of the candidate list
    const int seg = blockIdx.x * 8 + __builtin_amdgcn_readfirstlane(wave);
    const int64_t seg_base = (int64_t)seg * a.seg_cap;
    int count = 0;
    tail_init(&tail_sh[wave], lane);
    int64_t local = blockIdx.x >> 3;
    int tqi;
    int64_t tri;
    if (!raster(xcd, local, a.tq, a.tr, tqi, tri)) {
        if (lane == 0) a.seg_count[seg] = 0;
        return;
    }
    TileStream st;
    stream_begin(st, a.Q + (int64_t)tqi * BM * a.dpadh, a.R + tri * BN * a.dpadh, a.dpadh, smem, t);
    for (int parity = 0;; parity ^= 1) {
        int ntq = 0;
        int64_t ntr = 0;
        const bool has_next = raster(xcd, local + lstride, a.tq, a.tr, ntq, ntr);
        float* norm_max = norm_max_buf[parity];
        // largest row norm of each side of the tile (waves 0-3: query rows, 4-7: ref rows)
        float nv = tid < 256 ? a.qn[(int64_t)tqi * BM + tid] : a.rn[tri * BN + (tid - 256)];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nv = fmaxf(nv, __shfl_xor(nv, off));
        if (lane == 0) norm_max[wave] = nv;
        if (ROWTHR && tid < 256) row_thr_buf[parity][tid] = a.row_thr[(int64_t)tqi * BM + tid];
        f32x16 acc[4][2];
        stream_tile(st, has_next ? a.Q + (int64_t)ntq * BM * a.dpadh : st.q,
                    has_next ? a.R + ntr * BN * a.dpadh : st.r, a.dpadh, smem, t, acc);
        // (the barriers of the K loop ordered the norm_max writes)
        const float nq = fmaxf(fmaxf(norm_max[0], norm_max[1]), fmaxf(norm_max[2], norm_max[3]));
        const float nr = fmaxf(fmaxf(norm_max[4], norm_max[5]), fmaxf(norm_max[6], norm_max[7]));
        const float eps = (a.c1 * nq * nr + a.c2 * (nq + nr) + a.c3) * 1.001f;
        const bool all = !(eps < INFINITY);  // also catches NaN (inf * 0)
        const float thr = candidate_edge(radius, eps);
        float thrb[4] = {thr, thr, thr, thr};
        const float* rt = row_thr_buf[ROWTHR ? parity : 0];
        if (ROWTHR) {
            // smallest row threshold of each of the wave's four 32-row blocks
            float v0 = rt[wr * 128 + lane], v1 = rt[wr * 128 + 64 + lane];
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) {
                v0 = fminf(v0, __shfl_xor(v0, off));
                v1 = fminf(v1, __shfl_xor(v1, off));
            }
            thrb[0] = candidate_edge(__shfl(v0, 0), eps);
            thrb[1] = candidate_edge(__shfl(v0, 32), eps);
            thrb[2] = candidate_edge(__shfl(v1, 0), eps);
            thrb[3] = candidate_edge(__shfl(v1, 32), eps);
        }
        // candidates are rare: one max per 32x32 block first, one compare for the whole wave tile
        float bm[4][2];
        float mx = -INFINITY;
        bool any_blk = false;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                float v = fmaxf(acc[m][n][0], acc[m][n][1]);
#pragma unroll
                for (int r = 2; r < 16; r += 2) v = fmaxf(v, fmaxf(acc[m][n][r], acc[m][n][r + 1]));
                bm[m][n] = v;
                mx = fmaxf(mx, v);
                if (ROWTHR) any_blk |= v >= thrb[m];
            }
        if (VSC_F16_ABLATE & 7) {
            count += mx == 12345.678f;  // keeps the accumulators alive; garbage results are not emitted
        } else if (all || __any(ROWTHR ? any_blk : mx > thr))
            emit_candidates<ROWTHR>(a, all, thr, thrb, rt, eps, tqi * BM + wr * 128, wr * 128,
                                    tri * BN + wc * 64, acc, bm, lane, seg_base, a.seg_cap, count, &tail_sh[wave]);
        if (!has_next) break;
        local += lstride;
        tqi = ntq;
        tri = ntr;
    }
    // the stream fetched two K-tiles past its end: let them land before the LDS is handed back
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tail_close(a.tail_base, a.tail_shift, a.tail_fill, lane, &tail_sh[wave]);
    if (lane == 0) a.seg_count[seg] = count;
}

// grid of a launch: one persistent workgroup per CU (fewer for tiny problems); 8 segments each
int sim_f16_grid(int tq, int tr) {
    const int64_t nblk = (int64_t)tq * tr;
    int64_t grid = ((nblk + 7) / 8) * 8;
    if (grid > 256) grid = 256;
    return (int)grid;
}

int launch_sim_f16(const SimF16Args& a, hipStream_t stream) {
    static PerDeviceOnce once;
    if (once.first()) {
        VSC_HIP(hipFuncSetAttribute((const void*)sim_f16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    f16::LDS_BYTES));
        VSC_HIP(hipFuncSetAttribute((const void*)sim_f16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    f16::LDS_BYTES));
        once.commit();
    }
    if ((int64_t)a.tq * a.tr >= 0x7fffffffLL) {
        set_error("sim_f16: more than 2^31 output tiles in one launch");
        return VSC_ERR_INVALID;
    }
    const int grid = sim_f16_grid(a.tq, a.tr);
    if (grid <= 0) return VSC_OK;
    if (a.row_thr)
        hipLaunchKernelGGL(sim_f16_kernel<true>, dim3((unsigned)grid), dim3(512), f16::LDS_BYTES, stream, a);
    else
        hipLaunchKernelGGL(sim_f16_kernel<false>, dim3((unsigned)grid), dim3(512), f16::LDS_BYTES, stream, a);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

// ------------------------------------------------------------------------------ exact stage
//
// acc = fmaf(q[k], r[k], acc), k ascending from +0 (the arithmetic contract of the engine; packed rows hold
// every group of 8 k as [k0 k2 k4 k6 | k1 k3 k5 k7]).  The chain is serial, but nothing says it must stay
// in one lane: FOUR lanes share a candidate.  In every round of 32 k, lane g of the quad loads the g-th
// 32-byte group of both rows (so one load instruction touches one full 128-byte line per candidate and
// row instead of four different lines), and the running sum hops from lane to lane with a quad-rotate
// DPP move: lane 0 does k 0-7, hands over to lane 1 for k 8-15, ... and lane 3 hands back to lane 0 for the
// next round.  Every lane executes every step (the other three results are discarded), which costs 4x
// the fma issue slots of a chain that needs ~3 % of the VALU anyway; the kernel is bound by row traffic.
__device__ __forceinline__ float quad_rotate(float v) {  // lane g of every quad receives lane (g - 1) & 3
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x93, 0xf, 0xf, true));
}

// Hits are collected per wave in LDS and appended to the global list 49-64 at a time: one atomic on the
// (single, hot) list counter per flush instead of one per 16 candidates.
struct WaveHits {
    int i[64];
    int j[64];
    float s[64];
};

__device__ __forceinline__ void flush_hits(const RescoreArgs& a, WaveHits& buf, int& pend) {
    if (pend == 0) return;
    const int lane = threadIdx.x & 63;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's LDS writes are visible to its other lanes
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(a.counter, (unsigned long long)pend);
    base = __shfl(base, 0);
    if ((long long)(base + pend) > a.cap) {
        if (lane == 0) atomicOr(a.overflow, 1);
    } else if (lane < pend) {
        a.out_i[base + lane] = buf.i[lane];
        a.out_j[base + lane] = buf.j[lane];
        a.out_s[base + lane] = buf.s[lane];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the buffer is refilled
    pend = 0;
}

// candidates of one list; thread x serves candidate x >> 2 (x0 = first thread index, `step` threads apart)
__device__ __forceinline__ void rescore_list(const RescoreArgs& a, float radius, const int32_t* ci,
                                             const int32_t* cj, long long n, long long x0, long long step,
                                             WaveHits& buf, int& pend, const int* fill = nullptr, int shift = 0) {
    const int lane = threadIdx.x & 63, g = lane & 3;
    const long long n_thr = (4 * n + 63) & ~63ll;  // whole waves stay together (DPP, ballot)
    const int rounds = a.dpad / 32;
    for (long long x = x0; x < n_thr; x += step) {
        const long long c = x >> 2;
        // (the tail is a sequence of chunks, each filled up to its own level: cand_list.h)
        const bool valid = c < n && (fill == nullptr || (int)(c & ((1ll << shift) - 1)) < fill[c >> shift]);
        int i = valid ? ci[c] : 0;
        const int j = valid ? cj[c] + a.j0 : 0;
        if (a.perm && valid) i = a.perm_i0 + a.perm[i - a.perm_i0];  // position inside a permuted int8 launch -> row
        const f32x4* q = reinterpret_cast<const f32x4*>(a.Q + (int64_t)i * a.dpad) + 2 * g;
        const f32x4* r = reinterpret_cast<const f32x4*>(a.R + (int64_t)j * a.dpad) + 2 * g;
        float acc = 0.0f;  // the live value sits in lane 0 of the quad at the top of every round
        f32x4 qe = q[0], qo = q[1], re = r[0], ro = r[1];
        for (int rd = 0; rd < rounds; ++rd) {
            const int nx = rd + 1 < rounds ? rd + 1 : rd;  // prefetch the next round's groups
            const f32x4 nqe = q[8 * nx], nqo = q[8 * nx + 1], nre = r[8 * nx], nro = r[8 * nx + 1];
#pragma unroll
            for (int gp = 0; gp < 4; ++gp) {
                float v = acc;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    v = __fmaf_rn(qe[s], re[s], v);
                    v = __fmaf_rn(qo[s], ro[s], v);
                }
                const float passed = quad_rotate(v);  // lane gp's (the only meaningful) v -> lane gp + 1
                acc = (g == ((gp + 1) & 3)) ? passed : acc;
            }
            qe = nqe;
            qo = nqo;
            re = nre;
            ro = nro;
        }
        const bool hit = valid && g == 0 && (a.row_thr ? acc >= a.row_thr[i] : acc > radius);
        const unsigned long long m = __ballot(hit);  // <= 16 hits per pass
        if (hit) {
            const int p = pend + __popcll(m & ((1ull << lane) - 1));
            buf.i[p] = i;
            buf.j[p] = j;
            buf.s[p] = acc;
        }
        pend += __popcll(m);
        if (pend > 48) flush_hits(a, buf, pend);
    }
}

#ifndef VSC_RESCORE_SHARE
#define VSC_RESCORE_SHARE 4
#endif
constexpr int RESCORE_SHARE = VSC_RESCORE_SHARE;

__global__ __launch_bounds__(256) void rescore_kernel(RescoreArgs a) {
    // After an overflow the candidate list has holes (a wave whose tail reservation did not fit skipped its
    // writes but the tail counter moved on): the host reruns the search with larger buffers, so do nothing
    // rather than chase unwritten (row, ref) pairs through memory.
    if (*a.overflow) return;
    __shared__ WaveHits wave_hits[4];
    WaveHits& buf = wave_hits[threadIdx.x >> 6];
    int pend = 0;
    const float radius = a.row_thr ? 0.0f : *a.radius;
    unsigned long long seen = 0;
    // RESCORE_SHARE workgroups walk one segment together (the segments fill unevenly: more, smaller pieces balance
    // better and keep more loads in flight)
    {
        const int seg = blockIdx.x / RESCORE_SHARE, part = blockIdx.x % RESCORE_SHARE;
        const int n = min(a.seg_count[seg], a.seg_cap);
        if (part == 0) seen += (unsigned long long)n;
        rescore_list(a, radius, a.cand_i + (int64_t)seg * a.seg_cap, a.cand_j + (int64_t)seg * a.seg_cap, n,
                     part * 256 + threadIdx.x, 256 * RESCORE_SHARE, buf, pend);
    }
    // shared tail (normally empty)
    const unsigned long long nt_all = *a.tail_count;
    const long long nt = nt_all < (unsigned long long)a.tail_cap ? (long long)nt_all : a.tail_cap;
    if (nt > 0) {
        rescore_list(a, radius, a.cand_i + a.tail_base, a.cand_j + a.tail_base, nt,
                     (long long)blockIdx.x * 256 + threadIdx.x, (long long)gridDim.x * 256, buf, pend, a.tail_fill,
                     a.tail_shift);
        if (blockIdx.x == 0) seen += (unsigned long long)nt;
    }
    flush_hits(a, buf, pend);
    if (threadIdx.x == 0 && seen) atomicAdd(a.n_cand_total, seen);
}

__global__ void tail_reset_kernel(unsigned long long* tail_count) { *tail_count = 0; }

// ---- candidates ordered by reference row
// The candidates of a launch hit every reference row several times (int8 batches of the search: ~4 x, k-NN passes:
// 10-80 x).  Compacted out of the waves' segments and sorted by reference row, the chains of one reference row sit in
// neighbouring lanes: their loads of that row are one cache line request instead of several, and the row comes out of
// HBM once.  cand_compact: one workgroup per segment / tail chunk, dense position by one atomic per workgroup (the
// order inside the dense list does not matter: it is sorted next).
__global__ __launch_bounds__(256) void cand_compact_kernel(RescoreArgs a, uint32_t* __restrict__ key_j,
                                                           uint32_t* __restrict__ val_i, unsigned long long* n_out) {
    __shared__ unsigned long long base_sh;
    if (*a.overflow) return;
    const int b = blockIdx.x;
    const int32_t *ci, *cj;
    int n;
    if (b < a.n_seg) {
        n = min(a.seg_count[b], a.seg_cap);
        ci = a.cand_i + (int64_t)b * a.seg_cap;
        cj = a.cand_j + (int64_t)b * a.seg_cap;
    } else {
        const long long chunk = b - a.n_seg;
        const unsigned long long nt = *a.tail_count;
        if ((unsigned long long)(chunk << a.tail_shift) >= nt || (long long)nt > a.tail_cap) return;
        n = a.tail_fill[chunk];
        ci = a.cand_i + a.tail_base + (chunk << a.tail_shift);
        cj = a.cand_j + a.tail_base + (chunk << a.tail_shift);
    }
    if (n <= 0) return;
    if (threadIdx.x == 0) base_sh = atomicAdd(n_out, (unsigned long long)n);
    __syncthreads();
    const unsigned long long base = base_sh;
    for (int x = threadIdx.x; x < n; x += 256) {
        int i = ci[x];
        if (a.perm) i = a.perm_i0 + a.perm[i - a.perm_i0];  // position inside a permuted int8 launch -> row
        key_j[base + x] = (uint32_t)(cj[x] + a.j0);
        val_i[base + x] = (uint32_t)i;
    }
}

// fp16 screen between the int8 pre-filter and the exact stage.  The int8 bound is wide (one scale per row / panel:
// ~5-10 candidates per pair that really reaches the threshold); the fp16 bound is ~30x narrower, and an fp16 row is
// half the bytes of the fp32 row the exact stage gathers.  One candidate per quad: lane g takes the 16-byte pieces
// g, g + 4, ... of both rows (any summation order is inside the bound: products of two fp16 values are exact in fp32,
// each fma rounds once), the quad's sum is compared like the fp16 pre-filter compares its scores
// (sim_f16p.hip: candidate_edge).  Survivors are collected per wave in LDS and appended 49-64 at a time; the
// list stays (roughly) in reference-row order.
struct WavePairs {
    uint32_t i[64];
    uint32_t j[64];
};

__device__ __forceinline__ float screen_edge(float t, float eps) { return (t - eps) - 2.4e-7f * (fabsf(t) + eps); }

template <bool FRAG>
__global__ __launch_bounds__(256) void f16_screen_kernel(ScreenArgs a) {
    if (*a.overflow) return;
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    __shared__ WavePairs wave_pairs[4];
    WavePairs& buf = wave_pairs[threadIdx.x >> 6];
    int pend = 0;
    const int lane = threadIdx.x & 63, g = lane & 3;
    const long long n_thr = (4 * a.n + 63) & ~63ll;
    const int npiece = a.dpadh / 8, nks = a.dpadh / 16;
    const float radius = a.row_thr ? 0.0f : *a.radius;
    const f16x8* __restrict__ Rp = reinterpret_cast<const f16x8*>(a.Rh);
    auto flush = [&]() {
        if (pend == 0) return;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(a.n_out, (unsigned long long)pend);
        base = __shfl(base, 0);
        if (lane < pend) {
            a.out_i[base + lane] = buf.i[lane];
            a.out_j[base + lane] = buf.j[lane];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        pend = 0;
    };
    for (long long x = (long long)blockIdx.x * 256 + threadIdx.x; x < n_thr; x += (long long)gridDim.x * 256) {
        const long long c = x >> 2;
        const bool valid = c < a.n;
        const uint32_t i = valid ? a.si[c] : 0u, j = valid ? a.sj[c] : 0u;
        const f16x8* __restrict__ q = reinterpret_cast<const f16x8*>(a.Qh + (int64_t)i * a.dpadh);
        const int64_t rbase = FRAG ? (int64_t)(j >> 6) * nks * 128 + ((j >> 5) & 1) * 64 + (j & 31)
                                   : (int64_t)j * npiece;
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll 8
        for (int p = g; p < npiece; p += 4) {
            const f16x8 qv = q[p];
            const f16x8 rv = FRAG ? Rp[rbase + (int64_t)(p >> 1) * 128 + (p & 1) * 32] : Rp[rbase + p];
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                s0 = __fmaf_rn((float)qv[e], (float)rv[e], s0);
                s1 = __fmaf_rn((float)qv[e + 1], (float)rv[e + 1], s1);
            }
        }
        float acc = s0 + s1;
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        const float eps = (a.c1 * a.qn[i] * a.rn[j] + a.c2 * (a.qn[i] + a.rn[j]) + a.c3) * 1.001f;
        const bool pass = valid && g == 0 &&
                          (!(eps < INFINITY) || (a.row_thr ? acc >= screen_edge(a.row_thr[i], eps) : acc > screen_edge(radius, eps)));
        const unsigned long long m = __ballot(pass);  // <= 16 per pass
        if (pass) {
            const int p = pend + __popcll(m & ((1ull << lane) - 1));
            buf.i[p] = i;
            buf.j[p] = j;
        }
        pend += __popcll(m);
        if (pend > 48) flush();
    }
    flush();
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(a.n_cand_total, (unsigned long long)a.n);
}

int launch_f16_screen(const ScreenArgs& a, hipStream_t stream) {
    VSC_HIP(hipMemsetAsync(a.n_out, 0, sizeof(unsigned long long), stream));
    if (a.n > 0) {
        const unsigned grid = (unsigned)std::min<long long>(16384, (a.n * 4 + 255) / 256);
        if (a.frag) hipLaunchKernelGGL(f16_screen_kernel<true>, dim3(grid), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL(f16_screen_kernel<false>, dim3(grid), dim3(256), 0, stream, a);
        VSC_HIP(hipGetLastError());
    }
    return VSC_OK;
}

// n_dev: the list's length lives on the device (the fp16 screen's survivors; n = an upper bound for the grid)
__global__ __launch_bounds__(256) void rescore_dense_kernel(RescoreArgs a, const uint32_t* __restrict__ sj,
                                                            const uint32_t* __restrict__ si, long long n,
                                                            const unsigned long long* __restrict__ n_dev) {
    if (*a.overflow) return;
    if (n_dev) n = (long long)*n_dev;
    __shared__ WaveHits wave_hits[4];
    WaveHits& buf = wave_hits[threadIdx.x >> 6];
    int pend = 0;
    const float radius = a.row_thr ? 0.0f : *a.radius;
    a.perm = nullptr;  // (rows already)
    a.j0 = 0;          // (absolute reference rows already: cand_compact added the launch's offset)
    rescore_list(a, radius, reinterpret_cast<const int32_t*>(si), reinterpret_cast<const int32_t*>(sj), n,
                 (long long)blockIdx.x * 256 + threadIdx.x, (long long)gridDim.x * 256, buf, pend);
    flush_hits(a, buf, pend);
    if (blockIdx.x == 0 && threadIdx.x == 0 && !n_dev) atomicAdd(a.n_cand_total, (unsigned long long)n);
}

// How many candidates a launch left in its segments and tail chunks (one workgroup): the host sizes the dense lists
// of the sort from this number instead of from the list's capacity (round 4: 4 x 16 bytes per unit of CAPACITY went to
// buffers that a launch fills to a few percent).  An overflowed launch counts 0 (nothing of it is re-scored).
__global__ __launch_bounds__(1024) void cand_count_kernel(RescoreArgs a, int n_chunks_max, unsigned long long* n_out) {
    __shared__ unsigned long long red[16];
    unsigned long long s = 0;
    if (!*a.overflow) {
        for (int b = threadIdx.x; b < a.n_seg; b += 1024) s += (unsigned long long)max(0, min(a.seg_count[b], a.seg_cap));
        const unsigned long long nt = *a.tail_count;
        if ((long long)nt <= a.tail_cap) {
            const long long used = (long long)((nt + (1ull << a.tail_shift) - 1) >> a.tail_shift);
            for (long long c = threadIdx.x; c < used && c < n_chunks_max; c += 1024) s += (unsigned long long)max(0, a.tail_fill[c]);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < 16; ++w) t += red[w];
        *n_out = t;
    }
}

int launch_cand_count(const RescoreArgs& a, int n_chunks_max, unsigned long long* n_out, hipStream_t stream) {
    hipLaunchKernelGGL(cand_count_kernel, dim3(1), dim3(1024), 0, stream, a, n_chunks_max, n_out);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

int launch_cand_compact(const RescoreArgs& a, int n_chunks_max, uint32_t* key_j, uint32_t* val_i, unsigned long long* n_out,
                        hipStream_t stream) {
    VSC_HIP(hipMemsetAsync(n_out, 0, sizeof(unsigned long long), stream));
    hipLaunchKernelGGL(cand_compact_kernel, dim3((unsigned)(a.n_seg + n_chunks_max)), dim3(256), 0, stream, a, key_j, val_i, n_out);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

int launch_rescore_dense(const RescoreArgs& a, const uint32_t* sj, const uint32_t* si, long long n, hipStream_t stream,
                         const unsigned long long* n_dev) {
    if (n > 0) {
        const unsigned grid = (unsigned)std::min<long long>(16384, (n * 4 + 255) / 256);
        hipLaunchKernelGGL(rescore_dense_kernel, dim3(grid), dim3(256), 0, stream, a, sj, si, n, n_dev);
    }
    hipLaunchKernelGGL(tail_reset_kernel, dim3(1), dim3(1), 0, stream, a.tail_count);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

int launch_rescore(const RescoreArgs& a, hipStream_t stream) {
    if (a.n_seg <= 0) return VSC_OK;
    hipLaunchKernelGGL(rescore_kernel, dim3((unsigned)a.n_seg * RESCORE_SHARE), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(tail_reset_kernel, dim3(1), dim3(1), 0, stream, a.tail_count);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

}  // namespace vscmi
