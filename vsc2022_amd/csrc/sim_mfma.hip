// Similarity kernels (gfx950 / CDNA4 only): S = Q . R^T on the matrix cores with fused selection.
//
// This is the replacement for what FAISS does underneath vsc/index.py:147-154,174 and
// vsc/baseline/score_normalization.py:96 (exhaustive_inner_product_blas = blocked sgemm +
// threshold scan / heap): one pass over the 128x128x512 tile on v_mfma_f32_32x32x2_f32 with the
// selection done on the accumulators, so the NQ x NR score matrix never exists in HBM.
//
// Roofline: MFMA (fp32 matrix rate, 157.3 TFLOP/s).  Algorithmic flops = 2*NQ*NR*dim.
//
// Tile: 128 query rows x 128 ref rows per workgroup, 4 waves (2x2), each wave 64x64 = 2x2 MFMA
// tiles of 32x32 (64 accumulator VGPRs).  K is walked in steps of 32 floats; both operand tiles are
// streamed global -> LDS with global_load_lds_dwordx4 (no VGPR round trip) into a 2-stage ring
// (2 x 32 KiB => 2 workgroups per CU).  LDS image: 256-byte lines of two rows; 16-byte slot index
// s = (row&1)<<3 | (chunk ^ (line&7)), which makes every ds_read_b128 lane group hit 16 distinct
// slots (conflict-free) while the DMA destination stays lane-linear (swizzle on the source side).
#include <cfloat>
#include <cstdlib>

#include "kernels.h"

namespace vscmi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int BK = 32;
constexpr int TILE_BYTES = BM * BK * 4;      // 16 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * TILE_BYTES;  // A + B
constexpr int GEMM_LDS = 2 * STAGE_BYTES;    // 64 KiB

// LDS-DMA of one 16-byte piece per lane: buffer_load_dwordx4 ... lds.  The buffer descriptor is
// wave-uniform (tile base), the per-lane part is a 32-bit byte offset, the K-tile advance rides in
// the scalar offset: no 64-bit address arithmetic per issue.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, char* lds) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff,
                                             soff, 0, 0);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const float* base, int dpad) {
    // make the (already wave-uniform) base provably uniform so the descriptor lives in SGPRs
    const uint64_t p = reinterpret_cast<uint64_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
    void* up = reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(up, 0, BM * dpad * 4, 0x00020000);
}

// Per-thread constants of the tile pipeline.
struct TileThread {
    int src_off[4];   // BYTE offset (row*dpad + chunk*4)*4 of the 4 DMA pieces this thread issues
    int dst_off[4];   // wave-uniform LDS byte offset of those pieces inside an operand tile
    int rdA[2][4];    // LDS byte offsets of the A fragments [tm][kk]
    int rdB[2][4];    // LDS byte offsets of the B fragments [tn][kk]
};

__device__ __forceinline__ void tile_thread_init(TileThread& t, int tid, int dpad) {
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int p = n * 256 + tid;  // 16-byte slot in the tile image
        const int line = p >> 4, s = p & 15;
        const int row = 2 * line + (s >> 3);
        const int chunk = (s & 7) ^ (line & 7);
        t.src_off[n] = (row * dpad + chunk * 4) * 4;
        t.dst_off[n] = (n * 256 + __builtin_amdgcn_readfirstlane(wave) * 64) * 16;
    }
    const int hi = lane >> 5;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int rowA = wr * 64 + m * 32 + (lane & 31);
        const int rowB = wc * 64 + m * 32 + (lane & 31);
        const int lineA = rowA >> 1, lineB = rowB >> 1;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int c = 2 * kk + hi;
            t.rdA[m][kk] = lineA * 256 + ((((rowA & 1) << 3) | (c ^ (lineA & 7))) << 4);
            t.rdB[m][kk] = lineB * 256 + ((((rowB & 1) << 3) | (c ^ (lineB & 7))) << 4);
        }
    }
}

struct Frags {
    f32x4 a0, a1, b0, b1;
};

__device__ __forceinline__ Frags read_frags(const char* stage, const TileThread& t, int kk) {
    Frags f;
    f.a0 = *reinterpret_cast<const f32x4*>(stage + t.rdA[0][kk]);
    f.a1 = *reinterpret_cast<const f32x4*>(stage + t.rdA[1][kk]);
    f.b0 = *reinterpret_cast<const f32x4*>(stage + TILE_BYTES + t.rdB[0][kk]);
    f.b1 = *reinterpret_cast<const f32x4*>(stage + TILE_BYTES + t.rdB[1][kk]);
    return f;
}

__device__ __forceinline__ void mfma4(const Frags& f, int s, f32x16 (&acc)[2][2]) {
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0[s], f.b0[s], acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0[s], f.b1[s], acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1[s], f.b0[s], acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1[s], f.b1[s], acc[1][1], 0, 0, 0);
}

__device__ __forceinline__ void stage_tiles(__amdgpu_buffer_rsrc_t qrs, __amdgpu_buffer_rsrc_t rrs, int kt,
                                            char* stage, const TileThread& t) {
#pragma unroll
    for (int n = 0; n < 4; ++n) dma16(qrs, t.src_off[n], kt * BK * 4, stage + t.dst_off[n]);
#pragma unroll
    for (int n = 0; n < 4; ++n) dma16(rrs, t.src_off[n], kt * BK * 4, stage + TILE_BYTES + t.dst_off[n]);
}

// ---- the tile stream -------------------------------------------------------------------------
// A workgroup is persistent: it walks a list of output tiles and treats their K-tiles (32 floats each)
// as ONE continuous stream through the 2-stage LDS ring, so there is no load prologue per output
// tile: while the last K-tiles of tile T are multiplied, the first two K-tiles of tile T+1 are
// already in flight.  Pipeline per K-tile (4 groups of 16 MFMAs per wave):
//   * fragments are double-buffered in registers: the ds_read_b128 of group g+1 are issued before the
//     MFMAs of group g;
//   * the one barrier per K-tile sits BEFORE the last MFMA group of the K-tile (whose operands are
//     already in registers), so the first fragment reads of the next K-tile and the 8 LDS-DMA issues
//     of the one after it are covered by 16 MFMAs (1024 matrix-pipe cycles);
//   * the DMA issues are interleaved one per two MFMAs (sched_group_barrier), never ahead of them.
struct TileStream {
    const float* q;   // operand panels of the current tile (wave-uniform)
    const float* r;
    Frags cur;        // fragments (K-tile 0, group 0) of the current tile
    int sp;           // LDS stage holding K-tile 0 of the current tile
};

__device__ __forceinline__ void stream_begin(TileStream& st, const float* qrow0, const float* rrow0,
                                             int dpad, char* smem, const TileThread& t) {
    st.q = qrow0;
    st.r = rrow0;
    st.sp = 0;
    const __amdgpu_buffer_rsrc_t qrs = tile_rsrc(qrow0, dpad), rrs = tile_rsrc(rrow0, dpad);
    stage_tiles(qrs, rrs, 0, smem, t);
    // explicit: a K-tile has landed only when EVERY wave's own LDS-DMA count has drained (the compiler
    // usually adds this wait before the barrier, but not reliably for the builtin -- see sim_f16.hip)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // K-tile 0 landed
    stage_tiles(qrs, rrs, 1, smem + STAGE_BYTES, t);  // dpad >= 64: K-tile 1 always exists
    st.cur = read_frags(smem, t, 0);
}

// acc += Q[tile rows, :] . R[tile rows, :]^T for the stream's current tile.  If has_next, the stream is
// left positioned on the tile whose panels start at (nq, nr); otherwise pass the current panels.
//
// There is exactly ONE copy of every MFMA in this loop: the conditionals (is there a next K-tile to
// wait for / a K-tile two ahead to fetch) only guard the barrier, the fragment reads and the DMA
// issues.  Duplicating the MFMA groups across branches makes the register allocator copy the 64
// accumulator VGPRs at the merge points (v_mov + MFMA-result hazard nops every K-tile).
__device__ __forceinline__ void stream_tile(TileStream& st, bool has_next, const float* nq, const float* nr,
                                            int dpad, char* smem, const TileThread& t,
                                            f32x16 (&acc)[2][2]) {
    const int nkt = dpad / BK;
    Frags cur = st.cur;
    int sp = st.sp;
    for (int kt = 0; kt < nkt; ++kt) {
        const char* stage = smem + sp * STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            const Frags nxt = read_frags(stage, t, kk + 1);
#pragma unroll
            for (int s = 0; s < 4; ++s) mfma4(cur, s, acc);
            // pin the order: the 4 fragment reads of the NEXT group go out before this group's 16
            // MFMAs (left alone, the scheduler sinks them to just before their use and exposes the
            // LDS latency four times per K-tile)
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            cur = nxt;
        }
        // Last group of this K-tile: its operands are in registers, so the stage is free.
        const int k2 = kt + 2;
        const bool in_cur = k2 < nkt;
        const bool n1 = (kt + 1 < nkt) || has_next;  // a next K-tile exists in the stream
        const bool n2 = in_cur || has_next;          // ... and one after it, to be fetched now
        Frags nxt = cur;
        if (n1) {
            // the next K-tile (issued one K-tile ago) has landed once every wave's DMA count drains
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            nxt = read_frags(smem + (sp ^ 1) * STAGE_BYTES, t, 0);
        }
        const __amdgpu_buffer_rsrc_t q2 = tile_rsrc(in_cur ? st.q : nq, dpad);
        const __amdgpu_buffer_rsrc_t r2 = tile_rsrc(in_cur ? st.r : nr, dpad);
        const int soff = (in_cur ? k2 : k2 - nkt) * BK * 4;
        char* wstage = smem + sp * STAGE_BYTES;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            // one DMA per two MFMAs: an LDS-DMA issue occupies the wave for ~60 cycles, one matrix
            // instruction for 64 -- spaced like this the issue cost disappears behind the pipe
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a0[s], cur.b0[s], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a0[s], cur.b1[s], acc[0][1], 0, 0, 0);
            if (n2) dma16(q2, t.src_off[s], soff, wstage + t.dst_off[s]);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a1[s], cur.b0[s], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a1[s], cur.b1[s], acc[1][1], 0, 0, 0);
            if (n2) dma16(r2, t.src_off[s], soff, wstage + TILE_BYTES + t.dst_off[s]);
        }
        cur = nxt;
        if (n1) sp ^= 1;
    }
    st.cur = cur;
    st.sp = sp;
    st.q = nq;
    st.r = nr;
}

// XCD-aware rasterisation.  The dispatcher places workgroup b on XCD b % 8; every XCD owns a
// contiguous run of tiles, and inside a run walks bands of GQ query tiles so that the ~64 tiles in
// flight on one XCD share GQ query panels and a short run of ref panels in its private L2.
// `local` = index inside the XCD's run.
__device__ __forceinline__ bool raster(int xcd, int64_t local, int tq, int64_t tr, int& tqi, int64_t& tri) {
    const int64_t nblk = (int64_t)tq * tr;
    const int64_t per_xcd = (nblk + 7) / 8;
    const int64_t logical = (int64_t)xcd * per_xcd + local;
    if (local >= per_xcd || logical >= nblk) return false;
    constexpr int GQ = 8;
    const int64_t band_sz = (int64_t)GQ * tr;
    const int64_t band = logical / band_sz, rem = logical % band_sz;
    const int q0 = (int)band * GQ;
    const int gq = (tq - q0) < GQ ? (tq - q0) : GQ;
    tri = rem / gq;
    tqi = q0 + (int)(rem % gq);
    return true;
}

// ------------------------------------------------------------------ threshold compaction

// append every accumulator > radius of one 128x128 tile to the hit list (one atomic per wave)
__device__ __forceinline__ void emit_hits(const SimThreshArgs& a, float radius, int q0, int64_t r0,
                                          const f32x16 (&acc)[2][2], int lane, int wr, int wc) {
    // C layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int row_base = q0 + wr * 64 + 4 * (lane >> 5);
    const int col_base = (int)r0 + wc * 64 + (lane & 31);
    unsigned long long mask = 0;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = row_base + m * 32 + (r & 3) + 8 * (r >> 2);
                const int j = col_base + n * 32;
                const bool hit = (acc[m][n][r] > radius) && (i < a.nq) && (j < a.nr);
                mask |= (unsigned long long)hit << ((m * 2 + n) * 16 + r);
            }
    if (!__any(mask != 0ull)) return;
    const int cnt = __popcll(mask);
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    const int total = __shfl(incl, 63);
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(a.counter, (unsigned long long)total);
    base = __shfl(base, 0);
    if ((long long)(base + total) > a.cap) {
        if (lane == 0) atomicOr(a.overflow, 1);
        return;
    }
    unsigned long long pos = base + (unsigned)(incl - cnt);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if ((mask >> ((m * 2 + n) * 16 + r)) & 1ull) {
                    a.out_i[pos] = a.i0 + row_base + m * 32 + (r & 3) + 8 * (r >> 2);
                    a.out_j[pos] = col_base + n * 32;
                    a.out_s[pos] = acc[m][n][r];
                    ++pos;
                }
            }
}

__global__ __launch_bounds__(256, 2) void sim_thresh_kernel(SimThreshArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int xcd = blockIdx.x & 7;
    const int64_t lstride = gridDim.x >> 3;  // gridDim.x is a multiple of 8
    int64_t local = blockIdx.x >> 3;
    int tqi;
    int64_t tri;
    if (!raster(xcd, local, a.tq, a.tr, tqi, tri)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    TileThread t;
    tile_thread_init(t, tid, a.dpad);
    const float radius = *a.radius;
    TileStream st;
    stream_begin(st, a.Q + (int64_t)tqi * BM * a.dpad, a.R + tri * BN * a.dpad, a.dpad, smem, t);
    for (;;) {
        int ntq = 0;
        int64_t ntr = 0;
        const bool has_next = raster(xcd, local + lstride, a.tq, a.tr, ntq, ntr);
        f32x16 acc[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
        stream_tile(st, has_next, has_next ? a.Q + (int64_t)ntq * BM * a.dpad : st.q,
                    has_next ? a.R + ntr * BN * a.dpad : st.r, a.dpad, smem, t, acc);
        emit_hits(a, radius, tqi * BM, tri * BN, acc, lane, wr, wc);
        if (!has_next) break;
        local += lstride;
        tqi = ntq;
        tri = ntr;
    }
}

static int sim_grid_limit() {
    static int g = 0;
    if (g == 0) {
        const char* e = getenv("VSC_SIM_GRID");
        // Default: one tile per workgroup, i.e. let the hardware dispatcher hand tiles out in order.
        // Measured (65536 x 1M x 512): 136.6 TFLOP/s and 0.14 TB of L2->fabric reads per search,
        // against 134 TFLOP/s and 1.1 TB with 512 persistent workgroups striding statically (they
        // drift apart and stop sharing panels in L2) and 130 TFLOP/s / 0.28 TB with persistent
        // workgroups pulling in-order tickets.  VSC_SIM_GRID=512 re-enables the persistent stream.
        g = e ? atoi(e) : 0x7ffffff8;
        if (g < 8) g = 8;
        g = (g / 8) * 8;
    }
    return g;
}

int launch_sim_thresh(const SimThreshArgs& a, hipStream_t stream) {
    const int64_t nblk = (int64_t)a.tq * a.tr;
    if (nblk <= 0) return VSC_OK;
    int64_t grid = ((nblk + 7) / 8) * 8;
    if (grid > sim_grid_limit()) grid = sim_grid_limit();
    hipLaunchKernelGGL(sim_thresh_kernel, dim3((unsigned)grid), dim3(256), GEMM_LDS, stream, a);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

// ------------------------------------------------------------------------------ k-NN
//
// One workgroup owns 128 query rows and walks a contiguous run of reference tiles.  The running
// per-row top-k lives in LDS, sorted by (score desc, ref asc), one entry per lane (k <= 64) so
// an insertion is ballot + one wavefront shuffle.  Accumulators are compared against the per-row
// k-th best; the (rare) survivors go through a small LDS queue.

constexpr int KNN_QCAP = 1024;


__device__ __forceinline__ bool knn_better(float s, int j, float s2, int j2) {
    return (s > s2) || (s == s2 && j < j2);
}

// order-preserving fp32 <-> unsigned key (larger score, larger key; -0.0 sorts just below +0.0)
__device__ __forceinline__ unsigned ordered_key(float v) {
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(unsigned key) {
    return __uint_as_float((key & 0x80000000u) ? (key & 0x7fffffffu) : ~key);
}

__global__ __launch_bounds__(256, 1) void sim_knn_kernel(SimKnnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int k = a.k;
    // LDS carve (all inside the one dynamic array): stages | thr[128] | qcount | queue | lists
    float* thr = reinterpret_cast<float*>(smem + GEMM_LDS);
    int* qcount = reinterpret_cast<int*>(smem + GEMM_LDS + 512);
    float* q_s = reinterpret_cast<float*>(smem + GEMM_LDS + 1024);
    int* q_j = reinterpret_cast<int*>(smem + GEMM_LDS + 1024 + KNN_QCAP * 4);
    int* q_row = reinterpret_cast<int*>(smem + GEMM_LDS + 1024 + KNN_QCAP * 8);
    float* list_s = reinterpret_cast<float*>(smem + GEMM_LDS + 1024 + KNN_QCAP * 12);
    int* list_j = reinterpret_cast<int*>(list_s + BM * k);

    // block -> (query tile, chunk): chunks of one query tile are adjacent in blockIdx so that the
    // 8 XCDs stream different reference runs while sharing the query panel through L3.
    const int tqi = blockIdx.x / a.nchunk, chunk = blockIdx.x % a.nchunk;
    const int t_begin = (int)((int64_t)a.tr * chunk / a.nchunk);
    const int t_end = (int)((int64_t)a.tr * (chunk + 1) / a.nchunk);
    const int q0 = tqi * BM;

    for (int x = tid; x < BM * k; x += 256) {
        list_s[x] = -FLT_MAX;
        list_j[x] = -1;
    }
    if (tid < BM) thr[tid] = -INFINITY;
    if (tid == 0) *qcount = 0;
    TileThread t;
    tile_thread_init(t, tid, a.dpad);
    __syncthreads();

    if (t_begin >= t_end) {
        // nothing to scan (more runs than reference tiles cannot happen, but stay safe)
        for (int x = tid; x < BM * k; x += 256) {
            const int row = x / k, e = x % k;
            const int64_t o = ((int64_t)(q0 + row) * a.nchunk + chunk) * k + e;
            a.part_s[o] = -FLT_MAX;
            a.part_j[o] = -1;
        }
        return;
    }
    const float* qpanel = a.Q + (int64_t)q0 * a.dpad;
    TileStream st;
    stream_begin(st, qpanel, a.R + (int64_t)t_begin * BN * a.dpad, a.dpad, smem, t);
    for (int tri = t_begin; tri < t_end; ++tri) {
        f32x16 acc[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
        const int r0 = tri * BN;
        stream_tile(st, tri + 1 < t_end, qpanel, tri + 1 < t_end ? a.R + (int64_t)(tri + 1) * BN * a.dpad : st.r, a.dpad,
                    smem, t, acc);

        // survivors of this tile against the thresholds as they stood before the tile
        const int rl_base = wr * 64 + 4 * (lane >> 5);
        const int col_base = r0 + wc * 64 + (lane & 31);
        if (k == 1 && tri == t_begin) {
            // k = 1, first tile of the run: every score beats the empty list and all 128 x 128 of them would go through
            // the serial insertion (the "insertion storm").  The row maxima of the tile become the
            // thresholds first -- a wave reduces each of its 32 row registers over the 32 lanes that share the row, the
            // two waves of a row meet in LDS -- and only the maxima (and their exact ties) are inserted.
            int* thr_bits = reinterpret_cast<int*>(thr);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = col_base < a.nr ? acc[m][0][r] : -INFINITY;   // (padded reference rows do not count)
                    const float v1 = col_base + 32 < a.nr ? acc[m][1][r] : -INFINITY;
                    v = fmaxf(v, v1);                                       // (fmaxf drops NaN scores: never inserted)
#pragma unroll
                    for (int off = 16; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
                    if ((lane & 31) == 0 && v > -INFINITY) {
                        const int rl = rl_base + m * 32 + (r & 3) + 8 * (r >> 2);
                        // float maximum on the bits: non-negative floats order like ints, negative ones like reversed unsigneds
                        if (v >= 0.0f) atomicMax(&thr_bits[rl], __float_as_int(v));
                        else atomicMin(reinterpret_cast<unsigned int*>(&thr_bits[rl]), __float_as_uint(v));
                    }
                }
            __syncthreads();
        } else if (tri == t_begin && k <= 32 && !a.no_first_tile_select) {
            // k > 1, first tile of the run: the same storm, 128 insertions per row for k survivors.  Any
            // lower bound of a row's k-th largest score of the tile is a valid threshold; each of the two waves that share
            // a row takes the k-th largest of ITS 64 scores of that row (k <= 64) -- a bitwise search on the order-preserving
            // key, two ballots per bit, the two rows of a register (lanes 0-31 / 32-63) side by side -- and the larger of
            // the two bounds gates the insertions: between k and 2k entries per row get through instead of 128.  The search
            // stops 8 bits early (the prefix with zero low bits is still a lower bound).  200 k x 2 M subset pass: k = 5 11.5 ->
            // 6.5 ms, k = 20 21.5 -> 16.5 ms; k = 64 gains nothing (the 64-th of a wave's 64 scores is their minimum): k <= 32.
            int* thr_bits = reinterpret_cast<int*>(thr);
            const int half_shift = lane & 32;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v0 = col_base < a.nr ? acc[m][0][r] : -INFINITY;
                    const float v1 = col_base + 32 < a.nr ? acc[m][1][r] : -INFINITY;
                    const unsigned k0 = v0 == v0 ? ordered_key(v0) : 0u;      // (NaN scores are never inserted: lowest key)
                    const unsigned k1 = v1 == v1 ? ordered_key(v1) : 0u;
                    unsigned prefix = 0;
#pragma unroll 1
                    for (int bit = 31; bit >= 8; --bit) {
                        const unsigned cand = prefix | (1u << bit);
                        const unsigned long long b0 = __ballot(k0 >= cand), b1 = __ballot(k1 >= cand);
                        const int cnt = __popc((unsigned)(b0 >> half_shift)) + __popc((unsigned)(b1 >> half_shift));
                        if (cnt >= k) prefix = cand;
                    }
                    const float v = key_to_float(prefix);
                    if ((lane & 31) == 0 && prefix > ordered_key(-INFINITY)) {
                        const int rl = rl_base + m * 32 + (r & 3) + 8 * (r >> 2);
                        if (v >= 0.0f) atomicMax(&thr_bits[rl], __float_as_int(v));
                        else atomicMin(reinterpret_cast<unsigned int*>(&thr_bits[rl]), __float_as_uint(v));
                    }
                }
            __syncthreads();
        }
        unsigned long long mask = 0;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = rl_base + m * 32 + (r & 3) + 8 * (r >> 2);
                const float th = thr[rl];
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int j = col_base + n * 32;
                    // >= : a tie with the current k-th best may still win on the lower ref index
                    const bool hit = (acc[m][n][r] >= th) && (j < a.nr);
                    mask |= (unsigned long long)hit << ((m * 2 + n) * 16 + r);
                }
            }
        // rounds: push what fits into the queue, drain, repeat while anything is left
        while (__syncthreads_or(mask != 0ull)) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int bit = (m * 2 + n) * 16 + r;
                        if ((mask >> bit) & 1ull) {
                            const int slot = atomicAdd(qcount, 1);
                            if (slot < KNN_QCAP) {
                                q_s[slot] = acc[m][n][r];
                                q_j[slot] = col_base + n * 32;
                                q_row[slot] = rl_base + m * 32 + (r & 3) + 8 * (r >> 2);
                                mask &= ~(1ull << bit);
                            }
                        }
                    }
            __syncthreads();
            const int qn = *qcount < KNN_QCAP ? *qcount : KNN_QCAP;
            // wave w owns rows with (row & 3) == w; insertions into one row are serial in a wave
            for (int base = 0; base < qn; base += 64) {
                const int e = base + lane;
                float es = 0.0f;
                int ej = 0, er = -1;
                if (e < qn) {
                    es = q_s[e];
                    ej = q_j[e];
                    er = q_row[e];
                }
                unsigned long long mine = __ballot(er >= 0 && (er & 3) == wave);
                while (mine) {
                    const int src = __ffsll((long long)mine) - 1;
                    mine &= mine - 1;
                    const float s = __shfl(es, src);
                    const int j = __shfl(ej, src);
                    const int row = __shfl(er, src);
                    float ls = -FLT_MAX;
                    int lj = 0x7fffffff;
                    if (lane < k) {
                        ls = list_s[row * k + lane];
                        lj = list_j[row * k + lane];
                        if (lj < 0) lj = 0x7fffffff;  // empty slot ranks last
                    }
                    // entries that stay ahead of the newcomer
                    const unsigned long long ahead =
                        __ballot(lane < k && knn_better(ls, lj, s, j));
                    const int pos = __popcll(ahead);
                    if (pos < k) {
                        const float us = __shfl_up(ls, 1);
                        const int uj = __shfl_up(lj, 1);
                        if (lane < k) {
                            float ns = ls;
                            int nj = lj;
                            if (lane == pos) {
                                ns = s;
                                nj = j;
                            } else if (lane > pos) {
                                ns = us;
                                nj = uj;
                            }
                            list_s[row * k + lane] = ns;
                            list_j[row * k + lane] = (nj == 0x7fffffff) ? -1 : nj;
                            if (lane == k - 1) thr[row] = (nj == 0x7fffffff) ? -INFINITY : ns;
                        }
                    }
                }
            }
            __syncthreads();
            if (tid == 0) *qcount = 0;
            // the next round's atomics are ordered after this reset by the loop's barrier
        }
    }
    __syncthreads();
    for (int x = tid; x < BM * k; x += 256) {
        const int row = x / k, e = x % k;
        const int64_t o = ((int64_t)(q0 + row) * a.nchunk + chunk) * k + e;
        a.part_s[o] = list_s[x];
        a.part_j[o] = list_j[x];
    }
}


// one wave per query row: k rounds of wave arg-best over nchunk*k candidates
__global__ __launch_bounds__(256) void knn_merge_kernel(KnnMergeArgs a) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.nq) return;
    const int n = a.nchunk * a.k;
    const float* ps = a.part_s + (int64_t)row * n;
    const int32_t* pj = a.part_j + (int64_t)row * n;
    float prev_s = INFINITY;
    int prev_j = -1;
    for (int e = 0; e < a.k; ++e) {
        // best candidate strictly after (prev_s, prev_j) in (score desc, ref asc) order
        float bs = -FLT_MAX;
        int bj = 0x7fffffff;
        for (int c = lane; c < n; c += 64) {
            const float s = ps[c];
            const int j = pj[c];
            if (j < 0) continue;
            const bool after = (s < prev_s) || (s == prev_s && j > prev_j);
            if (after && knn_better(s, j, bs, bj)) {
                bs = s;
                bj = j;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float os = __shfl_xor(bs, off);
            const int oj = __shfl_xor(bj, off);
            if (knn_better(os, oj, bs, bj)) {
                bs = os;
                bj = oj;
            }
        }
        if (lane == 0) {
            const bool none = (bj == 0x7fffffff);
            float v = none ? -FLT_MAX : bs;
            if (a.l2) v = none ? FLT_MAX : -bs;
            a.out_s[(int64_t)row * a.k + e] = v;
            a.out_j[(int64_t)row * a.k + e] = none ? -1 : (int64_t)bj;
        }
        prev_s = bs;
        prev_j = bj;
        if (bj == 0x7fffffff) prev_s = -INFINITY;
    }
}

size_t knn_lds_bytes(int k) { return (size_t)GEMM_LDS + 1024 + (size_t)KNN_QCAP * 12 + (size_t)BM * k * 8; }

int launch_sim_knn(const SimKnnArgs& a, hipStream_t stream) {
    const int64_t grid = (int64_t)a.tq * a.nchunk;
    if (grid <= 0) return VSC_OK;
    const size_t lds = knn_lds_bytes(a.k);
    static PerDeviceOnce once;
    if (once.first()) {
        // the kernel also owns a few hundred bytes of static LDS (__syncthreads_or)
        VSC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sim_knn_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
        once.commit();
    }
    hipLaunchKernelGGL(sim_knn_kernel, dim3((unsigned)grid), dim3(256), lds, stream, a);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

int launch_knn_merge(const KnnMergeArgs& a, hipStream_t stream) {
    if (a.nq <= 0) return VSC_OK;
    hipLaunchKernelGGL(knn_merge_kernel, dim3((a.nq + 3) / 4), dim3(256), 0, stream, a);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

int set_thresh_kernel_attrs() {
    static PerDeviceOnce once;
    if (once.first()) {
        VSC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sim_thresh_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS));
        once.commit();
    }
    return VSC_OK;
}

// ---------------------------------------------------------------- generic (non-MFMA) scoring
// API-completeness path for METRIC_L2 (only tests/test_index.py of the reference uses it): the
// explicit score matrix, fp32 chain in ascending k.  L2 scores are stored NEGATED so that every
// consumer keeps "larger is better".


__global__ __launch_bounds__(256) void score_matrix_kernel(ScoreMatArgs a) {
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (x >= (int64_t)a.nq * a.nr) return;
    const int i = (int)(x / a.nr), j = (int)(x % a.nr);
    const float* q = a.Q + (int64_t)i * a.dpad;
    const float* r = a.R + (int64_t)j * a.dpad;
    float acc = 0.0f;
    if (a.metric == VSC_METRIC_INNER_PRODUCT) {
        for (int k = 0; k < a.dim; ++k) acc = __fmaf_rn(q[k_slot(k)], r[k_slot(k)], acc);
    } else {
        for (int k = 0; k < a.dim; ++k) {
            const float d = q[k_slot(k)] - r[k_slot(k)];
            acc = __fmaf_rn(d, d, acc);
        }
        acc = -acc;
    }
    a.S[x] = acc;
}


__global__ __launch_bounds__(256) void matrix_thresh_kernel(MatThreshArgs a) {
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool in = x < (int64_t)a.nq * a.nr;
    const float s = in ? a.S[x] : 0.0f;
    const bool hit = in && (s > *a.radius);
    const unsigned long long m = __ballot(hit);
    if (!m) return;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(a.counter, (unsigned long long)__popcll(m));
    base = __shfl(base, 0);
    if ((long long)(base + __popcll(m)) > a.cap) {
        if (lane == 0) atomicOr(a.overflow, 1);
        return;
    }
    if (hit) {
        const unsigned long long pos = base + __popcll(m & ((1ull << lane) - 1));
        a.out_i[pos] = a.i0 + (int)(x / a.nr);
        a.out_j[pos] = (int)(x % a.nr);
        a.out_s[pos] = s;
    }
}


// one wave per row: k rounds of wave arg-best (small inputs only)
__global__ __launch_bounds__(256) void matrix_knn_kernel(MatKnnArgs a) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.nq) return;
    const float* s = a.S + (int64_t)row * a.nr;
    float prev_s = INFINITY;
    int prev_j = -1;
    for (int e = 0; e < a.k; ++e) {
        float bs = -FLT_MAX;
        int bj = 0x7fffffff;
        for (int j = lane; j < a.nr; j += 64) {
            const float v = s[j];
            const bool after = (v < prev_s) || (v == prev_s && j > prev_j);
            if (after && knn_better(v, j, bs, bj)) {
                bs = v;
                bj = j;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float os = __shfl_xor(bs, off);
            const int oj = __shfl_xor(bj, off);
            if (knn_better(os, oj, bs, bj)) {
                bs = os;
                bj = oj;
            }
        }
        if (lane == 0) {
            a.part_s[(int64_t)row * a.k + e] = (bj == 0x7fffffff) ? -FLT_MAX : bs;
            a.part_j[(int64_t)row * a.k + e] = (bj == 0x7fffffff) ? -1 : bj;
        }
        prev_s = (bj == 0x7fffffff) ? -INFINITY : bs;
        prev_j = bj;
    }
}

int launch_score_matrix(const ScoreMatArgs& a, hipStream_t stream) {
    const int64_t n = (int64_t)a.nq * a.nr;
    if (n <= 0) return VSC_OK;
    hipLaunchKernelGGL(score_matrix_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}
int launch_matrix_thresh(const MatThreshArgs& a, hipStream_t stream) {
    const int64_t n = (int64_t)a.nq * a.nr;
    if (n <= 0) return VSC_OK;
    hipLaunchKernelGGL(matrix_thresh_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}
int launch_matrix_knn(const MatKnnArgs& a, hipStream_t stream) {
    if (a.nq <= 0) return VSC_OK;
    hipLaunchKernelGGL(matrix_knn_kernel, dim3((a.nq + 3) / 4), dim3(256), 0, stream, a);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

}  // namespace vscmi
