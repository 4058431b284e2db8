// Micro-benchmark / prototype of the panel-stationary fp16 pre-filter (gfx950).
//
//   * a 128-row query panel (all of K) is resident in LDS for the lifetime of a work item,
//   * the reference rows stream STRAIGHT INTO REGISTERS from a fragment-major fp16 image (one fully
//     coalesced 1 KiB global load per MFMA B fragment): no LDS-DMA, no barrier in the steady state,
//   * 8 waves, each owns all 128 panel rows x its own 64 reference columns (4 x 2 blocks of 32x32x16).
//
// Also times MFMA-only loops (fp16 / bf16, random operands) = what the matrix pipes sustain under the
// power cap, and checks the candidates of a sub-range against a naive kernel.
//   hipcc -O3 --offload-arch=gfx950 -o f16_panel f16_panel.hip && ./f16_panel [nq] [nr]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);      \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

constexpr int PR = 128;   // panel rows
constexpr int CSW = 512;  // reference columns per col-step (8 waves x 64)

// ------------------------------------------------------------------ data
__device__ inline uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// natural layout [rows][d] fp16, elements ~ N(0, 1/d)
__global__ void gen_rows(_Float16* x, int64_t rows, int d, uint32_t seed) {
    const int64_t n = rows * d;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h = hash32((uint32_t)e * 2654435761u + seed);
        float s = 0;
        for (int t = 0; t < 4; ++t) { h = hash32(h + t); s += (h >> 8) * (1.0f / 16777216.0f); }
        x[e] = (_Float16)((s - 2.0f) * 1.7320508f * rsqrtf((float)d));
    }
}
__global__ void row_norms(const _Float16* x, int64_t rows, int d, float* nrm) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float s = 0;
    for (int k = 0; k < d; ++k) { float v = (float)x[r * d + k]; s += v * v; }
    nrm[r] = sqrtf(s) * 1.0001f;
}
// natural -> fragment-major: wave tile t = row / 64, n = (row / 32) & 1, ks = k / 16, h = (k % 16) / 8:
//   16-byte piece index = (t * nks + ks) * 128 + n * 64 + h * 32 + (row % 32)
__global__ void to_fragment_major(const _Float16* x, int64_t rows, int d, f16x8* out) {
    const int nks = d / 16;
    const int64_t npiece = rows * (d / 8);
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npiece; p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = p / (d / 8);
        const int c = (int)(p % (d / 8));  // 8-element chunk
        const int ks = c >> 1, h = c & 1;
        const int64_t t = row >> 6;
        const int n = (int)(row >> 5) & 1, jl = (int)row & 31;
        out[(t * nks + ks) * 128 + n * 64 + h * 32 + jl] = *reinterpret_cast<const f16x8*>(x + row * d + c * 8);
    }
}

// ------------------------------------------------------------------ MFMA-only loops
template <bool BF>
__global__ __launch_bounds__(512) void mfma_only(const f16x8* src, float* out, int iters) {
    const int tid = threadIdx.x;
    f16x8 a[4], b[2];
    for (int m = 0; m < 4; ++m) a[m] = src[(blockIdx.x * 6 + m) * 512 + tid];
    for (int n = 0; n < 2; ++n) b[n] = src[(blockIdx.x * 6 + 4 + n) * 512 + tid];
    f32x16 acc[4][2];
    for (int m = 0; m < 4; ++m)
        for (int n = 0; n < 2; ++n)
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    if (BF)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[m]),
                                                                           __builtin_bit_cast(bf16x8, b[n]), acc[m][n], 0, 0, 0);
                    else
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m], b[n], acc[m][n], 0, 0, 0);
                }
    }
    float s = 0;
    for (int m = 0; m < 4; ++m)
        for (int n = 0; n < 2; ++n)
            for (int r = 0; r < 16; ++r) s += acc[m][n][r];
    out[blockIdx.x * 512 + tid] = s;
}

// ------------------------------------------------------------------ panel-stationary kernel
struct PanelArgs {
    const _Float16* Q;   // natural [rows pad 128][d]
    const f16x8* Rf;     // fragment-major
    const float* qn; const float* rn;
    int d; int nq; int nr;
    int npanel;          // ceil(nq / 128)
    int nsteps;          // col-steps of 512 columns
    int slice;           // col-steps per work item
    float radius, c1, c2, c3;
    int32_t* out_i; int32_t* out_j; int seg_cap; int* seg_count; long long* ts; int* next_slice;
};

__device__ __forceinline__ float candidate_edge(float t, float eps) { return (t - eps) - 2.4e-7f * (fabsf(t) + eps); }

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, char* lds) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

// One output tile (128 panel rows x the wave's 64 columns, all of K = NKC x 128): A fragments from the LDS
// panel one k-step ahead, B fragments from a register ring of PF k-steps that is refilled PF-1 k-steps
// ahead with buffer loads (voffset = 16 * lane + {0, 1024}, soffset = position in the item's slice).  The
// stream continues into the next tile at `so_next`.  Straight-line code; every LDS address is a base
// register + immediate.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f16x8 bload(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}
template <int NKC, int PF>
__device__ __forceinline__ void tile_mma(const char* smem, const int (&abase)[8], f16x8 (&a)[4], f16x8 (&ring)[PF][2],
                                         __amdgpu_buffer_rsrc_t rs, int so_tile, int so_next, int lane16,
                                         f32x16 (&acc)[4][2]) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    constexpr int NKS = NKC * 8;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int t = ks + PF - 1;  // k-step that goes into the ring slot freed by k-step ks-1
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

__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}

// Work = (panel, slice) items.  Every panel has a counter of the slices handed out so far; a workgroup stays
// on its panel while slices are left (no panel reload, the stream continues), then helps the panel with the
// most slices left.  XCDs run at visibly different speeds (10 % spread), so a static split would leave the
// fast ones idle at the end of the launch.
template <int NKC, int PF, bool STEAL>
__global__ __launch_bounds__(512) void panel_kernel(PanelArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float qn_max_w[8];
    __shared__ int item_sh[2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NKS = NKC * 8;
    constexpr int ROWB = NKC * 256;       // bytes per fp16 row
    constexpr int TILEB = NKS * 2048;     // bytes per wave tile of the fragment-major image
    const int lane16 = lane * 16;
    int abase[8];
    {
        const int hi = lane >> 5, r15 = lane & 15, rl = lane & 31;
#pragma unroll
        for (int u = 0; u < 8; ++u) abase[u] = rl * 256 + ((((2 * u) | hi) ^ r15) << 4);
    }
    const int seg = blockIdx.x * 8 + wave;
    const int64_t seg_base = (int64_t)seg * a.seg_cap;
    int count = 0;
    const int nslice = (a.nsteps + a.slice - 1) / a.slice;
    int cur_panel = -1;
    float nq_max = 0.f;
    int panel = blockIdx.x % a.npanel;
    int static_item = blockIdx.x;
    for (;;) {
        int sl;
        if (STEAL) {
            __syncthreads();
            if (wave == 0) {
                int p = panel, s = 0;
                for (;;) {
                    if (lane == 0) s = atomicAdd(&a.next_slice[p], 1);
                    s = __shfl(s, 0);
                    if (s < nslice) break;
                    // panel exhausted: the one with the most slices left (ties: nearest after this workgroup's own)
                    int best = 0x7fffffff, bp = -1;
                    for (int q0 = 0; q0 < a.npanel; q0 += 64) {
                        const int q = q0 + lane;
                        const int pp = (q + blockIdx.x) % a.npanel;
                        const int v = q < a.npanel ? __hip_atomic_load(&a.next_slice[pp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x7fffffff;
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
            sl = item_sh[1];
            if (panel < 0) break;
        } else {
            if (static_item >= nslice * a.npanel) break;
            sl = static_item / a.npanel;
            panel = static_item - sl * a.npanel;
            static_item += gridDim.x;
        }
        const int cs0 = sl * a.slice, cs1 = min(a.nsteps, cs0 + a.slice);
        if (panel != cur_panel) {
            __syncthreads();  // everybody is done with the old panel
            // query panel -> LDS [kc][row][slot ^ (row & 15)]
            const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(
                (void*)uniform_ptr(reinterpret_cast<const char*>(a.Q) + (int64_t)panel * PR * ROWB), 0, PR * ROWB, 0x00020000);
#pragma unroll
            for (int n = 0; n < NKC * 4; ++n) {
                const int p = n * 512 + tid;
                const int kc = p >> 11, row = (p >> 4) & 127, slot = p & 15;
                const int c = kc * 16 + (slot ^ (row & 15));
                dma16(qrs, row * ROWB + c * 16, 0, smem + (n * 512 + wave * 64) * 16);
            }
            float nv = tid < PR ? a.qn[(int64_t)panel * PR + tid] : 0.f;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) nv = fmaxf(nv, __shfl_xor(nv, off));
            if (lane == 0) qn_max_w[wave] = nv;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            nq_max = fmaxf(qn_max_w[0], qn_max_w[1]);
            cur_panel = panel;
        }
        // this wave's stream: tiles (cs * 8 + wave), cs = cs0 .. cs1-1; a tile = TILEB contiguous bytes.
        // Buffer resource = the item's slice of the fragment-major image; soffset walks through it.
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)uniform_ptr(reinterpret_cast<const char*>(a.Rf) + (int64_t)cs0 * 8 * TILEB), 0,
            (cs1 - cs0) * 8 * TILEB, 0x00020000);
        int so_tile = wave * TILEB;
        f16x8 ring[PF][2];
#pragma unroll
        for (int dd = 0; dd < PF - 1; ++dd) {
            ring[dd][0] = bload(rs, lane16, so_tile + dd * 2048);
            ring[dd][1] = bload(rs, lane16 + 1024, so_tile + dd * 2048);
        }
        f16x8 afr[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) afr[m] = *reinterpret_cast<const f16x8*>(smem + abase[0] + m * 8192);
        for (int cs = cs0; cs < cs1; ++cs) {
            const int col0 = cs * CSW + wave * 64;
            if ((cs & 127) == 0 && tid == 0 && cs < 2048) a.ts[blockIdx.x * 16 + (cs >> 7)] = wall_clock64();
            // per-lane thresholds of the lane's two columns
            const float rn0 = a.rn[col0 + (lane & 31)], rn1 = a.rn[col0 + 32 + (lane & 31)];
            f32x16 acc[4][2];
            tile_mma<NKC, PF>(smem, abase, afr, ring, rs, so_tile, so_tile + 8 * TILEB, lane16, acc);
            so_tile += 8 * TILEB;
            const float e0 = (a.c1 * nq_max * rn0 + a.c2 * (nq_max + rn0) + a.c3) * 1.001f;
            const float e1 = (a.c1 * nq_max * rn1 + a.c2 * (nq_max + rn1) + a.c3) * 1.001f;
            const float thr[2] = {candidate_edge(a.radius, e0), candidate_edge(a.radius, e1)};
            // candidates are rare: one max per 32x32 block first, then one ballot per accumulator register of
            // the (few) blocks that hold one
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
            const float x0 = fmaxf(fmaxf(bm[0][0], bm[1][0]), fmaxf(bm[2][0], bm[3][0]));
            const float x1 = fmaxf(fmaxf(bm[0][1], bm[1][1]), fmaxf(bm[2][1], bm[3][1]));
            if (__any(x0 > thr[0] || x1 > thr[1])) {
                const int row_base = panel * PR + 4 * (lane >> 5);
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        if (!__any(bm[m][n] > thr[n])) continue;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const unsigned long long hits = __ballot(acc[m][n][r] > thr[n]);
                            if (hits == 0ull) continue;
                            const int i = row_base + m * 32 + (r & 3) + 8 * (r >> 2);
                            const int j = col0 + n * 32 + (lane & 31);
                            const bool c = ((hits >> lane) & 1ull) && i < a.nq && j < a.nr;
                            const unsigned long long ok = __ballot(c);
                            if (ok == 0ull) continue;
                            const int total = __popcll(ok);
                            if (count + total <= a.seg_cap) {
                                if (c) {
                                    const int64_t pos = seg_base + count + __popcll(ok & ((1ull << lane) - 1));
                                    a.out_i[pos] = i;
                                    a.out_j[pos] = j;
                                }
                                count += total;
                            }
                        }
                    }
            }
        }
    }
    if (lane == 0) a.seg_count[seg] = count;
}

// naive: counts pairs of (all q rows) x (ref rows [0, ncheck)) with fp32-accumulated fp16 score above lo / hi
__global__ void naive_count(const _Float16* Q, const _Float16* R, int nq, int ncheck, int d, float lo, float hi,
                            unsigned long long* cnt) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (int64_t)nq * ncheck) return;
    const int i = (int)(p / ncheck), j = (int)(p % ncheck);
    float s = 0;
    for (int k = 0; k < d; ++k) s += (float)Q[(int64_t)i * d + k] * (float)R[(int64_t)j * d + k];
    if (s > lo) atomicAdd(&cnt[0], 1ull);
    if (s > hi) atomicAdd(&cnt[1], 1ull);
}
__global__ void check_cands(const _Float16* Q, const _Float16* R, const int32_t* ci, const int32_t* cj, const int* seg_count,
                            int nseg, int seg_cap, int nqc, int ncheck, int d, float lo, unsigned long long* cnt) {
    const int seg = blockIdx.x;
    if (seg >= nseg) return;
    for (int c = threadIdx.x; c < seg_count[seg]; c += blockDim.x) {
        const int i = ci[(int64_t)seg * seg_cap + c], j = cj[(int64_t)seg * seg_cap + c];
        atomicAdd(&cnt[2], 1ull);
        if (j < ncheck && i < nqc) {
            atomicAdd(&cnt[3], 1ull);
            float s = 0;
            for (int k = 0; k < d; ++k) s += (float)Q[(int64_t)i * d + k] * (float)R[(int64_t)j * d + k];
            if (!(s > lo)) atomicAdd(&cnt[4], 1ull);  // a candidate that should not be one
        }
    }
}

template <int PF, bool STEAL>
double run_panel(const char* name, PanelArgs a, int grid, int reps) {
    const int lds = 4 * 32768;
    CK(hipFuncSetAttribute((const void*)panel_kernel<4, PF, STEAL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipMemsetAsync(a.next_slice, 0, 4096, 0));
    hipLaunchKernelGGL((panel_kernel<4, PF, STEAL>), dim3(grid), dim3(512), lds, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) {
        CK(hipMemsetAsync(a.next_slice, 0, 4096, 0));
        hipLaunchKernelGGL((panel_kernel<4, PF, STEAL>), dim3(grid), dim3(512), lds, 0, a);
    }
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double tf = 2.0 * a.nq * (double)a.nr * a.d / ms / 1e9;
    printf("%-44s grid=%d slice=%d radius=%.3f  %.2f ms  %.1f TFLOP/s (%.3f of 2500)\n", name, grid, a.slice, a.radius, ms, tf,
           tf / 2500.0);
    if (!STEAL) {
        // when does each XCD (blockIdx & 7) pass every 128th col-step?  wall clock, 100 MHz ticks
        std::vector<long long> ts(grid * 16);
        CK(hipMemcpy(ts.data(), a.ts, grid * 16 * 8, hipMemcpyDeviceToHost));
        const int c = std::min(15, (a.nsteps - 1) / 128);
        long long t0 = 1ll << 62;
        for (int b = 0; b < grid; ++b) t0 = std::min(t0, ts[b * 16]);
        printf("    col-step %d reached after (us), per XCD min..max over its workgroups:", c * 128);
        for (int x = 0; x < 8; ++x) {
            long long lo = 1ll << 62, hi = 0;
            for (int b = x; b < grid; b += 8) { lo = std::min(lo, ts[b * 16 + c]); hi = std::max(hi, ts[b * 16 + c]); }
            printf(" %.0f..%.0f", (lo - t0) / 100.0, (hi - t0) / 100.0);
        }
        printf("\n");
    }
    fflush(stdout);
    return tf;
}

int main(int argc, char** argv) {
    const int nq = argc > 1 ? atoi(argv[1]) : 32768;
    const int nr_req = argc > 2 ? atoi(argv[2]) : 1000000;
    const int d = 512;
    const int nsteps = (nr_req + CSW - 1) / CSW;
    const int64_t nr_pad = (int64_t)nsteps * CSW;
    const int nq_pad = (nq + PR - 1) / PR * PR;
    _Float16 *Qn, *Rn;
    f16x8* Rf;
    float *qn, *rn;
    CK(hipMalloc(&Qn, (size_t)nq_pad * d * 2));
    CK(hipMalloc(&Rn, (size_t)nr_pad * d * 2));
    CK(hipMalloc(&Rf, (size_t)nr_pad * d * 2 + (1 << 20)));
    CK(hipMalloc(&qn, (size_t)nq_pad * 4));
    CK(hipMalloc(&rn, (size_t)nr_pad * 4));
    hipLaunchKernelGGL(gen_rows, dim3(4096), dim3(256), 0, 0, Qn, (int64_t)nq_pad, d, 1u);
    hipLaunchKernelGGL(gen_rows, dim3(4096), dim3(256), 0, 0, Rn, nr_pad, d, 77u);
    hipLaunchKernelGGL(row_norms, dim3((nq_pad + 255) / 256), dim3(256), 0, 0, Qn, (int64_t)nq_pad, d, qn);
    hipLaunchKernelGGL(row_norms, dim3((unsigned)((nr_pad + 255) / 256)), dim3(256), 0, 0, Rn, nr_pad, d, rn);
    hipLaunchKernelGGL(to_fragment_major, dim3(8192), dim3(256), 0, 0, Rn, nr_pad, d, Rf);
    CK(hipDeviceSynchronize());

    const bool pmc = argc > 3 && argv[3][0] == 'p';
    const bool one = argc > 3 && argv[3][0] == 'o';  // one <radius> <slice>: a single configuration  // profiling mode: only the panel kernel (PF=4), no candidates then bench-like density
    // ---- MFMA-only ceilings
    if (!pmc && !one) {
        float* out;
        CK(hipMalloc(&out, 256 * 512 * 4));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (int bf = 0; bf < 2; ++bf) {
            const int iters = 200000;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                if (bf) hipLaunchKernelGGL(mfma_only<true>, dim3(256), dim3(512), 0, 0, Rf, out, iters);
                else hipLaunchKernelGGL(mfma_only<false>, dim3(256), dim3(512), 0, 0, Rf, out, iters);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                const double fl = 256.0 * 8 * iters * 32.0 * 2 * 32 * 32 * 16;
                printf("mfma-only %s rep %d: %.1f ms  %.1f TFLOP/s\n", bf ? "bf16" : "fp16", rep, ms, fl / ms / 1e9);
            }
        }
        fflush(stdout);
    }

    PanelArgs a;
    a.Q = Qn; a.Rf = Rf; a.qn = qn; a.rn = rn; a.d = d; a.nq = nq; a.nr = nr_req;
    a.npanel = nq_pad / PR; a.nsteps = nsteps; a.slice = 32;
    a.radius = 0.154f;  // ~3.5 sigma: candidate density ~2.3e-4 like the bench's big batches
    const double D = d;
    a.c1 = (float)(ldexp(1.0, -10) + ldexp(1.0, -22) + (2.0 * D + D / 16.0 + 16.0) * ldexp(1.0, -23));
    a.c2 = (float)(ldexp(1.0, -25) * 1.001 * sqrt(D));
    a.c3 = (float)(D * ldexp(1.0, -50));
    const int grid = 256;
    a.seg_cap = 1 << 16;
    CK(hipMalloc(&a.out_i, (size_t)grid * 8 * a.seg_cap * 4));
    CK(hipMalloc(&a.out_j, (size_t)grid * 8 * a.seg_cap * 4));
    CK(hipMalloc(&a.seg_count, grid * 8 * 4));
    CK(hipMalloc(&a.next_slice, 4096));
    CK(hipMalloc(&a.ts, grid * 16 * 8));
    CK(hipMemset(a.ts, 0, grid * 16 * 8));

    if (one) {
        a.radius = atof(argv[4]);
        a.slice = atoi(argv[5]);
        run_panel<8, true>("panel PF=8 steal", a, grid, 3);
        run_panel<8, false>("panel PF=8 static", a, grid, 3);
        return 0;
    }
    if (pmc) {
        a.radius = 0.165f;
        run_panel<8, true>("panel PF=8 steal density 1e-4", a, grid, 2);
        return 0;
    }
    run_panel<8, true>("panel PF=8 steal", a, grid, 3);
    // correctness of the candidates over ref rows [0, ncheck)
    {
        const int ncheck = 2048, nqc = nq < 4096 ? nq : 4096;
        unsigned long long* cnt;
        CK(hipMalloc(&cnt, 5 * 8));
        CK(hipMemset(cnt, 0, 5 * 8));
        const float eps = a.c1 * 1.1f * 1.1f + 0.001f;
        hipLaunchKernelGGL(naive_count, dim3((unsigned)(((int64_t)nqc * ncheck + 255) / 256)), dim3(256), 0, 0, Qn, Rn, nqc,
                           ncheck, d, a.radius - 2 * eps, a.radius + eps * 0, cnt);
        hipLaunchKernelGGL(check_cands, dim3(grid * 8), dim3(64), 0, 0, Qn, Rn, a.out_i, a.out_j, a.seg_count, grid * 8,
                           a.seg_cap, nqc, ncheck, d, a.radius - 2 * eps, cnt);
        unsigned long long h[5];
        CK(hipMemcpy(h, cnt, 40, hipMemcpyDeviceToHost));
        printf("check: naive(q<%d, r<%d): > radius-2eps %llu, > radius %llu | candidates total %llu (density %.3g), in r<%d: %llu, bogus %llu\n",
               nqc, ncheck, h[0], h[1], h[2], (double)h[2] / ((double)nq * nr_req), ncheck, h[3], h[4]);
        // the kernel's candidates in the checked region restricted to q < nqc are not separated: compare only when nqc == nq
        printf("  -> %s\n", (h[3] >= h[1] && h[3] <= h[0] && h[4] == 0) ? "OK" : "MISMATCH");
        fflush(stdout);
    }
    a.radius = 0.4f;
    run_panel<8, false>("panel PF=8 static no candidates", a, grid, 3);
    run_panel<8, true>("panel PF=8 steal no candidates", a, grid, 3);
    run_panel<4, true>("panel PF=4 steal no candidates", a, grid, 3);
    a.slice = 16;
    run_panel<8, true>("panel PF=8 steal no candidates", a, grid, 3);
    a.slice = 64;
    run_panel<8, true>("panel PF=8 steal no candidates", a, grid, 3);
    a.slice = 32;
    a.radius = 0.165f;
    run_panel<8, false>("panel PF=8 static density 1e-4", a, grid, 3);
    run_panel<8, true>("panel PF=8 steal density 1e-4", a, grid, 3);
    a.radius = 0.154f;
    run_panel<8, true>("panel PF=8 steal density 2.8e-4", a, grid, 3);
    return 0;
}
