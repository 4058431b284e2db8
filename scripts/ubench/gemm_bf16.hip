// Micro-benchmark / development bed of the bf16 pre-filter GEMM (gfx950).
//   C[i][j] = sum_k A[i][k] * B[j][k]   (both operands row-major, K contiguous), bf16 in, fp32 acc.
// 256x256 tile per workgroup, 8 waves (2 x 4), wave tile 128 x 64 = 4 x 2 blocks of
// v_mfma_f32_32x32x16_bf16; operands staged by LDS-DMA into an XOR-swizzled image.
// Output: count of C > thr per launch (the pre-filter's job) or, for checking, the full C.
//
//   hipcc -O3 --offload-arch=gfx950 gemm_bf16.hip -o gemm_bf16 && ./gemm_bf16
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int BM = 256, BN = 256;
constexpr int BK = 64;                       // bf16 elements per K-tile
constexpr int ROWB = BK * 2;                 // bytes per row per K-tile (128)
constexpr int TILE_BYTES = BM * ROWB;        // 32 KiB per operand
constexpr int STAGE_BYTES = 2 * TILE_BYTES;  // A + B
constexpr int NSTAGE = 2;
constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, char* lds) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff,
                                             soff, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const void* base, int row_bytes) {
    const uint64_t p = reinterpret_cast<uint64_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
    void* up = reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(up, 0, BM * row_bytes, 0x00020000);
}

struct TT {
    int src_off[4];  // byte offset of the 4 DMA pieces (per operand) inside the tile's rows
    int dst_off[4];  // wave-uniform LDS byte offset
    int rdA[4];      // LDS byte offset of A block mb, k-step 0 (k-step ks: ^ (ks << 5))
    int rdB[2];
};

__device__ __forceinline__ void tt_init(TT& t, int tid, int row_bytes) {
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int p = n * 512 + tid;  // 16-byte slot in the operand image (2048 slots)
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
    bf16x8 a[4], b[2];
};

__device__ __forceinline__ Frags read_frags(const char* stage, const TT& t, int ks) {
    Frags f;
#pragma unroll
    for (int m = 0; m < 4; ++m) f.a[m] = *reinterpret_cast<const bf16x8*>(stage + (t.rdA[m] ^ (ks << 5)));
#pragma unroll
    for (int n = 0; n < 2; ++n)
        f.b[n] = *reinterpret_cast<const bf16x8*>(stage + TILE_BYTES + (t.rdB[n] ^ (ks << 5)));
    return f;
}

__device__ __forceinline__ void mfma8(const Frags& f, f32x16 (&acc)[4][2]) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[m], f.b[n], acc[m][n], 0, 0, 0);
}

__device__ __forceinline__ void stage_tiles(__amdgpu_buffer_rsrc_t ars, __amdgpu_buffer_rsrc_t brs, int kt,
                                            char* stage, const TT& t) {
#pragma unroll
    for (int n = 0; n < 4; ++n) dma16(ars, t.src_off[n], kt * ROWB, stage + t.dst_off[n]);
#pragma unroll
    for (int n = 0; n < 4; ++n) dma16(brs, t.src_off[n], kt * ROWB, stage + TILE_BYTES + t.dst_off[n]);
}

struct Args {
    const __bf16* A;
    const __bf16* B;
    int M, N, K;  // M, N multiples of 256; K multiple of 128
    float thr;
    unsigned long long* count;
    float* C;  // optional dump (row-major M x N)
};

// one K-loop over the current tile; 2-stage ring, K-tile kt in stage kt & 1
__device__ __forceinline__ void tile_gemm(const Args& a, int tm, int tn, char* smem, const TT& t,
                                          f32x16 (&acc)[4][2]) {
    const int row_bytes = a.K * 2;
    const __amdgpu_buffer_rsrc_t ars = tile_rsrc(a.A + (size_t)tm * BM * a.K, row_bytes);
    const __amdgpu_buffer_rsrc_t brs = tile_rsrc(a.B + (size_t)tn * BN * a.K, row_bytes);
    const int nkt = a.K / BK;
    stage_tiles(ars, brs, 0, smem, t);
    __syncthreads();
    stage_tiles(ars, brs, 1, smem + STAGE_BYTES, t);
    Frags cur = read_frags(smem, t, 0);
    int sp = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        const char* stage = smem + sp * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const Frags nxt = read_frags(stage, t, ks + 1);
            mfma8(cur, acc);
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            cur = nxt;
        }
        const bool n1 = kt + 1 < nkt, n2 = kt + 2 < nkt;
        Frags nxt = cur;
        if (n1) {
            __syncthreads();
            nxt = read_frags(smem + (sp ^ 1) * STAGE_BYTES, t, 0);
        }
        char* wstage = smem + sp * STAGE_BYTES;
        const int soff = (kt + 2) * ROWB;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.a[m], cur.b[0], acc[m][0], 0, 0, 0);
            if (n2) dma16(ars, t.src_off[m], soff, wstage + t.dst_off[m]);
            acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.a[m], cur.b[1], acc[m][1], 0, 0, 0);
            if (n2) dma16(brs, t.src_off[m], soff, wstage + TILE_BYTES + t.dst_off[m]);
        }
        cur = nxt;
        sp ^= 1;
    }
}

__global__ __launch_bounds__(512, 1) void gemm_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    TT t;
    tt_init(t, tid, a.K * 2);
    const int tmn = a.M / BM, tnn = a.N / BN;
    // XCD-aware: workgroup b runs on XCD b % 8; each XCD owns a contiguous run of tiles, bands of 8 M-tiles
    const int64_t nblk = (int64_t)tmn * tnn;
    const int xcd = blockIdx.x & 7;
    const int64_t per_xcd = (nblk + 7) / 8;
    const int64_t lstride = gridDim.x >> 3;
    unsigned long long cnt = 0;
    for (int64_t local = blockIdx.x >> 3; local < per_xcd; local += lstride) {
        const int64_t logical = xcd * per_xcd + local;
        if (logical >= nblk) break;
        constexpr int GQ = 4;
        const int64_t band_sz = (int64_t)GQ * tnn;
        const int64_t band = logical / band_sz, rem = logical % band_sz;
        const int q0 = (int)band * GQ;
        const int gq = (tmn - q0) < GQ ? (tmn - q0) : GQ;
        const int tn = (int)(rem / gq), tm = q0 + (int)(rem % gq);
        f32x16 acc[4][2];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
        tile_gemm(a, tm, tn, smem, t, acc);
        // epilogue: running max, one compare
        float mx = -INFINITY;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; r += 2) mx = __builtin_fmaxf(mx, __builtin_fmaxf(acc[m][n][r], acc[m][n][r + 1]));
        if (__any(mx > a.thr)) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) cnt += acc[m][n][r] > a.thr;
        }
        if (a.C) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = tm * BM + wr * 128 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        const int col = tn * BN + wc * 64 + n * 32 + (lane & 31);
                        a.C[(size_t)row * a.N + col] = acc[m][n][r];
                    }
        }
        __syncthreads();  // LDS ring restarts
    }
    if (cnt) atomicAdd(a.count, cnt);
}

static inline uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fff + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}
static inline float bf2f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 32768, N = argc > 2 ? atoi(argv[2]) : 262144, K = argc > 3 ? atoi(argv[3]) : 512;
    int grid = argc > 4 ? atoi(argv[4]) : 0;
    const bool check = M * (size_t)N <= (size_t)1 << 22;
    std::vector<uint16_t> hA((size_t)M * K), hB((size_t)N * K);
    srand(1);
    auto fill = [&](std::vector<uint16_t>& v, int rows) {
        // unit-norm gaussian-ish rows (sum of 4 uniforms), like L2-normalised descriptors
        for (int r = 0; r < rows; ++r) {
            double ss = 0;
            std::vector<float> row(K);
            for (int k = 0; k < K; ++k) {
                float x = 0;
                for (int u = 0; u < 4; ++u) x += (float)rand() / RAND_MAX - 0.5f;
                row[k] = x;
                ss += (double)x * x;
            }
            const float inv = (float)(1.0 / sqrt(ss));
            for (int k = 0; k < K; ++k) v[(size_t)r * K + k] = f2bf(row[k] * inv);
        }
    };
    fill(hA, M);
    fill(hB, N);
    __bf16 *dA, *dB;
    float* dC = nullptr;
    unsigned long long* dcnt;
    hipMalloc(&dA, hA.size() * 2);
    hipMalloc(&dB, hB.size() * 2);
    hipMalloc(&dcnt, 8);
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
    if (check) hipMalloc(&dC, (size_t)M * N * 4);
    hipFuncSetAttribute((const void*)gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    const int64_t nblk = (int64_t)(M / BM) * (N / BN);
    if (grid <= 0) grid = 256;
    if (grid > nblk) grid = (int)((nblk + 7) / 8 * 8);
    Args a{dA, dB, M, N, K, 0.18f, dcnt, dC};
    hipMemset(dcnt, 0, 8);
    hipLaunchKernelGGL(gemm_kernel, dim3(grid), dim3(512), LDS_BYTES, 0, a);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 1; }
    unsigned long long cnt = 0;
    hipMemcpy(&cnt, dcnt, 8, hipMemcpyDeviceToHost);
    if (check) {
        std::vector<float> C((size_t)M * N);
        hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        double maxerr = 0;
        unsigned long long ref_cnt = 0;
        for (int i = 0; i < M; ++i)
            for (int j = 0; j < N; ++j) {
                double s = 0;
                for (int k = 0; k < K; ++k) s += (double)bf2f(hA[(size_t)i * K + k]) * bf2f(hB[(size_t)j * K + k]);
                maxerr = fmax(maxerr, fabs(s - C[(size_t)i * N + j]));
                ref_cnt += C[(size_t)i * N + j] > a.thr;
            }
        printf("check %dx%dx%d: max |err| = %.3g, count %llu (ref %llu)\n", M, N, K, maxerr, cnt, ref_cnt);
        return maxerr < 1e-4 ? 0 : 2;
    }
    a.C = nullptr;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(dcnt, 0, 8);
        hipEventRecord(e0);
        hipLaunchKernelGGL(gemm_kernel, dim3(grid), dim3(512), LDS_BYTES, 0, a);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&cnt, dcnt, 8, hipMemcpyDeviceToHost);
        printf("%dx%dx%d grid %d: %.3f ms  %.1f TFLOP/s  (count %llu = %.3g of pairs)\n", M, N, K, grid, ms,
               2.0 * M * N * K / ms * 1e-9, cnt, (double)cnt / ((double)M * N));
    }
    return 0;
}
