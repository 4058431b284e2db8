// Micro-benchmark: what do the matrix pipes of an MI355X sustain on FP6 (E2M3) against int8 under this pool's power cap,
// and is the block-scaled MFMA's fp32 accumulation of FP6 products exact?  (round 5: the k-NN / 1-NN threshold passes'
// pre-filter -- thresholds 4.6 ... 5 sigma above the mean, where a bound 3.7x looser than int8's still passes < 4e-4 of the matrix.)
//   v_mfma_scale_f32_16x16x128_f8f6f4 / 32x32x64 with cbsz = blgp = 2 (FP6 E2M3), unit block scales (E8M0 127).
// Operands: E2M3 codes of N(0, 1.9^2) values (what a per-row scale "largest element -> 7.5" makes of descriptor rows).
//   hipcc -O3 --offload-arch=gfx950 -o fp6_mfma fp6_mfma.hip && ./fp6_mfma
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

__host__ __device__ inline uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// E2M3: sign, 2 exponent bits (bias 1), 3 mantissa bits; value in units of 1/8: e = 0: m; e > 0: (8 + m) << (e - 1)
__host__ __device__ inline int e2m3_units(int code) {
    const int e = (code >> 3) & 3, m = code & 7;
    const int mag = e == 0 ? m : (8 + m) << (e - 1);
    return (code & 32) ? -mag : mag;
}
__host__ __device__ inline int e2m3_encode(float v) {  // round to nearest (ties away), saturating at 7.5
    const float a = fminf(fabsf(v), 7.5f);
    int best = 0;
    float bd = 1e9f;
    for (int c = 0; c < 32; ++c) {
        const float d = fabsf(a - 0.125f * e2m3_units(c));
        if (d < bd) { bd = d; best = c; }
    }
    return best | (v < 0 ? 32 : 0);
}
// n6 elements -> packed 6-bit codes, element e at bits [6 e, 6 e + 6) of the little-endian bit string
__global__ void gen_fp6(uint32_t* x, int64_t ndw, uint32_t seed) {
    // one thread per group of 16 elements = 3 dwords
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g * 3 + 2 < ndw; g += (int64_t)gridDim.x * blockDim.x) {
        unsigned long long lo = 0, hi = 0;  // 96 bits
        for (int e = 0; e < 16; ++e) {
            const uint32_t h = hash32((uint32_t)(g * 16 + e) * 2654435761u + seed);
            const uint32_t h2 = hash32(h ^ 0x9e3779b9u);
            // Box-Muller
            const float u1 = ((h >> 8) + 1) * (1.0f / 16777217.0f), u2 = (h2 >> 8) * (1.0f / 16777216.0f);
            const float z = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
            const unsigned long long c = (unsigned long long)e2m3_encode(1.9f * z);
            const int bit = 6 * e;
            if (bit < 64) { lo |= c << bit; if (bit > 58) hi |= c >> (64 - bit); } else hi |= c << (bit - 64);
        }
        x[g * 3] = (uint32_t)lo; x[g * 3 + 1] = (uint32_t)(lo >> 32); x[g * 3 + 2] = (uint32_t)hi;
    }
}
__global__ void gen_i8(int8_t* x, int64_t n, uint32_t seed) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t h = hash32((uint32_t)e * 2654435761u + seed), h2 = hash32(h ^ 0x9e3779b9u);
        const float u1 = ((h >> 8) + 1) * (1.0f / 16777217.0f), u2 = (h2 >> 8) * (1.0f / 16777216.0f);
        const float z = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
        x[e] = (int8_t)fminf(fmaxf(rintf(40.f * z), -127.f), 127.f);
    }
}

constexpr int FMT_FP6 = 2;  // cbsz / blgp: 0 fp8 e4m3, 1 bf8 e5m2, 2 fp6 e2m3, 3 bf6 e3m2, 4 fp4 e2m1

// MODE 0: i32_16x16x64_i8   1: scale_f32_16x16x128 fp6   2: scale_f32_32x32x64 fp6   3: 16x16x128 fp4   4: 16x16x128 fp8
template <int MODE>
__global__ __launch_bounds__(512) void mfma_only(const int* src, float* out, int iters, int scale) {
    const int tid = threadIdx.x;
    i32x8 a[4], b[2];
    for (int m = 0; m < 4; ++m)
        for (int r = 0; r < 8; ++r) a[m][r] = src[((blockIdx.x * 6 + m) * 8 + r) * 512 + tid];
    for (int n = 0; n < 2; ++n)
        for (int r = 0; r < 8; ++r) b[n][r] = src[((blockIdx.x * 6 + 4 + n) * 8 + r) * 512 + tid];
    float s = 0;
    if (MODE == 0) {
        i32x4 acc[4][2] = {};
        i32x4 a4[4], b4[2];
        for (int m = 0; m < 4; ++m) a4[m] = i32x4{a[m][0], a[m][1], a[m][2], a[m][3]};
        for (int n = 0; n < 2; ++n) b4[n] = i32x4{b[n][0], b[n][1], b[n][2], b[n][3]};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4[m], b4[n], acc[m][n], 0, 0, 0);
        for (int m = 0; m < 4; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 4; ++r) s += acc[m][n][r];
    } else if (MODE == 2) {
        f32x16 acc[4][2] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[m], b[n], acc[m][n], FMT_FP6, FMT_FP6, 0, scale, 0, scale);
        for (int m = 0; m < 4; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) s += acc[m][n][r];
    } else {
        constexpr int F = MODE == 1 ? 2 : MODE == 3 ? 4 : 0;
        f32x4 acc[4][2] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[m], b[n], acc[m][n], F, F, 0, scale, 0, scale);
        for (int m = 0; m < 4; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 4; ++r) s += acc[m][n][r];
    }
    out[blockIdx.x * 512 + tid] = s;
}

// Exactness + operand layout: one wave, D = A (16 x 128) . B^T (16 x 128), rows packed as 96 bytes of 6-bit codes.
// Assumed layout (as for the other 16x16 shapes): lane l holds row l & 15, k = 32 (l >> 4) ... + 31 = 24 bytes = 6 dwords.
__global__ void one_mfma(const uint32_t* A, const uint32_t* B, float* D, int scale, int chain) {
    const int lane = threadIdx.x;
    i32x8 a = {}, b = {};
    for (int r = 0; r < 6; ++r) {
        a[r] = A[(lane & 15) * 24 + (lane >> 4) * 6 + r];
        b[r] = B[(lane & 15) * 24 + (lane >> 4) * 6 + r];
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < chain; ++c)  // the same product accumulated `chain` times: sums up to chain x 128 terms
        acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, FMT_FP6, FMT_FP6, 0, scale, 0, scale);
    for (int r = 0; r < 4; ++r) D[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[r];  // D[row of A][row of B]
}

int main(int argc, char** argv) {
    const int64_t ndw = 256 * 6 * 8 * 512 + 1024;
    int *src6, *src8;
    CK(hipMalloc(&src6, ndw * 4));
    CK(hipMalloc(&src8, ndw * 4));
    hipLaunchKernelGGL(gen_fp6, dim3(2048), dim3(256), 0, 0, (uint32_t*)src6, ndw, 7u);
    hipLaunchKernelGGL(gen_i8, dim3(2048), dim3(256), 0, 0, (int8_t*)src8, ndw * 4, 9u);
    CK(hipDeviceSynchronize());
    // ---- exactness
    {
        uint32_t *A, *B;
        float* D;
        CK(hipMalloc(&A, 16 * 24 * 4));
        CK(hipMalloc(&B, 16 * 24 * 4));
        CK(hipMalloc(&D, 256 * 4));
        long long bad = 0, total = 0;
        double worst = 0;
        std::vector<uint32_t> hA(16 * 24), hB(16 * 24);
        std::vector<float> hD(256);
        for (int trial = 0; trial < 200; ++trial) {
            const int chain = trial % 4 == 3 ? 4 : 1;
            for (int i = 0; i < 16 * 24; ++i) {
                // trial % 3 == 0: uniform random codes (many large magnitudes); else Gaussian-like codes
                if (trial % 3 == 0) { hA[i] = hash32(i * 31 + trial * 977 + 1); hB[i] = hash32(i * 17 + trial * 131 + 5); }
            }
            if (trial % 3 != 0) {
                auto fill = [&](std::vector<uint32_t>& h, uint32_t seed) {
                    for (int row = 0; row < 16; ++row) {
                        unsigned char bits[96] = {0};
                        for (int e = 0; e < 128; ++e) {
                            const uint32_t x = hash32(seed + row * 128 + e), y = hash32(x ^ 0xabcdef);
                            const float u1 = ((x >> 8) + 1) * (1.0f / 16777217.0f), u2 = (y >> 8) * (1.0f / 16777216.0f);
                            const int c = e2m3_encode(1.9f * sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2));
                            for (int bb = 0; bb < 6; ++bb)
                                if (c >> bb & 1) bits[(6 * e + bb) >> 3] |= 1 << ((6 * e + bb) & 7);
                        }
                        std::memcpy(&h[row * 24], bits, 96);
                    }
                };
                fill(hA, trial * 1000003u);
                fill(hB, trial * 7000003u + 11);
            }
            CK(hipMemcpy(A, hA.data(), 16 * 24 * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(B, hB.data(), 16 * 24 * 4, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(one_mfma, dim3(1), dim3(64), 0, 0, A, B, D, 0x7f7f7f7f, chain);
            CK(hipMemcpy(hD.data(), D, 256 * 4, hipMemcpyDeviceToHost));
            auto code = [](const std::vector<uint32_t>& h, int row, int e) {
                const unsigned char* p = (const unsigned char*)&h[row * 24];
                int c = 0;
                for (int bb = 0; bb < 6; ++bb) c |= ((p[(6 * e + bb) >> 3] >> ((6 * e + bb) & 7)) & 1) << bb;
                return c;
            };
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j) {
                    long long ref = 0;
                    for (int e = 0; e < 128; ++e) ref += (long long)e2m3_units(code(hA, i, e)) * e2m3_units(code(hB, j, e));
                    ref *= chain;
                    const double got = (double)hD[i * 16 + j] * 64.0;
                    ++total;
                    if (got != (double)ref) {
                        ++bad;
                        worst = fmax(worst, fabs(got - (double)ref));
                        if (bad <= 5) printf("  mismatch trial %d (%d,%d): got %.3f ref %lld\n", trial, i, j, got, ref);
                    }
                }
        }
        printf("fp6 16x16x128 exactness (layout: lane = row + 16 x k-quarter, 24 contiguous bytes): %lld / %lld mismatches, worst |diff| %.3f units of 1/64\n",
               bad, total, worst);
        fflush(stdout);
    }
    // ---- rates
    float* out;
    CK(hipMalloc(&out, 256 * 512 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char* names[5] = {"i32_16x16x64_i8", "f32_16x16x128_fp6", "f32_32x32x64_fp6", "f32_16x16x128_fp4", "f32_16x16x128_fp8"};
    const double per[5] = {2.0 * 16 * 16 * 64, 2.0 * 16 * 16 * 128, 2.0 * 32 * 32 * 64, 2.0 * 16 * 16 * 128, 2.0 * 16 * 16 * 128};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 5; ++mode) {
            const int iters = 40000;
            CK(hipEventRecord(e0));
            const int sc = 0x7f7f7f7f;
            if (mode == 0) hipLaunchKernelGGL(mfma_only<0>, dim3(256), dim3(512), 0, 0, src8, out, iters, sc);
            if (mode == 1) hipLaunchKernelGGL(mfma_only<1>, dim3(256), dim3(512), 0, 0, src6, out, iters, sc);
            if (mode == 2) hipLaunchKernelGGL(mfma_only<2>, dim3(256), dim3(512), 0, 0, src6, out, iters, sc);
            if (mode == 3) hipLaunchKernelGGL(mfma_only<3>, dim3(256), dim3(512), 0, 0, src6, out, iters, sc);
            if (mode == 4) hipLaunchKernelGGL(mfma_only<4>, dim3(256), dim3(512), 0, 0, src6, out, iters, sc);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("mfma-only %-20s rep %d: %8.1f ms  %8.1f TOP/s\n", names[mode], rep, ms, 256.0 * 8 * iters * 32.0 * per[mode] / ms / 1e9);
            fflush(stdout);
        }
    return 0;
}
