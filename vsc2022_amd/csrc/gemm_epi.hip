// 1x1 convolution of the folded SSCD trunk as ONE kernel (config 3, frame inference):
//     out[m, n] = act( sum_k a[m, k] w[n, k] + bias[n] (+ res[m, n]) )        a, w, res, out bf16; bias fp32
// a = NHWC activations seen as [M = N*H*W, K = Cin], w = the convolution's weight [N = Cout, K] as PyTorch stores it.
// These products are HBM-bound at batch 256 (K = 64 ... 2048 against ~1 GB of activations moved per call): the point
// of the kernel is that the epilogue (bias, identity, ReLU) costs no extra pass over the output -- stock PyTorch
// spends three, FastSSCD's first version one (vsc_bias_act_bf16).
//
// v_mfma_f32_32x32x16_bf16 with the WEIGHTS as the row operand and the activations as the column operand: a lane then
// owns one output row m and 4 x 4 consecutive channels per 32 x 32 block (8-byte stores, 8-byte identity loads), and
// both operands are 16 contiguous bytes per lane straight from global memory (row-major [rows, K] on both sides): no
// LDS, no transposes.  A wave computes 64 rows x 64 channels; the 4 waves of a workgroup sit side by side along the
// channels (N >= 256), so a row of `a` is fetched from HBM once and served to the other waves by L1 / L2.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "../../include/vscmi.h"
#include "vscmi_common.h"

namespace vscmi {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    const unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// WN waves along the channels (64 each), 4 / WN along the rows (64 each)
template <int WN, bool RES, bool RELU>
__global__ __launch_bounds__(256) void gemm_bias_act_bf16_kernel(const __bf16* __restrict__ a, const __bf16* __restrict__ w,
                                                                 const float* __restrict__ bias,
                                                                 const unsigned short* __restrict__ res,
                                                                 unsigned short* __restrict__ out, long long M, int N, int K) {
    constexpr int WM = 4 / WN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wn = wave % WN, wm = wave / WN;
    const long long m0 = ((long long)blockIdx.x * WM + wm) * 64;
    const int n0 = (blockIdx.y * WN + wn) * 64;
    if (m0 >= M) return;
    const int l31 = lane & 31, kh = (lane >> 5) * 8;
    // rows past the end read row M - 1 (their results are not stored)
    const long long ra0 = std::min<long long>(m0 + l31, M - 1), ra1 = std::min<long long>(m0 + 32 + l31, M - 1);
    const bf16x8* pa0 = reinterpret_cast<const bf16x8*>(a + ra0 * K + kh);
    const bf16x8* pa1 = reinterpret_cast<const bf16x8*>(a + ra1 * K + kh);
    const bf16x8* pw0 = reinterpret_cast<const bf16x8*>(w + (long long)(n0 + l31) * K + kh);
    const bf16x8* pw1 = reinterpret_cast<const bf16x8*>(w + (long long)(n0 + 32 + l31) * K + kh);
    f32x16 acc[2][2];  // [channel block][row block]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    for (int k0 = 0; k0 < K; k0 += 64) {  // K % 64 == 0; 16 elements = 2 pieces of 8 per k-step
        bf16x8 fa0[4], fa1[4], fw0[4], fw1[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int p = (k0 >> 3) + 2 * s;
            fa0[s] = pa0[p];
            fa1[s] = pa1[p];
            fw0[s] = pw0[p];
            fw1[s] = pw1[p];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw0[s], fa0[s], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw0[s], fa1[s], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw1[s], fa0[s], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw1[s], fa1[s], acc[1][1], 0, 0, 0);
        }
    }
    // epilogue.  In the accumulators a lane owns ONE row and 4 x 4 channels per block: stored like that, a wave's store
    // touches 32 rows with 16 bytes each.  So the tile goes through LDS (fp32, wave-private, 32 rows at a time, row
    // stride 68 floats: conflict-free for both access patterns) and comes back as 16-byte pieces of whole rows -- 8
    // lanes per 128-byte row segment, for the identity loads as well as for the stores.
    __shared__ float epi[4][32 * 68];
    float* tile = epi[wave];
    const int hi = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the previous half has been read
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
                *reinterpret_cast<f32x4*>(tile + l31 * 68 + 32 * i + 8 * g + 4 * hi) = v;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int c8 = (lane & 7) * 8;
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + n0 + c8), b1 = *reinterpret_cast<const f32x4*>(bias + n0 + c8 + 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int r = (lane >> 3) + 8 * t;
            const long long m = m0 + 32 * j + r;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(tile + r * 68 + c8), v1 = *reinterpret_cast<const f32x4*>(tile + r * 68 + c8 + 4);
            if (m >= M) continue;
            u16x8 rv;
            if (RES) rv = *reinterpret_cast<const u16x8*>(res + m * N + n0 + c8);
            u16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = (e < 4 ? v0[e] : v1[e - 4]) + (e < 4 ? b0[e] : b1[e - 4]);
                if (RES) f += bf16_bits_to_f32(rv[e]);
                if (RELU) f = f > 0.0f ? f : (f == f ? 0.0f : f);
                o[e] = f32_to_bf16_bits(f);
            }
            *reinterpret_cast<u16x8*>(out + m * N + n0 + c8) = o;
        }
    }
}

template <int WN>
static void launch_gemm(const void* a, const void* w, const float* bias, const void* res, void* out, long long M, int N, int K,
                        int relu, hipStream_t s) {
    constexpr int WM = 4 / WN;
    const dim3 grid((unsigned)((M + 64 * WM - 1) / (64 * WM)), (unsigned)(N / (64 * WN)));
    const __bf16* aa = (const __bf16*)a;
    const __bf16* ww = (const __bf16*)w;
    const unsigned short* rr = (const unsigned short*)res;
    unsigned short* oo = (unsigned short*)out;
    if (res && relu) hipLaunchKernelGGL((gemm_bias_act_bf16_kernel<WN, true, true>), grid, dim3(256), 0, s, aa, ww, bias, rr, oo, M, N, K);
    else if (res) hipLaunchKernelGGL((gemm_bias_act_bf16_kernel<WN, true, false>), grid, dim3(256), 0, s, aa, ww, bias, rr, oo, M, N, K);
    else if (relu) hipLaunchKernelGGL((gemm_bias_act_bf16_kernel<WN, false, true>), grid, dim3(256), 0, s, aa, ww, bias, rr, oo, M, N, K);
    else hipLaunchKernelGGL((gemm_bias_act_bf16_kernel<WN, false, false>), grid, dim3(256), 0, s, aa, ww, bias, rr, oo, M, N, K);
}

}  // namespace vscmi

extern "C" int vsc_gemm_bias_act_bf16(const void* a, const void* w, const float* bias, const void* res, void* out,
                                      int64_t M, int64_t N, int64_t K, int relu, void* hip_stream) {
    using namespace vscmi;
    if (!a || !w || !bias || !out || M < 0 || N <= 0 || K <= 0 || (N & 63) || (K & 63) || N > (1 << 20) || K > (1 << 20) ||
        (((uintptr_t)a | (uintptr_t)w | (uintptr_t)bias | (uintptr_t)res | (uintptr_t)out) & 15)) {
        set_error("vsc_gemm_bias_act_bf16: invalid argument (N and K must be multiples of 64, pointers 16-byte aligned)");
        return VSC_ERR_INVALID;
    }
    if (M == 0) return VSC_OK;
    hipStream_t s = (hipStream_t)hip_stream;
    if (N % 256 == 0) launch_gemm<4>(a, w, bias, res, out, M, (int)N, (int)K, relu, s);
    else if (N % 128 == 0) launch_gemm<2>(a, w, bias, res, out, M, (int)N, (int)K, relu, s);
    else launch_gemm<1>(a, w, bias, res, out, M, (int)N, (int)K, relu, s);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}
