// Epilogue of the folded SSCD trunk's convolutions (config 3, frame inference): y = act(y + bias[c] (+ res)), bf16
// activations in NHWC memory seen as a [rows, cols] matrix (rows = N*H*W, cols = channels), fp32 arithmetic, one
// rounding.  The trunk at batch 256 is bound by the HBM traffic of its activations, not by its GEMMs: stock
// PyTorch runs bias, residual add and ReLU as separate passes (5 tensor reads/writes after a block's last
// convolution); this kernel does them in one (read y, read res, write y).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vscmi.h"
#include "vscmi_common.h"

namespace vscmi {

typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {  // round to nearest even, NaN stays NaN
    const unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// cols % 8 == 0: one 16-byte piece (8 channels) per lane and step
template <bool RES, bool RELU>
__global__ __launch_bounds__(256) void bias_act_bf16_kernel(unsigned short* __restrict__ y, const unsigned short* __restrict__ res,
                                                            const float* __restrict__ bias, long long n_piece, int cols8) {
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < n_piece; p += (long long)gridDim.x * 256) {
        const int c8 = (int)(p % cols8);
        const f32x4 b0 = reinterpret_cast<const f32x4*>(bias)[2 * c8], b1 = reinterpret_cast<const f32x4*>(bias)[2 * c8 + 1];
        u16x8 v = reinterpret_cast<const u16x8*>(y)[p];
        u16x8 r;
        if (RES) r = reinterpret_cast<const u16x8*>(res)[p];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = bf16_to_f32(v[e]) + (e < 4 ? b0[e] : b1[e - 4]);
            if (RES) f += bf16_to_f32(r[e]);
            if (RELU) f = f > 0.0f ? f : (f == f ? 0.0f : f);  // (NaN stays NaN, as torch.relu keeps it)
            v[e] = f32_to_bf16(f);
        }
        reinterpret_cast<u16x8*>(y)[p] = v;
    }
}

// The stem's tail: max-pool 3x3 / stride 2 / padding 1 over relu(x + bias), NHWC bf16, as ONE pass.  relu(. + b) and
// the rounding to bf16 are monotonic, so relu(max(x) + b) rounded once has the bits of max over the rounded
// relu(x + b) -- what the two separate passes (bias + ReLU, then nn.MaxPool2d) produce.  One 16-byte piece
// (8 channels) of one output pixel per lane.
__global__ __launch_bounds__(256) void pool3x3s2_bias_relu_bf16_kernel(const unsigned short* __restrict__ x, const float* __restrict__ bias,
                                                                       unsigned short* __restrict__ out, int N, int H, int W, int C8,
                                                                       int Ho, int Wo) {
    const long long n_piece = (long long)N * Ho * Wo * C8;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < n_piece; p += (long long)gridDim.x * 256) {
        const int c8 = (int)(p % C8);
        long long q = p / C8;
        const int wo = (int)(q % Wo);
        q /= Wo;
        const int ho = (int)(q % Ho);
        const long long n = q / Ho;
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
        bool nan[8] = {false, false, false, false, false, false, false, false};
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int h = 2 * ho + dy;
            if (h < 0 || h >= H) continue;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int w = 2 * wo + dx;
                if (w < 0 || w >= W) continue;
                const u16x8 v = reinterpret_cast<const u16x8*>(x)[((n * H + h) * W + w) * C8 + c8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = bf16_to_f32(v[e]);
                    nan[e] |= f != f;
                    m[e] = f > m[e] ? f : m[e];
                }
            }
        }
        const f32x4 b0 = reinterpret_cast<const f32x4*>(bias)[2 * c8], b1 = reinterpret_cast<const f32x4*>(bias)[2 * c8 + 1];
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = m[e] + (e < 4 ? b0[e] : b1[e - 4]);
            f = f > 0.0f ? f : (f == f ? 0.0f : f);
            if (nan[e]) f = __uint_as_float(0x7fc00000u);  // (nn.MaxPool2d propagates NaN)
            o[e] = f32_to_bf16(f);
        }
        reinterpret_cast<u16x8*>(out)[p] = o;
    }
}

}  // namespace vscmi

extern "C" int vsc_pool3x3s2_bias_relu_bf16(const void* x, const float* bias, void* out, int64_t N, int64_t H, int64_t W,
                                            int64_t C, void* hip_stream) {
    using namespace vscmi;
    if (!x || !bias || !out || N < 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || H > (1 << 20) || W > (1 << 20) ||
        (((uintptr_t)x | (uintptr_t)bias | (uintptr_t)out) & 15)) {
        set_error("vsc_pool3x3s2_bias_relu_bf16: invalid argument (C must be a multiple of 8, pointers 16-byte aligned)");
        return VSC_ERR_INVALID;
    }
    const int Ho = (int)((H - 1) / 2 + 1), Wo = (int)((W - 1) / 2 + 1);
    const long long n_piece = (long long)N * Ho * Wo * (C / 8);
    if (n_piece == 0) return VSC_OK;
    const unsigned grid = (unsigned)std::min<long long>((n_piece + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(pool3x3s2_bias_relu_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)hip_stream, (const unsigned short*)x, bias,
                       (unsigned short*)out, (int)N, (int)H, (int)W, (int)(C / 8), Ho, Wo);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

extern "C" int vsc_bias_act_bf16(void* y, const void* res, const float* bias, int64_t rows, int64_t cols, int relu,
                                 void* hip_stream) {
    using namespace vscmi;
    if (!y || !bias || rows < 0 || cols <= 0 || (cols & 7) || ((uintptr_t)y & 15) || ((uintptr_t)res & 15) || ((uintptr_t)bias & 15)) {
        set_error("vsc_bias_act_bf16: invalid argument (cols must be a multiple of 8, pointers 16-byte aligned)");
        return VSC_ERR_INVALID;
    }
    const long long n_piece = (long long)rows * (cols / 8);
    if (n_piece == 0) return VSC_OK;
    hipStream_t s = (hipStream_t)hip_stream;
    const unsigned grid = (unsigned)std::min<long long>((n_piece + 255) / 256, 256 * 32);
    unsigned short* yy = (unsigned short*)y;
    const unsigned short* rr = (const unsigned short*)res;
    const int c8 = (int)(cols / 8);
    if (res && relu) hipLaunchKernelGGL((bias_act_bf16_kernel<true, true>), dim3(grid), dim3(256), 0, s, yy, rr, bias, n_piece, c8);
    else if (res) hipLaunchKernelGGL((bias_act_bf16_kernel<true, false>), dim3(grid), dim3(256), 0, s, yy, rr, bias, n_piece, c8);
    else if (relu) hipLaunchKernelGGL((bias_act_bf16_kernel<false, true>), dim3(grid), dim3(256), 0, s, yy, rr, bias, n_piece, c8);
    else hipLaunchKernelGGL((bias_act_bf16_kernel<false, false>), dim3(grid), dim3(256), 0, s, yy, rr, bias, n_piece, c8);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}
