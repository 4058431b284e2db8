// Convolutions of the folded SSCD trunk (3x3 / padding 1 / stride 1 or 2, and 1x1) as an implicit GEMM with the epilogue
// inside (config 3, frame inference):
//     out[m, n] = act( sum_tap sum_c x[pixel(m, tap), c] w[n, tap, c] + bias[n] (+ res[m, n]) )
// x NHWC bf16 [B, H, W, C]; w [N][taps][C] bf16 (the channels-last memory of a PyTorch [N, C, kh, kw] weight); bias fp32;
// res / out [M = B*Ho*Wo, N] bf16.  MIOpen runs these 3x3 convolutions at 0.34-0.50 PFLOP/s and PyTorch follows them with a
// bias pass and a ReLU pass; here the same product runs on v_mfma_f32_32x32x16_bf16 with bias, identity and ReLU applied to
// the accumulators.
//
// Work split: a workgroup = 4 waves x 64 output pixels = 256 pixels, all of them against the same 32*NB output channels.
//   * weights: shared by the 4 waves -> staged through LDS, 64 k at a time (one tap's 64 channels), double buffered, one
//     barrier per stage; slot (k piece, channel) so that the row operand of an MFMA is 512 contiguous bytes per half-wave;
//   * activations: every wave owns its pixels -> whole 128-byte lines (64 channels of one tap of one pixel) from global
//     memory, one stage ahead in registers, into a wave-private LDS tile, from there as the column operand; a tap outside the
//     image reads a line of zeros.  Neighbouring pixels / taps re-read the same lines: L1 / L2 serve them.
//   * per stage and wave: 8 activation loads, 8 + 4 NB LDS reads, 8 NB MFMAs.
// The accumulators leave through a wave-private LDS tile (as in gemm_epi.hip) so that identity loads and stores are 16-byte
// pieces of whole rows.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "../../include/vscmi.h"
#include "vscmi_common.h"

namespace vscmi {
namespace cg {

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

__device__ __attribute__((aligned(128))) const unsigned short zero_line[64] = {0};  // what a tap outside the image reads

struct Args {
    const __bf16* x; const __bf16* w; const float* bias; const unsigned short* res; unsigned short* out;
    long long M;          // output pixels
    int H, W, C, N;       // input height / width / channels, output channels
    int Ho, Wo, stride;   // (3x3 only)
};

constexpr int LDS_BYTES = 65536;  // 2 weight stages (2 x 16 KiB) + 4 wave-private activation tiles (4 x 8 KiB); the epilogue tiles reuse it

template <int TAPS, int NB, bool RES, bool RELU>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_gemm_bf16_kernel(Args a) {
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    constexpr int NCH = 32 * NB;                 // channels of the workgroup
    constexpr int STAGE = 128 * 64 * 2;          // bytes reserved per weight stage (NB = 4 fills it)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const long long m0 = ((long long)blockIdx.x * 4 + wave) * 64;
    const int n0 = blockIdx.y * NCH;
    const int K = TAPS * a.C;
    const int nstage = K / 64;

    // Activations of a stage = one 128-byte line (64 channels of one tap) per pixel.  They are fetched as whole lines --
    // lane -> (pixel 8 t + lane / 8, 16-byte piece lane % 8), 8 lines per load instruction (fetched in the MFMA operand
    // layout, 16 bytes from each of 64 pixels, a load instruction touches 64 lines and the L1's one-tag-per-clock rate
    // bounds the kernel at a quarter of this version's speed) -- and turned into operands through a wave-private LDS
    // tile: slot (pixel, piece ^ ((pixel >> 1) & 7)), conflict-free both ways.
    const int lp = lane >> 3, lq = lane & 7;
    int px[8];   // the centre pixel's line in 16-byte pieces from a.x (+ this lane's piece); the host checks it fits 31 bits
    int pyx[8];  // (h << 16) | w of the centre pixel, 3x3 only
    const int cp8 = a.C / 8;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const long long m = std::min<long long>(m0 + 8 * t + lp, a.M - 1);
        if (TAPS == 1) {
            px[t] = (int)(m * cp8) + lq;
            pyx[t] = 0;
        } else {
            const int wo = (int)(m % a.Wo);
            const long long r = m / a.Wo;
            const int ho = (int)(r % a.Ho);
            const long long b = r / a.Ho;
            const int h = ho * a.stride, w = wo * a.stride;
            pyx[t] = (h << 16) | w;
            px[t] = (int)(((b * a.H + h) * a.W + w) * cp8) + lq;
        }
    }
    const bf16x8* xp = reinterpret_cast<const bf16x8*>(a.x);
    const bf16x8* zp = reinterpret_cast<const bf16x8*>(zero_line) + lq;
    auto load_act = [&](int st, bf16x8 (&r)[8]) {
        const int k0 = st * 64;
        const int tap = TAPS == 1 ? 0 : k0 / a.C, c0 = TAPS == 1 ? k0 : k0 - tap * a.C;
        const int dy = TAPS == 1 ? 0 : tap / 3 - 1, dx = TAPS == 1 ? 0 : tap % 3 - 1;
        const int off = (dy * a.W + dx) * cp8 + c0 / 8;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            // a tap outside the image reads a line of zeros instead: no branch, no masked loads
            const bool ok = TAPS == 1 || ((unsigned)((pyx[t] >> 16) + dy) < (unsigned)a.H && (unsigned)((pyx[t] & 0xffff) + dx) < (unsigned)a.W);
            r[t] = *(ok ? xp + (px[t] + off) : zp);
        }
    };
    bf16x8* atile = reinterpret_cast<bf16x8*>(smem + 2 * STAGE + wave * 8192);
    auto store_act = [&](const bf16x8 (&r)[8]) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int pixel = 8 * t + lp;
            atile[pixel * 8 + (lq ^ ((pixel >> 1) & 7))] = r[t];
        }
    };
    // weight stage: thread -> channel tid % NCH, part tid / NCH of the stage's 64 k
    constexpr int PARTS = 256 / NCH;       // 2 or 4
    constexpr int PIECES = 8 / PARTS;      // 16-byte pieces per thread and stage: 4 or 2
    const int wch = tid % NCH, wpart = tid / NCH;
    const bf16x8* wsrc = reinterpret_cast<const bf16x8*>(a.w + (long long)(n0 + wch) * K) + wpart * PIECES;
    auto load_w = [&](int st, bf16x8 (&r)[PIECES]) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) r[i] = wsrc[st * 8 + i];
    };
    auto store_w = [&](int buf, const bf16x8 (&r)[PIECES]) {
        bf16x8* dst = reinterpret_cast<bf16x8*>(smem + buf * STAGE);
#pragma unroll
        for (int i = 0; i < PIECES; ++i) dst[(wpart * PIECES + i) * NCH + wch] = r[i];
    };

    f32x16 acc[NB][2];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][rb][r] = 0.0f;

    bf16x8 ra[8];
    bf16x8 wr[PIECES];
    load_w(0, wr);
    load_act(0, ra);
    store_w(0, wr);
    for (int st = 0; st < nstage; ++st) {
        store_act(ra);    // this stage's lines -> the wave's tile (its previous contents were read during stage st - 1)
        __syncthreads();  // weight stage st is in LDS; everybody is done reading stage st - 1
        const bool more = st + 1 < nstage;
        if (more) {
            load_w(st + 1, wr);
            load_act(st + 1, ra);
        }
        const bf16x8* wl = reinterpret_cast<const bf16x8*>(smem + (st & 1) * STAGE);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 fw[NB], fa[2];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const int pixel = 32 * rb + l31;
                fa[rb] = atile[pixel * 8 + ((2 * s + hi) ^ ((pixel >> 1) & 7))];
            }
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) fw[cb] = wl[(2 * s + hi) * NCH + cb * 32 + l31];
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) {
                acc[cb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[cb], fa[0], acc[cb][0], 0, 0, 0);
                acc[cb][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[cb], fa[1], acc[cb][1], 0, 0, 0);
            }
        }
        if (more) store_w((st + 1) & 1, wr);
    }
    __syncthreads();  // the stages are dead: the LDS becomes the waves' epilogue tiles
    if (m0 >= a.M) return;
    float* tile = reinterpret_cast<float*>(smem) + wave * (32 * 68);
    const int c8 = (lane & 7) * 8;
#pragma unroll
    for (int cp = 0; cp < NB / 2; ++cp) {
        const int nb = n0 + 64 * cp;
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias + nb + c8), b1 = *reinterpret_cast<const f32x4*>(a.bias + nb + c8 + 4);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the previous tile has been read
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[2 * cp + i][rb][4 * g + e];
                    *reinterpret_cast<f32x4*>(tile + l31 * 68 + 32 * i + 8 * g + 4 * hi) = v;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int r = (lane >> 3) + 8 * t;
                const long long m = m0 + 32 * rb + r;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(tile + r * 68 + c8), v1 = *reinterpret_cast<const f32x4*>(tile + r * 68 + c8 + 4);
                if (m >= a.M) continue;
                u16x8 rv;
                if (RES) rv = *reinterpret_cast<const u16x8*>(a.res + m * a.N + nb + c8);
                u16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float f = (e < 4 ? v0[e] : v1[e - 4]) + (e < 4 ? b0[e] : b1[e - 4]);
                    if (RES) f += bf16_bits_to_f32(rv[e]);
                    if (RELU) f = f > 0.0f ? f : (f == f ? 0.0f : f);
                    o[e] = f32_to_bf16_bits(f);
                }
                *reinterpret_cast<u16x8*>(a.out + m * a.N + nb + c8) = o;
            }
        }
    }
}

template <int TAPS, int NB>
static void launch(const Args& a, bool res, bool relu, hipStream_t s) {
    const dim3 grid((unsigned)((a.M + 255) / 256), (unsigned)(a.N / (32 * NB)));
    if (res && relu) hipLaunchKernelGGL((conv_gemm_bf16_kernel<TAPS, NB, true, true>), grid, dim3(256), 0, s, a);
    else if (res) hipLaunchKernelGGL((conv_gemm_bf16_kernel<TAPS, NB, true, false>), grid, dim3(256), 0, s, a);
    else if (relu) hipLaunchKernelGGL((conv_gemm_bf16_kernel<TAPS, NB, false, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_gemm_bf16_kernel<TAPS, NB, false, false>), grid, dim3(256), 0, s, a);
}

}  // namespace cg
}  // namespace vscmi

extern "C" int vsc_conv_bias_act_bf16(const void* x, const void* w, const float* bias, const void* res, void* out, int64_t B,
                                      int64_t H, int64_t W, int64_t C, int64_t N, int taps, int stride, int relu,
                                      void* hip_stream) {
    using namespace vscmi;
    if (!x || !w || !bias || !out || B < 0 || H <= 0 || W <= 0 || C <= 0 || N <= 0 || (C & 63) || (N & 63) || (taps != 1 && taps != 9) ||
        stride < 1 || stride > 2 || (taps == 1 && stride != 1) || B * H * W * (C / 8) >= (1ll << 31) || H > 32767 || W > 65535 || C > (1 << 16) || N > (1 << 16) ||
        (((uintptr_t)x | (uintptr_t)w | (uintptr_t)bias | (uintptr_t)res | (uintptr_t)out) & 15)) {
        set_error("vsc_conv_bias_act_bf16: invalid argument (C and N multiples of 64, taps 1 or 9, stride 1 or 2 (1 for taps = 1), "
                  "pointers 16-byte aligned, H <= 32767 and W <= 65535: a pixel is packed as (h << 16) | w in a signed int)");
        return VSC_ERR_INVALID;
    }
    cg::Args a;
    a.x = (const __bf16*)x; a.w = (const __bf16*)w; a.bias = bias; a.res = (const unsigned short*)res; a.out = (unsigned short*)out;
    a.H = (int)H; a.W = (int)W; a.C = (int)C; a.N = (int)N; a.stride = stride;
    a.Ho = (int)((H - 1) / stride + 1); a.Wo = (int)((W - 1) / stride + 1);
    a.M = (long long)B * a.Ho * a.Wo;
    if (a.M == 0) return VSC_OK;
    hipStream_t s = (hipStream_t)hip_stream;
    const bool wide = (N % 128) == 0;
    if (taps == 9) { if (wide) cg::launch<9, 4>(a, res != nullptr, relu != 0, s); else cg::launch<9, 2>(a, res != nullptr, relu != 0, s); }
    else { if (wide) cg::launch<1, 4>(a, res != nullptr, relu != 0, s); else cg::launch<1, 2>(a, res != nullptr, relu != 0, s); }
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}
