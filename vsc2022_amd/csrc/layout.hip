// HBM layout transforms (gfx950): HBM-bound, one read + one write per element.
//   pack_rows      row-major fp32 [n][dim] -> padded, k-interleaved engine layout (see vscmi_common.h)
//   row_normalize  sklearn.preprocessing.normalize (row L2), vsc/baseline/score_normalization.py:84
#include "kernels.h"

namespace vscmi {

// one thread per (row, group of 8 k): reads 8 floats, writes two float4
__global__ __launch_bounds__(256) void pack_rows_kernel(const float* __restrict__ src, int64_t n, int dim,
                                                        float* __restrict__ dst, int64_t rows_pad,
                                                        int dpad) {
    const int groups = dpad / 8;
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (x >= rows_pad * groups) return;
    const int64_t row = x / groups;
    const int g = (int)(x % groups);
    float v[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const int k = g * 8 + w;
        v[w] = (row < n && k < dim) ? src[row * dim + k] : 0.0f;
    }
    float4 even = make_float4(v[0], v[2], v[4], v[6]);
    float4 odd = make_float4(v[1], v[3], v[5], v[7]);
    float4* o = reinterpret_cast<float4*>(dst + row * dpad + g * 8);
    o[0] = even;
    o[1] = odd;
}

int launch_pack_rows(const float* src, int64_t n, int dim, float* dst, int64_t rows_pad, int dpad,
                     hipStream_t stream) {
    const int64_t total = rows_pad * (dpad / 8);
    if (total <= 0) return VSC_OK;
    hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src,
                       n, dim, dst, rows_pad, dpad);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

// fp16 image for the pre-filter GEMM (natural k order, rows zero-padded to dpadh) and, per row, an
// UPPER bound of its L2 norm (the pre-filter's error bound is built from these).  A row with an
// element fp16 cannot hold (NaN, inf, |x| > 65504) gets norm +inf: the pre-filter then passes every
// pair of that row to the exact stage.  One wave per row; HBM-bound (4 B read + 2 B written per element).
__global__ __launch_bounds__(256) void pack_half_kernel(const float* __restrict__ src, int64_t n, int dim,
                                                        _Float16* __restrict__ dst, float* __restrict__ norms,
                                                        int64_t rows_pad, int dpadh) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows_pad) return;
    typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
    half2_t* o = reinterpret_cast<half2_t*>(dst + row * dpadh);
    const float* r = src + row * dim;
    float ss = 0.0f;
    bool bad = false;
    for (int k = 2 * lane; k < dpadh; k += 128) {
        const float x0 = (row < n && k < dim) ? r[k] : 0.0f;
        const float x1 = (row < n && k + 1 < dim) ? r[k + 1] : 0.0f;
        bad |= !(fabsf(x0) <= 65504.0f) || !(fabsf(x1) <= 65504.0f);
        ss = __fmaf_rn(x0, x0, ss);
        ss = __fmaf_rn(x1, x1, ss);
        half2_t h;
        h.x = (_Float16)x0;  // round to nearest even
        h.y = (_Float16)x1;
        o[k >> 1] = h;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    const bool any_bad = __any(bad);
    // fp32 summation error is <= dim * 2^-24 relative (all terms positive): 1.0005 covers dim <= 8192;
    // larger rows use the +inf route
    if (lane == 0) norms[row] = (any_bad || dim > 8192) ? INFINITY : sqrtf(ss) * 1.0005f;
}

int launch_pack_half(const float* src, int64_t n, int dim, _Float16* dst, float* norms, int64_t rows_pad,
                     int dpadh, hipStream_t stream) {
    if (rows_pad <= 0) return VSC_OK;
    hipLaunchKernelGGL(pack_half_kernel, dim3((unsigned)((rows_pad + 3) / 4)), dim3(256), 0, stream, src, n,
                       dim, dst, norms, rows_pad, dpadh);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

// The same for the REFERENCE side of the panel-stationary pre-filter (sim_f16p.hip): fragment-major image.
// Rows are grouped in wave tiles of 64 (two 32-row MFMA blocks n = 0, 1); the 16-byte piece holding k =
// 16 ks + 8 h .. + 7 of row j sits at piece index
//     ((j / 64) * (dpadh / 16) + ks) * 128 + ((j / 32) & 1) * 64 + h * 32 + (j % 32)
// i.e. the B operand of one v_mfma_f32_32x32x16_f16 (lane l: row l & 31, k half l >> 5) is 1 KiB of
// consecutive memory.  `row0` = absolute index of the first row written (incremental adds append to a
// partly filled tile); rows [row0 + n, row0 + rows_out) are zero filled.  One wave per row, one piece per lane.
__global__ __launch_bounds__(256) void pack_half_frag_kernel(const float* __restrict__ src, int64_t n, int dim,
                                                             _Float16* __restrict__ image, float* __restrict__ norms,
                                                             int64_t row0, int64_t rows_out, int dpadh) {
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    const int lane = threadIdx.x & 63;
    const int64_t rel = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (rel >= rows_out) return;
    const int64_t row = row0 + rel;
    const int nks = dpadh / 16;
    const float* r = src + rel * dim;
    float ss = 0.0f;
    bool bad = false;
    for (int c = lane; c < dpadh / 8; c += 64) {
        f16x8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = c * 8 + e;
            const float x = (rel < n && k < dim) ? r[k] : 0.0f;
            bad |= !(fabsf(x) <= 65504.0f);
            ss = __fmaf_rn(x, x, ss);
            h[e] = (_Float16)x;  // round to nearest even
        }
        const int ks = c >> 1, hh = c & 1;
        const int64_t piece = ((row >> 6) * nks + ks) * 128 + ((row >> 5) & 1) * 64 + hh * 32 + (row & 31);
        reinterpret_cast<f16x8*>(image)[piece] = h;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    const bool any_bad = __any(bad);
    if (lane == 0) norms[rel] = (any_bad || dim > 8192) ? INFINITY : sqrtf(ss) * 1.0005f;
}

// image: base of the whole fragment-major image; norms: first norm to write (row0's)
int launch_pack_half_frag(const float* src, int64_t n, int dim, _Float16* image, float* norms, int64_t row0,
                          int64_t rows_out, int dpadh, hipStream_t stream) {
    if (rows_out <= 0) return VSC_OK;
    hipLaunchKernelGGL(pack_half_frag_kernel, dim3((unsigned)((rows_out + 3) / 4)), dim3(256), 0, stream, src, n, dim,
                       image, norms, row0, rows_out, dpadh);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

// The squared norm is the ascending-k fp32 fma chain (the oracle's order): one chain per row, sequential by
// definition.  One LANE per row: a wave owns 64 rows and walks them 64 columns at a time -- the chunk is read with
// coalesced 256-byte row segments into a padded LDS tile, then every lane runs the next 64 links of ITS row's chain
// out of the tile (column stride 65 words: conflict-free).  Second sweep (the rows are L2-hot): divide and write,
// coalesced again.  Round 3's kernel gave a whole wave to a row and let lane 0 walk it alone: 14-28 ms per 1 M rows of
// 511 floats (the score normalisation of configs[3]'s query set), against ~1.5 ms of traffic.
__global__ __launch_bounds__(256) void row_normalize_kernel(const float* __restrict__ x, int64_t n,
                                                            int dim, float* __restrict__ out) {
    __shared__ float tile[4][64][65];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * 64;
    if (row0 >= n) return;
    const int rows = (int)min((int64_t)64, n - row0);
    float (*t)[65] = tile[wave];
    float acc = 0.0f;
    for (int c0 = 0; c0 < dim; c0 += 64) {
        const int k = c0 + lane;
        for (int r = 0; r < rows; ++r) t[r][lane] = k < dim ? x[(row0 + r) * dim + k] : 0.0f;
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int kn = min(64, dim - c0);
        if (lane < rows)
            for (int kk = 0; kk < kn; ++kk) acc = __fmaf_rn(t[lane][kk], t[lane][kk], acc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    float nrm = sqrtf(acc);  // correctly rounded (__fsqrt_rn maps to the approximate native sqrt)
    if (nrm == 0.0f) nrm = 1.0f;
    for (int r = 0; r < rows; ++r) {
        const float d = __shfl(nrm, r);
        const float* src = x + (row0 + r) * dim;
        float* dst = out + (row0 + r) * dim;
        for (int k = lane; k < dim; k += 64) dst[k] = src[k] / d;
    }
}

int launch_row_normalize(const float* x, int64_t n, int dim, float* out, hipStream_t stream) {
    if (n <= 0) return VSC_OK;
    hipLaunchKernelGGL(row_normalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, n,
                       dim, out);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

}  // namespace vscmi
