// int8 images for the int8 panel pre-filter (sim_i8p.hip), gfx950.  HBM-bound layout transforms.
//
// A row x is stored as integers q with x = s (q + e): s > 0 the scale, |q| <= 127, e the rounding / clamping
// residual.  What the pre-filter's error bound needs from a row is exact bookkeeping, not a particular rounding:
//     E >= || x - s q ||_2     (computed from the residuals that were actually produced)
//     N >= || x ||_2
// so that for two rows  | x . y - s_x s_y (q_x . q_y) | <= E_x N_y + (N_x + E_x) E_y   (Cauchy-Schwarz on
// x . y - x~ . y~ = (x - x~) . y + x~ . (y - y~), x~ = s q).  Rows holding NaN / inf get N = +inf: the
// pre-filter then hands every pair of that row to the exact stage.
//
//   quant_ref_frag    reference rows: one scale per ROW (max |x| / 127), fragment-major image (the B operand of one
//                     v_mfma_i32_16x16x64_i8 -- lane l: row l & 15, k bytes 16 (l >> 4) .. + 15 of a 64-k step -- is
//                     1 KiB of consecutive memory; a 64-row tile's step is 4 KiB: [16-row block][k piece][row]),
//                     meta[row] = {1 / s, E, N, N'}
//   quant_query_panels  query rows of ONE launch: one scale per 128-row PANEL (the kernel's epilogue compares the
//                     integer accumulators of a whole panel against one threshold per reference column), natural
//                     image [panel rows][dpad8], pstat[panel] = {1 / s, max E, max N, max N'}
//
// EXCLUDED coordinates.  A coordinate that has the same value v_c in every reference row contributes q_c v_c to every
// score of query row q: a per-row constant, not something 8 bits should be spent on -- and when it is large it
// ruins the scale of the whole row (score-normalised descriptors, vsc/baseline/score_normalization.py:99-104: the
// last coordinate of every reference is 1, the others ~0.04; one scale for both leaves +-5 levels for the
// descriptor).  Up to 8 such coordinates are left out of both images (quantised as 0, not counted in E and N');
// their contribution b_q = sum_c q_c v_c moves the row's threshold instead (row_bias_thresholds), which the kernel's
// per-row-threshold variant applies.  N stays the norm of the WHOLE row (the rounding of the exact fp32 chain scales
// with it), N' is the norm of what the image represents:
//     | x . y - b - s_x s_y (q_x . q_y) | <= E_x N'_y + (N'_x + E_x) E_y       (x, y restricted to the kept coordinates)
//
// CENTRED references (round 6).  Descriptors with a common direction (every pair at cosine 0.2 ... 0.5: uncentred
// embeddings) spend their 8 bits on that direction: the scale follows mu, the bound grows, and at cosine 0.5 the
// score-normalised search left int8 for fp16 (profiles/r06_distributions.md).  For ANY fixed vector mu
//     x . y = x . (y - mu) + x . mu ,
// so the reference image may hold y - mu (mu = the mean of the rows present at the first search over >= 1024 rows; kept
// from then on) and the second term, a per-query-row constant like b above, moves the row's threshold
// (row_center + row_bias_thresholds).  The image then represents y' = fl(y - mu): E is the residual of y', N' = ||y'||,
// and the rounding of the subtraction itself, |y - mu - y'| <= 2^-24 |y - mu| per coordinate, is added to E.  The
// queries are not touched.  Excluded coordinates stay excluded (mu is 0 there).
#include <algorithm>

#include "kernels.h"

namespace vscmi {

typedef int i32x4 __attribute__((ext_vector_type(4)));

// squares below FLT_MIN may flush to zero inside the sums: each such residual is < 1.1e-19 in magnitude
__device__ __forceinline__ float norm_up(float ss, int d) { return sqrtf(ss) * 1.001f + 2.2e-19f * sqrtf((float)d); }

__device__ __forceinline__ int quant1(float x, float inv_s, float s, float& ss_e) {
    float q = rintf(x * inv_s);
    q = fminf(127.0f, fmaxf(-127.0f, q));
    if (!(q == q)) q = 0.0f;               // NaN input (the row's N is +inf anyway)
    const float d = __fmaf_rn(-s, q, x);   // the exact residual, rounded once
    ss_e = __fmaf_rn(d, d, ss_e);
    return (int)q;
}

// One wave per row, lane c holds k = 16 c .. 16 c + 15 (dims <= 1024); source = the PACKED fp32 rows of the index
// (vscmi_common.h: dpad floats per row, every group of 8 stored [k0 k2 k4 k6 | k1 k3 k5 k7]; rows past the last one
// and coordinates past dim are zero).  Rows [row0, row0 + rows) of the index are (re)written.
__global__ __launch_bounds__(256) void quant_ref_frag_kernel(const float* __restrict__ packed, int dpad,
                                                             i32x4* __restrict__ image, float4* __restrict__ meta,
                                                             int64_t row0, int64_t rows, int dpad8, ExcludedDims ex,
                                                             const float* __restrict__ mu) {
    const int lane = threadIdx.x & 63;
    const int64_t rel = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (rel >= rows) return;
    const int64_t row = row0 + rel;
    const int npiece = dpad8 / 16, nk4 = dpad8 / 64;
    float v[16];
    float amax = 0.0f, ss_n = 0.0f, ss_k = 0.0f;
    bool bad = false;
    if (lane * 16 < dpad) {
        const float4* src = reinterpret_cast<const float4*>(packed + row * dpad + lane * 16);
        const float4 e0 = src[0], o0 = src[1], e1 = src[2], o1 = src[3];
        const float t[16] = {e0.x, o0.x, e0.y, o0.y, e0.z, o0.z, e0.w, o0.w, e1.x, o1.x, e1.y, o1.y, e1.z, o1.z, e1.w, o1.w};
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = t[e];
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = 0.0f;
    }
    float m16[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) m16[e] = 0.0f;
    if (mu && lane * 16 < dpad) {   // the centre, in the rows' packed order (zero on excluded coordinates)
        const float4* ms = reinterpret_cast<const float4*>(mu + lane * 16);
        const float4 e0 = ms[0], o0 = ms[1], e1 = ms[2], o1 = ms[3];
        const float t[16] = {e0.x, o0.x, e0.y, o0.y, e0.z, o0.z, e0.w, o0.w, e1.x, o1.x, e1.y, o1.y, e1.z, o1.z, e1.w, o1.w};
#pragma unroll
        for (int e = 0; e < 16; ++e) m16[e] = t[e];
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float x = v[e];
        bad |= !(fabsf(x) <= 3.0e38f);
        ss_n = __fmaf_rn(x, x, ss_n);        // N: the norm of the row as the exact chain sees it
        v[e] = x - m16[e];                   // what the image represents (mu = 0: the row itself, exactly)
        if (ex.holds(lane * 16 + e)) v[e] = 0.0f;  // excluded coordinate: not in the image
        amax = fmaxf(amax, fabsf(v[e]));
        ss_k = __fmaf_rn(v[e], v[e], ss_k);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        amax = fmaxf(amax, __shfl_xor(amax, off));
        ss_n += __shfl_xor(ss_n, off);
        ss_k += __shfl_xor(ss_k, off);
    }
    bad = __any(bad);
    float s = amax / 127.0f;
    if (bad || !(s > 0.0f) || !(s < 3.0e38f)) s = 1.0f;
    const float inv_s = 1.0f / s;
    float ss_e = 0.0f;
    int b[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) b[e] = bad ? 0 : quant1(v[e], inv_s, s, ss_e);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss_e += __shfl_xor(ss_e, off);
    if (lane < npiece) {
        i32x4 w;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            w[c] = (b[4 * c] & 255) | ((b[4 * c + 1] & 255) << 8) | ((b[4 * c + 2] & 255) << 16) | ((b[4 * c + 3] & 255) << 24);
        const int k4 = lane >> 2, kp = lane & 3;
        image[((row >> 6) * nk4 + k4) * 256 + ((row >> 4) & 3) * 64 + kp * 16 + (row & 15)] = w;
    }
    if (lane == 0) {
        float4 m;
        m.x = inv_s;
        m.w = bad ? INFINITY : norm_up(ss_k, dpad);
        // (centred image: + the rounding of y - mu itself, <= 2^-24 per coordinate relative to |y - mu|)
        m.y = bad ? INFINITY : norm_up(ss_e, dpad) + (mu ? 6.0e-8f * m.w : 0.0f);
        m.z = bad ? INFINITY : norm_up(ss_n, dpad);
        meta[row] = m;
    }
}

// image / meta: bases of the whole fragment-major image and of the whole meta table
int launch_quant_ref_frag(const float* packed, int dpad, void* image, float4* meta, int64_t row0, int64_t rows, int dpad8,
                          const ExcludedDims& ex, const float* mu, hipStream_t stream) {
    if (rows <= 0) return VSC_OK;
    hipLaunchKernelGGL(quant_ref_frag_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, packed, dpad,
                       reinterpret_cast<i32x4*>(image), meta, row0, rows, dpad8, ex, mu);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

// Per-coordinate minimum and maximum over packed rows, as order-preserving keys (f2key) in packed-position order:
// mn[p] / mx[p] must hold 0xffffffff / 0 on entry.  A coordinate is constant over the rows iff its two keys agree.
__global__ __launch_bounds__(256) void dim_minmax_kernel(const float* __restrict__ packed, int64_t rows, int dpad,
                                                         unsigned* __restrict__ mn, unsigned* __restrict__ mx) {
    const int64_t per = (rows + gridDim.y - 1) / gridDim.y;
    const int64_t r0 = (int64_t)blockIdx.y * per, r1 = min(rows, r0 + per);
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= dpad || r0 >= r1) return;
    unsigned lo = 0xffffffffu, hi = 0u;
    for (int64_t r = r0; r < r1; ++r) {
        const unsigned k = f2key(packed[r * dpad + p]);
        lo = min(lo, k);
        hi = max(hi, k);
    }
    atomicMin(&mn[p], lo);
    atomicMax(&mx[p], hi);
}

int launch_dim_minmax(const float* packed, int64_t rows, int dpad, unsigned* mn, unsigned* mx, hipStream_t stream) {
    VSC_HIP(hipMemsetAsync(mn, 0xff, (size_t)dpad * sizeof(unsigned), stream));
    VSC_HIP(hipMemsetAsync(mx, 0x00, (size_t)dpad * sizeof(unsigned), stream));
    if (rows <= 0) return VSC_OK;
    const unsigned chunks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(1024, rows / 256));
    hipLaunchKernelGGL(dim_minmax_kernel, dim3((unsigned)((dpad + 255) / 256), chunks), dim3(256), 0, stream, packed, rows,
                       dpad, mn, mx);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

// Per-coordinate sum and sum of squares over packed rows (double), packed-position order: the centre of the int8 image
// and the share of the rows' energy that lies along it.
__global__ __launch_bounds__(256) void col_sums_kernel(const float* __restrict__ packed, int64_t rows, int dpad,
                                                       double* __restrict__ sum, double* __restrict__ sumsq) {
    const int64_t per = (rows + gridDim.y - 1) / gridDim.y;
    const int64_t r0 = (int64_t)blockIdx.y * per, r1 = min(rows, r0 + per);
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= dpad || r0 >= r1) return;
    double a = 0.0, b = 0.0;
    for (int64_t r = r0; r < r1; ++r) {
        const double x = (double)packed[r * dpad + p];
        a += x;
        b += x * x;
    }
    atomicAdd(&sum[p], a);
    atomicAdd(&sumsq[p], b);
}

int launch_col_sums(const float* packed, int64_t rows, int dpad, double* sum, double* sumsq, hipStream_t stream) {
    VSC_HIP(hipMemsetAsync(sum, 0, (size_t)dpad * sizeof(double), stream));
    VSC_HIP(hipMemsetAsync(sumsq, 0, (size_t)dpad * sizeof(double), stream));
    if (rows <= 0) return VSC_OK;
    const unsigned chunks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(1024, rows / 256));
    hipLaunchKernelGGL(col_sums_kernel, dim3((unsigned)((dpad + 255) / 256), chunks), dim3(256), 0, stream, packed, rows, dpad, sum,
                       sumsq);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

// c[r] = x_r . mu and cmag[r] = sum_k |x_rk mu_k| over the packed coordinates (mu is 0 on excluded ones): the share of a
// query row's scores that the centred reference image does not carry.  One wave per row.
__global__ __launch_bounds__(256) void row_center_kernel(const float* __restrict__ qpacked, int dpad, int nq,
                                                         const float* __restrict__ mu, float* __restrict__ c,
                                                         float* __restrict__ cmag) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nq) return;
    const float* q = qpacked + (int64_t)r * dpad;
    float a = 0.0f, m = 0.0f;
    for (int p = lane; p < dpad; p += 64) {
        const float x = q[p], u = mu[p];
        a = __fmaf_rn(x, u, a);
        m = __fmaf_rn(fabsf(x), fabsf(u), m);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); m += __shfl_xor(m, off); }
    if (lane == 0) { c[r] = a; cmag[r] = m; }
}

int launch_row_center(const float* qpacked, int dpad, int nq, const float* mu, float* c, float* cmag, hipStream_t stream) {
    if (nq <= 0) return VSC_OK;
    hipLaunchKernelGGL(row_center_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, stream, qpacked, dpad, nq, mu, c, cmag);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

// sum over rows of E / N' (rows with a finite, non-zero N') and their count: how loose the 8-bit bound is relative to
// the rows it describes.  For two sets of rows with isotropic directions eps / sigma(score) ~ sqrt(dim) (E_q / N_q +
// E_r / N_r): api.hip keeps the int8 kernel off when the references alone already spend the budget.
__global__ __launch_bounds__(256) void meta_looseness_kernel(const float4* __restrict__ meta, int64_t n, double* __restrict__ out) {
    double s = 0.0, c = 0.0;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256) {
        const float4 m = meta[r];
        if (m.w > 0.0f && m.w < INFINITY && m.y < INFINITY) { s += (double)m.y / (double)m.w; c += 1.0; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off); c += __shfl_xor(c, off); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], s); atomicAdd(&out[1], c); }
}

int launch_meta_looseness(const float4* meta, int64_t n, double* out2, hipStream_t stream) {
    VSC_HIP(hipMemsetAsync(out2, 0, 2 * sizeof(double), stream));
    if (n <= 0) return VSC_OK;
    const unsigned grid = (unsigned)std::min<int64_t>(1024, (n + 255) / 256);
    hipLaunchKernelGGL(meta_looseness_kernel, dim3(grid), dim3(256), 0, stream, meta, n, out2);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

// One workgroup per 128-row panel of the launch's query rows; source = the PACKED fp32 image (vscmi_common.h:
// dpad floats per row, inside every group of 8 the order [k0 k2 k4 k6 | k1 k3 k5 k7], zero padded rows and columns).
// Thread t: row t >> 2, groups (t & 3), (t & 3) + 4, ...  Two passes over the panel (256 KiB at 512-d: L2-resident).
// `perm` (optional): position p of the launch holds row perm[p] (rows sorted by threshold, sortpairs.hip); positions
// past the batch are zero rows.  thr_src / thr_out (optional): the row thresholds, gathered into position order.
__global__ __launch_bounds__(512) void quant_query_panels_kernel(const float* __restrict__ qpacked, int dpad, int nq,
                                                                 int8_t* __restrict__ q8, int dpad8,
                                                                 float4* __restrict__ pstat,
                                                                 const int32_t* __restrict__ perm,
                                                                 const float* __restrict__ thr_src,
                                                                 float* __restrict__ thr_out, ExcludedDims ex) {
    __shared__ float red[8];
    __shared__ unsigned int emax_sh, nmax_sh, kmax_sh;
    __shared__ int bad_sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = tid >> 2, part = tid & 3;
    const int64_t grow = (int64_t)blockIdx.x * 128 + row;  // position inside the launch
    const int64_t srow = grow < nq ? (perm ? (int64_t)perm[grow] : grow) : -1;  // (positions past the batch: zero rows)
    const float4* src = reinterpret_cast<const float4*>(qpacked + (srow < 0 ? 0 : srow) * dpad);
    const int ngroup = srow < 0 ? 0 : dpad / 8, ngroup8 = dpad8 / 8;
    if (thr_out && part == 0) thr_out[grow] = grow < nq ? thr_src[srow] : INFINITY;
    if (tid == 0) { emax_sh = 0u; nmax_sh = 0u; kmax_sh = 0u; bad_sh = 0; }
    float amax = 0.0f;
    bool bad = false;
    for (int g = part; g < ngroup; g += 4) {
        const float4 a = src[2 * g], b = src[2 * g + 1];  // a = k0 k2 k4 k6, b = k1 k3 k5 k7
        const float x[8] = {a.x, b.x, a.y, b.y, a.z, b.z, a.w, b.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            bad |= !(fabsf(x[e]) <= 3.0e38f);
            if (!ex.holds(g * 8 + e)) amax = fmaxf(amax, fabsf(x[e]));
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
    if (lane == 0) red[wave] = amax;
    __syncthreads();
    if (__any(bad) && lane == 0) atomicOr(&bad_sh, 1);
    amax = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7])));
    __syncthreads();
    const bool panel_bad = bad_sh != 0;
    float s = amax / 127.0f;
    if (panel_bad || !(s > 0.0f) || !(s < 3.0e38f)) s = 1.0f;
    const float inv_s = 1.0f / s;
    float ss_e = 0.0f, ss_n = 0.0f, ss_k = 0.0f;
    int8_t* dst = q8 + grow * dpad8;
    for (int g = part; g < ngroup8; g += 4) {
        int lo = 0, hi = 0;
        if (g < ngroup && !panel_bad) {
            const float4 a = src[2 * g], b = src[2 * g + 1];
            float x[8] = {a.x, b.x, a.y, b.y, a.z, b.z, a.w, b.w};
            int q[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                ss_n = __fmaf_rn(x[e], x[e], ss_n);
                if (ex.holds(g * 8 + e)) x[e] = 0.0f;  // excluded coordinate: in the row's threshold, not in the image
                ss_k = __fmaf_rn(x[e], x[e], ss_k);
                q[e] = quant1(x[e], inv_s, s, ss_e);
            }
            lo = (q[0] & 255) | ((q[1] & 255) << 8) | ((q[2] & 255) << 16) | ((q[3] & 255) << 24);
            hi = (q[4] & 255) | ((q[5] & 255) << 8) | ((q[6] & 255) << 16) | ((q[7] & 255) << 24);
        }
        *reinterpret_cast<int2*>(dst + g * 8) = make_int2(lo, hi);
    }
    // the four threads of a row are neighbours in the wave
    ss_e += __shfl_xor(ss_e, 1); ss_e += __shfl_xor(ss_e, 2);
    ss_n += __shfl_xor(ss_n, 1); ss_n += __shfl_xor(ss_n, 2);
    ss_k += __shfl_xor(ss_k, 1); ss_k += __shfl_xor(ss_k, 2);
    if (part == 0 && grow < nq) {  // rows past the batch do not count (they are never reported)
        atomicMax(&emax_sh, __float_as_uint(norm_up(ss_e, dpad)));  // non-negative floats order like their bits
        atomicMax(&nmax_sh, __float_as_uint(norm_up(ss_n, dpad)));
        atomicMax(&kmax_sh, __float_as_uint(norm_up(ss_k, dpad)));
    }
    __syncthreads();
    if (tid == 0) {
        float4 m;
        m.x = inv_s;
        m.y = panel_bad ? INFINITY : __uint_as_float(emax_sh);
        m.z = panel_bad ? INFINITY : __uint_as_float(nmax_sh);
        m.w = panel_bad ? INFINITY : __uint_as_float(kmax_sh);
        pstat[blockIdx.x] = m;
    }
}

int launch_quant_query_panels(const float* qpacked, int dpad, int nq, int npanel, void* q8, int dpad8, float4* pstat,
                              const int32_t* perm, const float* thr_src, float* thr_out, const ExcludedDims& ex,
                              hipStream_t stream) {
    if (npanel <= 0) return VSC_OK;
    hipLaunchKernelGGL(quant_query_panels_kernel, dim3((unsigned)npanel), dim3(512), 0, stream, qpacked, dpad, nq,
                       reinterpret_cast<int8_t*>(q8), dpad8, pstat, perm, thr_src, thr_out, ex);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

// Largest |x| of every row of a launch over the coordinates the image keeps (one wave per row): the radius search
// sorts its rows by this value so that the rows of a 128-row panel -- which share one scale -- are rows that would
// have picked nearly the same scale on their own (E_q of the panel ~ E of its rows: ~20 % fewer candidates).
__global__ __launch_bounds__(256) void row_absmax_kernel(const float* __restrict__ qpacked, int dpad, int nq,
                                                         ExcludedDims ex, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nq) return;
    const float* q = qpacked + (int64_t)r * dpad;
    float m = 0.0f;
    for (int g = lane; g < dpad / 8; g += 64) {
        const float4 a = reinterpret_cast<const float4*>(q)[2 * g], b = reinterpret_cast<const float4*>(q)[2 * g + 1];
        const float x[8] = {a.x, b.x, a.y, b.y, a.z, b.z, a.w, b.w};
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (!ex.holds(g * 8 + e)) m = fmaxf(m, fabsf(x[e]));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (lane == 0) out[r] = m;
}

int launch_row_absmax(const float* qpacked, int dpad, int nq, const ExcludedDims& ex, float* out, hipStream_t stream) {
    if (nq <= 0) return VSC_OK;
    hipLaunchKernelGGL(row_absmax_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, stream, qpacked, dpad, nq, ex, out);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

// Row thresholds of one launch when coordinates are excluded: a pair (q, r) with exact score > t (t = the row's own
// threshold `base_thr[row]`, or the search radius *radius) has
//     (score restricted to the kept coordinates)  >  t - b_q,      b_q = sum_c q_c v_c.
// b is evaluated in fp32 (an fma chain over <= 8 terms): its own rounding, <= n 2^-23 sum |q_c v_c|, is subtracted too.
// Centred reference image: c[r] = x_r . mu (row_center_kernel) is subtracted as well; its rounding (a tree of fp32
// fmas over <= dpad terms: far below (dpad + 2) 2^-23 cmag[r]) goes into the slack.
__global__ __launch_bounds__(256) void row_bias_thresholds_kernel(const float* __restrict__ qpacked, int dpad, int nq,
                                                                  const float* __restrict__ base_thr,
                                                                  const float* __restrict__ radius, ExcludedDims ex,
                                                                  const float* __restrict__ cen,
                                                                  const float* __restrict__ cmag,
                                                                  float* __restrict__ thr) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= nq) return;
    const float* q = qpacked + (int64_t)r * dpad;
    float b = 0.0f, mag = 0.0f;
#pragma unroll
    for (int c = 0; c < I8_MAX_EXCLUDED; ++c)
        if (c < ex.n) {
            const float x = q[k_slot(ex.idx[c])];
            b = __fmaf_rn(x, ex.val[c], b);
            mag = __fmaf_rn(fabsf(x), fabsf(ex.val[c]), mag);
        }
    const float t = base_thr ? base_thr[r] : *radius;
    const float c = cen ? cen[r] : 0.0f;
    const float lowered = (t - b) - c;
    // (the subtractions' own rounding: 2^-24 relative to the larger operand; a NaN / inf bias leaves a NaN / -inf
    // threshold, and the kernel passes every pair of such a row)
    thr[r] = lowered - 1.2e-7f * ((float)ex.n * mag + (cen ? (float)(dpad + 2) * cmag[r] : 0.0f) + fabsf(t) + fabsf(b) + 2.0f * fabsf(c));
}

int launch_row_bias_thresholds(const float* qpacked, int dpad, int nq, const float* base_thr, const float* radius,
                               const ExcludedDims& ex, const float* cen, const float* cmag, float* thr, hipStream_t stream) {
    if (nq <= 0) return VSC_OK;
    hipLaunchKernelGGL(row_bias_thresholds_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream, qpacked, dpad,
                       nq, base_thr, radius, ex, cen, cmag, thr);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

}  // namespace vscmi
