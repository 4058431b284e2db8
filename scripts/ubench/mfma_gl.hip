// Micro-benchmark: fp32 MFMA 64x64 wave tile fed straight from global memory (no LDS, no barrier):
// fragment-major layout => every fragment is one fully coalesced 1 KiB load per wave.
// Emulates the similarity kernel's traffic: each wave streams A (2 fragments) and B (2 fragments)
// per 16 MFMAs; waves of a workgroup share panels like a 128x128 tile (2x2 waves).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(256, 2) void k(float* out, int ngroups, const float* src, int panel_floats) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // panels: A panel of this block (128 rows = 4 row-blocks of 32), B panel likewise
    const float* A = src + (size_t)(blockIdx.x % 61) * panel_floats;        // "query" panel
    const float* B = src + (size_t)(64 + (blockIdx.x * 7) % 997) * panel_floats;  // "ref" panel
    // fragment (rowblock rb, group g): 256 floats at ((g * 4 + rb) * 256); lane reads 4 floats at lane*4
    const float* a0p = A + (wr * 2 + 0) * 256 + lane * 4;
    const float* a1p = A + (wr * 2 + 1) * 256 + lane * 4;
    const float* b0p = B + (wc * 2 + 0) * 256 + lane * 4;
    const float* b1p = B + (wc * 2 + 1) * 256 + lane * 4;
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    f32x4 fa0[DEPTH], fa1[DEPTH], fb0[DEPTH], fb1[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        fa0[d] = *(const f32x4*)(a0p + d * 1024); fa1[d] = *(const f32x4*)(a1p + d * 1024);
        fb0[d] = *(const f32x4*)(b0p + d * 1024); fb1[d] = *(const f32x4*)(b1p + d * 1024);
    }
    const int gmask = panel_floats / 1024 - 1;  // groups per panel (power of two)
    for (int g = 0; g < ngroups; g += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const f32x4 a0 = fa0[d], a1 = fa1[d], b0 = fb0[d], b1 = fb1[d];
            const int gn = (g + d + DEPTH) & gmask;
            fa0[d] = *(const f32x4*)(a0p + gn * 1024); fa1[d] = *(const f32x4*)(a1p + gn * 1024);
            fb0[d] = *(const f32x4*)(b0p + gn * 1024); fb1[d] = *(const f32x4*)(b1p + gn * 1024);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[s], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b1[s], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b0[s], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1[s], acc[3], 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int DEPTH>
void run(const char* name, int blocks, int ngroups, const float* src, int panel_floats) {
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<DEPTH>, dim3(blocks), dim3(256), 0, 0, out, 64, src, panel_floats);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<DEPTH>, dim3(blocks), dim3(256), 0, 0, out, ngroups, src, panel_floats);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * ngroups * 16.0 * 4096.0;
    printf("%-34s blocks=%6d  %.1f ms  %.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    const int panel_floats = 128 * 512;  // 128 rows x 512 k = 64 groups x 1024 floats
    float* src; size_t n = (size_t)1100 * panel_floats;
    hipMalloc(&src, n * 4);
    hipMemset(src, 0, n * 4);
    // fill with small pseudo-random values from the host
    float* h = (float*)malloc(n * 4);
    unsigned x = 12345; for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((int)(x >> 16) - 32768) / 740000.0f; }
    hipMemcpy(src, h, n * 4, hipMemcpyHostToDevice);
    run<2>("global frags depth 2, 2WG/CU", 512 * 40, 64 * 4, src, panel_floats);
    run<4>("global frags depth 4, 2WG/CU", 512 * 40, 64 * 4, src, panel_floats);
    run<8>("global frags depth 8, 2WG/CU", 512 * 40, 64 * 4, src, panel_floats);
    run<4>("global frags depth 4, long", 512 * 400, 64 * 4, src, panel_floats);
    return 0;
}
