// Micro-benchmark: peak of v_mfma_f32_32x32x2_f32 on this chip, alone and with the LDS fragment reads
// of the similarity kernel.  hipcc --offload-arch=gfx950 -O3 mfma_f32.hip -o mfma_f32 && ./mfma_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int THREADS = 256, int BAR = 4>  // 0: MFMA only; 1: + 4 ds_read_b128 per 16 MFMA (prefetched); 2: + barrier every 4 groups; 3: + 8 LDS-DMA per wave per 4 groups
__global__ __launch_bounds__(THREADS, THREADS == 256 ? 2 : 2) void k(float* out, int iters, const float* src) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    for (int x = tid; x < 16384; x += THREADS) { unsigned h = (x * 2654435761u) ^ (blockIdx.x * 40503u); h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15; reinterpret_cast<float*>(smem)[x] = ((int)(h & 0xffff) - 32768) / 740000.0f; }
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const char* base = smem + (tid & 63) * 16 + ((tid >> 6) & 3) * 4096;
    f32x4 a0 = *(const f32x4*)(base), a1 = *(const f32x4*)(base + 1024), b0 = *(const f32x4*)(base + 2048), b1 = *(const f32x4*)(base + 3072);
    for (int it = 0; it < iters; ++it) {
        f32x4 na0 = a0, na1 = a1, nb0 = b0, nb1 = b1;
        if (MODE >= 1) {
            const char* p = base + ((it & 3) << 14);
            na0 = *(const f32x4*)(p); na1 = *(const f32x4*)(p + 1024); nb0 = *(const f32x4*)(p + 2048); nb1 = *(const f32x4*)(p + 3072);
        }
        if (MODE >= 2 && (it % BAR) == BAR - 1) __syncthreads();
        if (MODE >= 3 && (it % BAR) == BAR - 1) {
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)(blockIdx.x & 1023) * 65536), 0, 262144, 0x00020000);
            char* dst = smem + 32768 + ((tid >> 6) & 3) * 8192;
#pragma unroll
            for (int n = 0; n < 8; ++n)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + n * 1024), 16, (tid & 63) * 16 + n * 1024 + ((tid >> 6) & 3) * 8192, ((it >> 2) & 7) * 32768, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[s], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b1[s], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b0[s], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1[s], acc[3], 0, 0, 0);
        }
        a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
    float s = 0;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * THREADS + tid] = s;
}

template <int MODE, int THREADS = 256, int BAR = 4>
void run(const char* name, int blocks, int lds, int iters = 20000) {
    float* out; hipMalloc(&out, blocks * THREADS * 4);
    static float* src = nullptr; if (!src) { hipMalloc(&src, (size_t)1024 * 65536 * 4 + (1 << 20)); hipMemset(src, 0, (size_t)1024 * 65536 * 4 + (1 << 20)); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k<MODE, THREADS, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipLaunchKernelGGL((k<MODE, THREADS, BAR>), dim3(blocks), dim3(THREADS), lds, 0, out, 100, src);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, THREADS, BAR>), dim3(blocks), dim3(THREADS), lds, 0, out, iters, src);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * (THREADS / 64) * iters * 16.0 * 4096.0;
    printf("%-28s blocks=%4d  %.1f ms  %.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    run<0>("mfma only, 1 WG/CU", 256, 65536);
    run<0>("mfma only, 2 WG/CU", 512, 65536);
    run<1>("mfma + ds_read, 1 WG/CU", 256, 65536);
    run<1>("mfma + ds_read, 2 WG/CU", 512, 65536);
    run<0>("mfma only, 2 WG/CU (again)", 512, 65536);
    run<2>("+barrier/4grp 2WG/CU", 512, 65536, 400000);
    run<2>("+barrier/4grp 1WG/CU", 256, 65536, 400000);
    run<3>("+barrier+DMA 2WG/CU", 512, 65536, 400000);
    run<3>("+barrier+DMA 1WG/CU", 256, 65536, 400000);
    run<2, 512, 8>("8 waves, barrier/8grp 1WG/CU", 256, 131072, 400000);
    run<3, 512, 8>("8 waves, bar+DMA /8grp 1WG/CU", 256, 131072, 400000);
    run<2, 512, 4>("8 waves, barrier/4grp 1WG/CU", 256, 131072, 400000);
    run<2, 256, 8>("4 waves, barrier/8grp 2WG/CU", 512, 65536, 400000);
    run<3, 256, 8>("4 waves, bar+DMA/8grp 2WG/CU", 512, 65536, 400000);
    run<1>("mfma+ds_read 2WG/CU 1.7s", 512, 65536, 2000000);
    run<0>("mfma only 2WG/CU 1.7s", 512, 65536, 2000000);
    run<1>("mfma+ds_read 1WG/CU 0.9s", 256, 65536, 2000000);
    return 0;
}
