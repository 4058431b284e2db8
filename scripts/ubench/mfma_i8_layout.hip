// Operand / result layout of v_mfma_i32_16x16x64_i8 as sim_i8p.hip assumes it, checked against a scalar loop:
//   A: lane l holds row l & 15, k bytes 16 (l >> 4) .. + 15;  B: lane l holds column l & 15, same k bytes;
//   D: lane l, register r holds row 4 (l >> 4) + r, column l & 15.
//   hipcc -O2 --offload-arch=gfx950 -o mfma_i8_layout mfma_i8_layout.hip && ./mfma_i8_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const int8_t* A, const int8_t* B, int* D) {  // A[16][64], B[16][64] (row-major, B[j][k]), D[16][16]
    const int l = threadIdx.x;
    i32x4 a = *reinterpret_cast<const i32x4*>(A + (l & 15) * 64 + 16 * (l >> 4));
    i32x4 b = *reinterpret_cast<const i32x4*>(B + (l & 15) * 64 + 16 * (l >> 4));
    i32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}
int main() {
    int8_t hA[1024], hB[1024];
    srand(1);
    for (int i = 0; i < 1024; ++i) { hA[i] = (int8_t)(rand() % 255 - 127); hB[i] = (int8_t)(rand() % 255 - 127); }
    int8_t *dA, *dB; int* dD; int hD[256];
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 1024);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0, badT = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        int s = 0;
        for (int kk = 0; kk < 64; ++kk) s += (int)hA[i * 64 + kk] * (int)hB[j * 64 + kk];
        bad += hD[i * 16 + j] != s;
        badT += hD[j * 16 + i] != s;
    }
    printf("mismatches: assumed layout %d, transposed %d\n", bad, badT);
    return bad != 0;
}
