// Micro-benchmark of csrc/radix.h (round 5): the stable LSD radix sort behind the exact stage's candidate order, the final
// ordering of the kept hits and the pair-max.  Times every kernel class of a sort with HIP events and checks the result.
//   hipcc -O3 --offload-arch=gfx950 -I../../vsc2022_amd/csrc -o radix_bench radix_bench.hip && ./radix_bench [n] [bits]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

#include "radix.h"

namespace vscmi {
thread_local char g_err[512];
void set_error(const char* fmt, ...) { (void)fmt; }
}  // namespace vscmi
using namespace vscmi;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <class K, class V>
static void run(const char* name, int64_t n, int bits, bool check) {
    std::vector<K> hk((size_t)n);
    std::vector<V> hv((size_t)n);
    uint64_t s = 88172645463325252ull;
    for (int64_t i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        hk[(size_t)i] = (K)(s >> 11) & (K)((bits >= (int)sizeof(K) * 8) ? ~(K)0 : (((K)1 << bits) - 1));
        hv[(size_t)i] = (V)i;
    }
    K *ka, *kb; V *va, *vb; void* tmp;
    CK(hipMalloc(&ka, n * sizeof(K))); CK(hipMalloc(&kb, n * sizeof(K)));
    CK(hipMalloc(&va, n * sizeof(V))); CK(hipMalloc(&vb, n * sizeof(V)));
    CK(hipMalloc(&tmp, radix_tmp_bytes(n)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f; int where = 0;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemcpy(ka, hk.data(), n * sizeof(K), hipMemcpyHostToDevice));
        CK(hipMemcpy(va, hv.data(), n * sizeof(V), hipMemcpyHostToDevice));
        CK(hipEventRecord(e0));
        where = radix_sort_pairs<K, V>(ka, kb, va, vb, n, 0, bits, false, tmp, 0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    const int passes = (bits + 7) / 8;
    const double bytes = (double)n * passes * (3.0 * sizeof(K) + 2.0 * sizeof(V));
    printf("%-28s n=%lld bits=%d passes=%d: %.3f ms  (%.3f ms/pass, %.0f GB/s of the passes' algorithmic bytes)\n", name, (long long)n, bits,
           passes, best, best / passes, bytes / best / 1e6);
    if (check) {
        std::vector<K> ok((size_t)n); std::vector<V> ov((size_t)n);
        CK(hipMemcpy(ok.data(), where ? kb : ka, n * sizeof(K), hipMemcpyDeviceToHost));
        CK(hipMemcpy(ov.data(), where ? vb : va, n * sizeof(V), hipMemcpyDeviceToHost));
        std::vector<uint32_t> idx((size_t)n);
        std::iota(idx.begin(), idx.end(), 0u);
        std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return hk[a] < hk[b]; });
        int64_t bad = 0;
        for (int64_t i = 0; i < n; ++i) bad += (ok[(size_t)i] != hk[idx[(size_t)i]]) || ((uint64_t)ov[(size_t)i] != idx[(size_t)i]);
        printf("    check: %s (%lld mismatches)\n", bad ? "FAILED" : "ok", (long long)bad);
    }
    CK(hipFree(ka)); CK(hipFree(kb)); CK(hipFree(va)); CK(hipFree(vb)); CK(hipFree(tmp));
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 39000000;
    run<uint32_t, uint32_t>("u32 key, u32 value", n, 21, true);            // the candidate order of an int8 launch
    run<uint32_t, uint32_t>("u32 key, u32 value", n, 32, false);
    run<uint64_t, uint32_t>("u64 key, u32 value", n * 2, 56, false);       // the final ordering (7 passes of it)
    run<uint32_t, uint64_t>("u32 key, u64 value", n, 32, false);           // pair-max
    run<uint32_t, uint32_t>("u32 key, u32 value", 300000, 21, true);
    run<uint32_t, uint32_t>("u32 key, u32 value", 4097, 21, true);
    return 0;
}
