// Internal helpers shared by the libvscmi translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/vscmi.h"

namespace vscmi {

void set_error(const char* fmt, ...);

#define VSC_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            ::vscmi::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                               __LINE__);                                                     \
            return VSC_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)

#define VSC_TRY(expr)                \
    do {                             \
        int _rc = (expr);            \
        if (_rc != VSC_OK) return _rc; \
    } while (0)

// ---- device data layout -------------------------------------------------------------------
// Descriptor rows live in HBM as fp32 [rows padded to ROW_PAD][dpad], dpad = dim rounded up to
// K_PAD, zero filled.  Inside every group of 8 consecutive k the order is
//   [k0 k2 k4 k6 | k1 k3 k5 k7]
// so that one 16-byte LDS read hands lanes 0-31 the even k and lanes 32-63 the odd k of four
// consecutive v_mfma_f32_32x32x2_f32 steps: the MFMA chain then runs in ascending k, which is
// the arithmetic contract shared with the oracle (acc = fmaf(q[k], r[k], acc), k = 0..d-1).
constexpr int ROW_PAD = 128;
constexpr int ROW_PAD_H = 256;  // row padding of everything the 256x256 fp16 pre-filter tiles read
constexpr int ROW_PAD_REF = 512;  // row padding of the reference images (whole 512-column steps of sim_f16p.hip)
constexpr int K_PAD = 64;  // >= 2 K-tiles of 32 per row: the tile stream prefetches two K-tiles ahead

__host__ __device__ inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
__host__ __device__ inline int64_t round_up64(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// position of logical k inside its row
__host__ __device__ inline int k_slot(int k) {
    const int g = k & ~7, w = k & 7;
    return g | ((w & 1) ? 4 + (w >> 1) : (w >> 1));
}

// order-preserving float -> uint32 key (larger float => larger key)
__host__ __device__ inline uint32_t f2key(float f) {
    uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
    u = __float_as_uint(f);
#else
    union { float f; uint32_t u; } c; c.f = f; u = c.u;
#endif
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float key2f(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    union { float f; uint32_t u; } c; c.u = u; return c.f;
#endif
}

// Debug aid: with VSC_POISON_ALLOC=1 every fresh device buffer is filled with 0xFF bytes (NaN as
// fp32, -1 as an index), so a kernel that consumes memory nobody wrote shows up as a parity failure
// instead of passing on zero-filled fresh pages (tests/test_gpu_smoke.py).  Defined in api.hip.
bool poison_mode();

// Kernel attributes (the > 64 KiB dynamic-LDS opt-in) are per DEVICE, not per process: a process that creates
// handles on a second device must set them there too.  true exactly once per (call site, current device).
struct PerDeviceOnce {
    std::atomic<bool> done[64] = {};
    int dev_seen = -1;
    // true while the attributes of the current device are not known to be set: the caller sets them and then calls
    // commit() -- a failed hipFuncSetAttribute returns early (VSC_HIP) and is retried by the next launch.  Two threads
    // racing on a fresh device both set the (idempotent) attributes.
    bool first() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { dev_seen = -1; return true; }  // unknown: always set
        dev_seen = dev;
        return !done[dev].load(std::memory_order_acquire);
    }
    void commit() {
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) done[dev].store(true, std::memory_order_release);
    }
};

// growable device buffer
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int reserve(size_t need) {
        if (need <= bytes) return VSC_OK;
        if (p) {
            hipError_t e = hipFree(p);
            p = nullptr;
            bytes = 0;
            if (e != hipSuccess) { set_error("hipFree failed: %s", hipGetErrorString(e)); return VSC_ERR_HIP; }
        }
        size_t want = need + need / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
            return VSC_ERR_NOMEM;
        }
        bytes = want;
        if (poison_mode()) {
            // hipMemset on device memory is asynchronous and runs on the NULL stream, which the library's
            // non-blocking streams do not wait for: without the sync the fill can land AFTER a kernel has
            // written the buffer (seen as "missing references" after an incremental add)
            (void)hipMemset(p, 0xFF, bytes);
            (void)hipDeviceSynchronize();
        }
        return VSC_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

}  // namespace vscmi
