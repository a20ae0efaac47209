// Internal helpers shared by the HIP translation units behind include/dlc.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <mutex>

#include "../../include/dlc.h"

struct dlc_host_staging;                           // pinned staging ring + host copy threads (host_staging.hip)

struct dlc_ctx {
    int device;
    char err[512];
    int profiling;
    long long prof_calls;                       // cosine_topk calls recorded since profiling was enabled
    hipEvent_t ev_start[DLC_PROFILE_RING];
    hipEvent_t ev_stop[DLC_PROFILE_RING];
    void* scratch;                              // caller-owned split-K scratch (dlc_set_scratch), may be null
    size_t scratch_bytes;
    // hipFuncSetAttribute is per DEVICE and a context is bound to one device: which kernels already
    // have their dynamic-LDS limit raised on this context's device (bit = DLC_ATTR_* id)
    unsigned long long func_attr_set;
    void* zero_page;                            // 4 KiB of zeros in device memory (source of masked LDS-DMA pieces)
    dlc_host_staging* staging;                  // created by the first dlc_host_to_device / dlc_device_to_host
    std::mutex* host_lock;                      // serialises the staged transfers, the ring's creation and its teardown
    int host_threads;                           // host copy threads of the staging (0 = min(16, hardware threads))
    unsigned long long* host_flag;              // 64 page-locked bytes: the one word dlc_sdav_similarity_matrix reads back
    hipEvent_t ev_flag;                         // ... and the event behind its copy
};

// ids of the kernels that need hipFuncAttributeMaxDynamicSharedMemorySize (bits of dlc_ctx::func_attr_set)
enum {
    DLC_ATTR_GEMM_BASE = 0,      // + tag (0 bf16, 1 f16) * 8 + mode (0..3) * 2 + maskq   -> bits 0..15
    DLC_ATTR_GEMV_BASE = 16,     // + tag * 3 + {QB 1,2,4 -> 0,1,2}                        -> bits 16..21
    DLC_ATTR_DGEMM_BASE = 24,    // + (fp32 ? 4 : 0) + {KN, NK, conv C%8, conv any -> 0..3}  -> bits 24..31
    DLC_ATTR_DMA64_BASE = 32,    // gemm_dma_f64_kernel: + {KN, NK, conv} + (96-column form ? 3 : 0) + (128-row tile ? 6 : 0) -> bits 32..43
    DLC_ATTR_PAIR_TILE = 48,     // pair_score_tile_kernel
    DLC_ATTR_GRAM_I8 = 49,       // gram_i8_kernel
    DLC_ATTR_PAIR_FILTER = 50,   // pair_score_filter_kernel
    DLC_ATTR_SPLIT_F16 = 51,     // gemm_split_f16_kernel
    DLC_ATTR_DS_DMA = 52,        // distinctive_score_dma_kernel
};

namespace dlc {

void staging_free(dlc_host_staging* s);

inline int fail(dlc_ctx* ctx, int status, const char* fmt, ...) {
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return status;
}

#define DLC_HIP_CHECK(ctx, expr)                                                              \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess)                                                                \
            return dlc::fail((ctx), DLC_ERR_HIP, "%s failed: %s (%s:%d)", #expr,              \
                             hipGetErrorString(e__), __FILE__, __LINE__);                     \
    } while (0)

#define DLC_LAUNCH_CHECK(ctx, what)                                                           \
    do {                                                                                      \
        hipError_t e__ = hipGetLastError();                                                   \
        if (e__ != hipSuccess)                                                                \
            return dlc::fail((ctx), DLC_ERR_HIP, "launch of %s failed: %s", (what),           \
                             hipGetErrorString(e__));                                         \
    } while (0)

__host__ __device__ inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
__host__ __device__ inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// RAII-free device guard: the C ABI is re-entrant per context and each context
// is bound to one device.
struct DeviceGuard {
    int prev;
    bool ok;
    explicit DeviceGuard(int dev) : prev(-1), ok(true) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace dlc

// ---- device-side vector types ------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) double f64x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

struct dlc_bf16_tag {};
struct dlc_f16_tag {};

__device__ __forceinline__ float dlc_bf16_bits_to_f32(unsigned short h) {
    return __uint_as_float(((unsigned)h) << 16);
}
__device__ __forceinline__ float dlc_f16_bits_to_f32(unsigned short h) {
    _Float16 v = __builtin_bit_cast(_Float16, h);
    return (float)v;
}

// Order-preserving 64-bit key of a double (atomicMin / atomicMax on unsigned long long): per-frame minima / maxima of
// the CnnVtl descriptor are folded this way by several kernels (cnnvtl.hip, the convolution epilogue of gemm_dma_f64.hip).
// keys[2 f] = key of the minimum of frame f (initially ~0), keys[2 f + 1] = key of its maximum (initially 0).
__device__ __forceinline__ unsigned long long dlc_f64_key(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double dlc_f64_unkey(unsigned long long k) {
    const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}
