// Context management and status strings of the C ABI (include/dlc.h).
#include "dlc_internal.h"

#include <new>

extern "C" int dlc_abi_version(void) { return DLC_ABI_VERSION; }

extern "C" const char* dlc_status_string(int status) {
    switch (status) {
        case DLC_OK: return "ok";
        case DLC_ERR_BAD_ARG: return "bad argument";
        case DLC_ERR_BAD_SHAPE: return "bad shape";
        case DLC_ERR_UNSUPPORTED: return "unsupported dtype or mode";
        case DLC_ERR_HIP: return "HIP runtime error";
        case DLC_ERR_WORKSPACE: return "workspace too small";
        default: return "unknown status";
    }
}

extern "C" int dlc_create(int device, dlc_ctx** out) {
    if (!out) return DLC_ERR_BAD_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1) return DLC_ERR_HIP;   // no GPU: fail loudly
    if (device < 0 || device >= count) return DLC_ERR_BAD_ARG;
    dlc_ctx* c = new (std::nothrow) dlc_ctx;
    if (!c) return DLC_ERR_HIP;
    memset(c, 0, sizeof(*c));
    c->device = device;
    dlc::DeviceGuard guard(device);
    if (!guard.ok) {
        delete c;
        return DLC_ERR_HIP;
    }
    for (int i = 0; i < DLC_PROFILE_RING; ++i) {
        if (hipEventCreate(&c->ev_start[i]) != hipSuccess || hipEventCreate(&c->ev_stop[i]) != hipSuccess) {
            delete c;   // a few leaked events on this failure path are acceptable
            return DLC_ERR_HIP;
        }
    }
    c->host_lock = new (std::nothrow) std::mutex;
    if (!c->host_lock) {
        delete c;
        return DLC_ERR_HIP;
    }
    if (hipMalloc(&c->zero_page, 4096) != hipSuccess || hipMemset(c->zero_page, 0, 4096) != hipSuccess ||
        hipHostMalloc((void**)&c->host_flag, 64, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_flag, hipEventDisableTiming) != hipSuccess) {
        delete c->host_lock;
        delete c;
        return DLC_ERR_HIP;
    }
    *out = c;
    return DLC_OK;
}

extern "C" int dlc_destroy(dlc_ctx* ctx) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    {
        dlc::DeviceGuard guard(ctx->device);
        for (int i = 0; i < DLC_PROFILE_RING; ++i) {
            (void)hipEventDestroy(ctx->ev_start[i]);
            (void)hipEventDestroy(ctx->ev_stop[i]);
        }
        if (ctx->zero_page) (void)hipFree(ctx->zero_page);
        if (ctx->host_flag) (void)hipHostFree(ctx->host_flag);
        if (ctx->ev_flag) (void)hipEventDestroy(ctx->ev_flag);
        if (ctx->staging) {
            (void)hipDeviceSynchronize();            // no DMA may still read / write the pinned pieces
            dlc::staging_free(ctx->staging);
        }
    }
    delete ctx->host_lock;
    delete ctx;
    return DLC_OK;
}

extern "C" const char* dlc_last_error(const dlc_ctx* ctx) { return ctx ? ctx->err : "null context"; }

extern "C" int dlc_set_scratch(dlc_ctx* ctx, void* scratch, size_t bytes) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if ((scratch && bytes == 0) || ((uintptr_t)scratch & 255))
        return dlc::fail(ctx, DLC_ERR_BAD_ARG, "set_scratch: need a 256-byte aligned buffer (or NULL, 0)");
    ctx->scratch = scratch;
    ctx->scratch_bytes = scratch ? bytes : 0;
    return DLC_OK;
}

extern "C" int dlc_set_profiling(dlc_ctx* ctx, int enabled) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    ctx->profiling = enabled ? 1 : 0;
    ctx->prof_calls = 0;
    return DLC_OK;
}

extern "C" int dlc_profile_gemm_ms(dlc_ctx* ctx, float* out_ms, int capacity) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!out_ms || capacity < 1) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "profile_gemm_ms: bad output");
    dlc::DeviceGuard guard(ctx->device);
    long long n = ctx->prof_calls;
    if (n > DLC_PROFILE_RING) n = DLC_PROFILE_RING;
    if (n > capacity) n = capacity;
    for (long long i = 0; i < n; ++i) {
        const long long call = ctx->prof_calls - n + i;
        const int slot = (int)(call % DLC_PROFILE_RING);
        DLC_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev_stop[slot]));
        float ms = 0.f;
        DLC_HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->ev_start[slot], ctx->ev_stop[slot]));
        out_ms[i] = ms;
    }
    return (int)n;
}
