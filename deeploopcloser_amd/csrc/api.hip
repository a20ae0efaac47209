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
    if (!guard.ok || hipEventCreate(&c->ev_gemm_start) != hipSuccess || hipEventCreate(&c->ev_gemm_stop) != hipSuccess) {
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
        (void)hipEventDestroy(ctx->ev_gemm_start);
        (void)hipEventDestroy(ctx->ev_gemm_stop);
    }
    delete ctx;
    return DLC_OK;
}

extern "C" const char* dlc_last_error(const dlc_ctx* ctx) { return ctx ? ctx->err : "null context"; }

extern "C" int dlc_set_profiling(dlc_ctx* ctx, int enabled) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    ctx->profiling = enabled ? 1 : 0;
    ctx->have_gemm_events = 0;
    return DLC_OK;
}

extern "C" float dlc_last_gemm_ms(dlc_ctx* ctx) {
    if (!ctx || !ctx->profiling || !ctx->have_gemm_events) return -1.0f;
    dlc::DeviceGuard guard(ctx->device);
    if (hipEventSynchronize(ctx->ev_gemm_stop) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, ctx->ev_gemm_start, ctx->ev_gemm_stop) != hipSuccess) return -1.0f;
    return ms;
}
