// Patch front-end after key-point detection (src/sdav/input/CvInputParser.py:19-33, 49-123):
// grey conversion as cv2.imread(IMREAD_GRAYSCALE) does it, and the key-point-centred
// patch gather with the reference's clamp-inside-the-image rule and /255.0.
// HBM-bound byte gathers; one output element per thread, contiguous along the patch row.
#include "dlc_internal.h"

namespace {

__global__ __launch_bounds__(256) void rgb_to_gray_kernel(const unsigned char* __restrict__ rgb, long long n,
                                                          unsigned char* __restrict__ gray) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const unsigned r = rgb[3 * i], g = rgb[3 * i + 1], b = rgb[3 * i + 2];
        gray[i] = (unsigned char)((r * 4899u + g * 9617u + b * 1868u + 8192u) >> 14);   // OpenCV fixed-point BT.601
    }
}

// CvInputParser.py:74-86 for one axis
__device__ __forceinline__ int window_lo(int c, int dim, int half) {
    const int lo = c - half, hi = c + half;
    const int fwd = lo < 0 ? -lo : 0;
    const int aux = hi - dim + 1;
    const int back = aux > 0 ? aux : 0;
    return lo - back + fwd;
}

template <typename T>
__global__ __launch_bounds__(256) void extract_patches_kernel(const unsigned char* __restrict__ gray, int H, int W,
                                                              const int* __restrict__ kp, int P, int ps,
                                                              T* __restrict__ out) {
    const long long fp = blockIdx.x;                       // frame * P + patch
    const long long frame = fp / P;
    const int cx = kp[fp * 2], cy = kp[fp * 2 + 1];       // (x, y) = kp.pt rounded; x walks dim 0 (:111-119)
    const int x0 = window_lo(cx, H, ps / 2), y0 = window_lo(cy, W, ps / 2);
    const unsigned char* img = gray + frame * (long long)H * W;
    T* o = out + fp * (long long)ps * ps;
    for (int e = threadIdx.x; e < ps * ps; e += 256) {
        const int dx = e / ps, dy = e - dx * ps;
        o[e] = (T)img[(long long)(x0 + dx) * W + (y0 + dy)] / (T)255.0;
    }
}

}  // namespace

extern "C" int dlc_rgb_to_gray_u8(dlc_ctx* ctx, const uint8_t* rgb, int64_t n_pixels, uint8_t* gray, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!rgb || !gray || n_pixels < 1) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "rgb_to_gray: bad argument");
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    long long blocks = dlc::cdiv(n_pixels, 256);
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(rgb_to_gray_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rgb,
                       (long long)n_pixels, gray);
    DLC_LAUNCH_CHECK(ctx, "rgb_to_gray_kernel");
    return DLC_OK;
}

extern "C" int dlc_extract_patches(dlc_ctx* ctx, const uint8_t* gray, int64_t frames, int H, int W,
                                   const int32_t* key_points, int P, int patch_size, int out_dtype, void* out,
                                   void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!gray || !key_points || !out || frames < 1 || P < 1) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "extract_patches: bad argument");
    if (patch_size < 1 || patch_size % 2 == 0)
        return dlc::fail(ctx, DLC_ERR_BAD_ARG, "Invalid patch size. Patch size must be an odd number");   // CvInputParser.py:61-62
    if (H < patch_size || W < patch_size)
        return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "extract_patches: image %dx%d smaller than the %d patch", H, W, patch_size);
    if (frames * P > 0x7fffffffll) return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "extract_patches: too many patches");
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    dim3 grid((unsigned)(frames * P));
    if (out_dtype == DLC_F64)
        hipLaunchKernelGGL(extract_patches_kernel<double>, grid, dim3(256), 0, (hipStream_t)stream, gray, H, W,
                           (const int*)key_points, P, patch_size, (double*)out);
    else if (out_dtype == DLC_F32)
        hipLaunchKernelGGL(extract_patches_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, gray, H, W,
                           (const int*)key_points, P, patch_size, (float*)out);
    else
        return dlc::fail(ctx, DLC_ERR_UNSUPPORTED, "extract_patches: out dtype %d", out_dtype);
    DLC_LAUNCH_CHECK(ctx, "extract_patches_kernel");
    return DLC_OK;
}
