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

// ---- key-point detector (NOT in the reference: it uses OpenCV-contrib's non-free SURF,
// CvInputParser.py:36-46; SURVEY section 8f-1 asks for "a simple GPU detector" instead) ----------------
// Harris corners in exact integer arithmetic: Sobel 3x3 gradients Ix, Iy (int32), structure
// tensor summed over the 5x5 window (Sxx, Syy, Sxy), response R16 = 16*(Sxx*Syy - Sxy^2) -
// (Sxx+Syy)^2 (k = 1/16, int64).  Defined for pixels at least 3 away from the border, 0 elsewhere.
__device__ __forceinline__ int px(const uint8_t* g, int W, int r, int c) { return (int)g[(long long)r * W + c]; }

// Sobel gradients once per pixel, packed (ix in the low, iy in the high 16 bits: |.| <= 1020), then the 5 x 5 structure
// tensor from them: the one-pass form recomputed both gradients for each of a pixel's 25 window positions -- 600 byte
// loads and ~800 integer operations per pixel, 0.76 ms for 1063 frames.  Same integers, same response.
__global__ __launch_bounds__(256) void harris_grad_kernel(const uint8_t* __restrict__ gray, int H, int W,
                                                          int* __restrict__ grad) {
    const long long frame = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) return;
    const int r = p / W, c = p - r * W;
    int out = 0;
    if (r >= 1 && r < H - 1 && c >= 1 && c < W - 1) {
        const uint8_t* g = gray + frame * H * W;
        const int ix = (px(g, W, r - 1, c + 1) + 2 * px(g, W, r, c + 1) + px(g, W, r + 1, c + 1)) -
                       (px(g, W, r - 1, c - 1) + 2 * px(g, W, r, c - 1) + px(g, W, r + 1, c - 1));
        const int iy = (px(g, W, r + 1, c - 1) + 2 * px(g, W, r + 1, c) + px(g, W, r + 1, c + 1)) -
                       (px(g, W, r - 1, c - 1) + 2 * px(g, W, r - 1, c) + px(g, W, r - 1, c + 1));
        out = (ix & 0xffff) | (int)((unsigned)iy << 16);
    }
    grad[frame * H * W + p] = out;
}
__device__ __forceinline__ long long harris_response_at(const int* __restrict__ g, int H, int W, int r, int c) {
    if (r < 3 || r >= H - 3 || c < 3 || c >= W - 3) return 0;
    int sxx32 = 0, syy32 = 0, sxy32 = 0;                  // 25 products of at most 1020^2: well inside 32 bits
#pragma unroll
    for (int dr = -2; dr <= 2; ++dr)
#pragma unroll
        for (int dc = -2; dc <= 2; ++dc) {
            const int v = g[(long long)(r + dr) * W + c + dc];
            const int ix = (int)(short)(v & 0xffff), iy = v >> 16;
            sxx32 += ix * ix; syy32 += iy * iy; sxy32 += ix * iy;
        }
    const long long sxx = sxx32, syy = syy32, sxy = sxy32;
    return 16 * (sxx * syy - sxy * sxy) - (sxx + syy) * (sxx + syy);
}
// Response + 3 x 3 non-maximum suppression in one pass: a workgroup computes the responses of its 16 x 16 pixels and
// their one-pixel halo into LDS and suppresses from there (as two kernels the responses made a round trip through
// 392 MB of int64 per 1063 frames).  cand[p] = the response where it is a strict local maximum (ties: the lower
// row-major index wins), else 0.
__global__ __launch_bounds__(256) void harris_response_nms_kernel(const int* __restrict__ grad, int H, int W,
                                                                  long long* __restrict__ cand) {
    __shared__ long long tile[18][19];
    const long long frame = blockIdx.z;
    const int r0 = blockIdx.y * 16, c0 = blockIdx.x * 16;
    const int* g = grad + frame * H * W;
    for (int e = threadIdx.x; e < 18 * 18; e += 256) {
        const int tr = e / 18, tc = e - tr * 18;
        const int r = r0 + tr - 1, c = c0 + tc - 1;
        tile[tr][tc] = (r >= 0 && r < H && c >= 0 && c < W) ? harris_response_at(g, H, W, r, c) : 0;
    }
    __syncthreads();
    const int tr = threadIdx.x >> 4, tc = threadIdx.x & 15;
    const int r = r0 + tr, c = c0 + tc;
    if (r >= H || c >= W) return;
    const long long v = tile[tr + 1][tc + 1];
    bool keep = v > 0;
    if (keep) {
#pragma unroll
        for (int dr = -1; dr <= 1; ++dr)
#pragma unroll
            for (int dc = -1; dc <= 1; ++dc) {
                if (dr == 0 && dc == 0) continue;
                const long long u = tile[tr + 1 + dr][tc + 1 + dc];
                const bool before = dr < 0 || (dr == 0 && dc < 0);          // the neighbour's row-major index is lower
                if (u > v || (u == v && before)) keep = false;
            }
    }
    cand[frame * H * W + (long long)r * W + c] = keep ? v : 0;
}
__global__ __launch_bounds__(1024) void harris_select_kernel(long long* __restrict__ cand, int H, int W, int n,
                                                             int* __restrict__ pts, long long* __restrict__ resp_out,
                                                             int* __restrict__ count) {
    __shared__ long long sv[2][16];
    __shared__ int si[2][16];
    const long long frame = blockIdx.x;
    long long* C = cand + frame * H * W;
    const int tid = threadIdx.x, total = H * W, lane = tid & 63, w = tid >> 6;
    int found = n;
    // every thread keeps the best candidate of its own pixels (p = tid, tid + 1024, ...); only the
    // owner of a round's winner clears it and rescans
    long long bv = 0;
    int bi = 0x7fffffff;
    auto rescan = [&]() {
        bv = 0; bi = 0x7fffffff;
        for (int p = tid; p < total; p += 1024) {
            const long long v = C[p];
            if (v > bv) { bv = v; bi = p; }      // p ascending: the first maximum is the lowest index
        }
    };
    auto better = [](long long ov, int oi, long long v, int i) { return ov > v || (ov == v && oi < i); };
    rescan();
    for (int j = 0; j < n; ++j) {
        // the round's winner (largest response, ties to the lower index): shuffles inside a wave, the 16 waves'
        // winners through LDS (two buffers, one barrier per round; a 10-step tree over 1024 LDS slots with a barrier
        // per step made the 30 rounds 1.1 ms for 1063 frames)
        long long v = bv;
        int i = bi;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const long long ov = __shfl_xor(v, o);
            const int oi = __shfl_xor(i, o);
            if (better(ov, oi, v, i)) { v = ov; i = oi; }
        }
        if (lane == 0) { sv[j & 1][w] = v; si[j & 1][w] = i; }
        __syncthreads();
        v = sv[j & 1][lane & 15]; i = si[j & 1][lane & 15];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            const long long ov = __shfl_xor(v, o);
            const int oi = __shfl_xor(i, o);
            if (better(ov, oi, v, i)) { v = ov; i = oi; }
        }
        const long long wv = v;
        const int wi = i;
        if (wv <= 0) { found = j; break; }       // uniform: every thread holds the same winner
        if (tid == 0) {
            pts[(frame * n + j) * 2 + 0] = wi % W;               // cv2.KeyPoint.pt = (x = column, y = row)
            pts[(frame * n + j) * 2 + 1] = wi / W;
            resp_out[frame * n + j] = wv;
        }
        // the winner's owner retires it and needs the best of its remaining pixels (p = owner, owner + 1024, ...): its
        // whole wave fetches them, one pixel per lane (the owner alone walked 45 dependent loads per round -- 30 rounds
        // of that were 1 ms for 1063 frames)
        const int owner = wi & 1023;
        if ((owner >> 6) == w) {
            if (tid == owner) C[wi] = 0;
            long long rv = 0;
            int ri = 0x7fffffff;
            for (int p0 = owner; p0 < total; p0 += 1024 * 64) {
                const int p = p0 + 1024 * lane;
                const long long cv = (p < total && p != wi) ? C[p] : 0;
                if (cv > rv) { rv = cv; ri = p; }                // p ascending within a lane
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const long long ov = __shfl_xor(rv, o);
                const int oi = __shfl_xor(ri, o);
                if (better(ov, oi, rv, ri)) { rv = ov; ri = oi; }
            }
            if (tid == owner) { bv = rv; bi = rv > 0 ? ri : 0x7fffffff; }
        }
    }
    for (int j = found + tid; j < n; j += 1024) {
        pts[(frame * n + j) * 2 + 0] = -1;
        pts[(frame * n + j) * 2 + 1] = -1;
        resp_out[frame * n + j] = 0;
    }
    if (tid == 0) count[frame] = found;
}

}  // namespace

extern "C" size_t dlc_harris_keypoints_workspace_bytes(int64_t frames, int H, int W) {
    if (frames < 1 || H < 7 || W < 7) return 0;
    return 2 * dlc::align_up((size_t)frames * H * W * 8, 256);
}

extern "C" int dlc_harris_keypoints_u8(dlc_ctx* ctx, const uint8_t* gray, int64_t frames, int H, int W, int n,
                                       int32_t* points, int64_t* responses, int32_t* counts, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!gray || !points || !responses || !counts || frames < 1 || n < 1)
        return dlc::fail(ctx, DLC_ERR_BAD_ARG, "harris_keypoints: bad argument");
    if (H < 7 || W < 7 || (long long)H * W > 0x7fffffffll || frames > 65535)
        return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "harris_keypoints: %lld frames of %dx%d unsupported", (long long)frames, H, W);
    const size_t need = dlc_harris_keypoints_workspace_bytes(frames, H, W);
    if (!workspace || workspace_bytes < need)
        return dlc::fail(ctx, DLC_ERR_WORKSPACE, "harris_keypoints: workspace %zu < %zu bytes", workspace_bytes, need);
    if ((uintptr_t)workspace & 255) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "harris_keypoints: workspace must be 256-byte aligned");
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    hipStream_t st = (hipStream_t)stream;
    long long* resp = (long long*)workspace;
    long long* cand = (long long*)((char*)workspace + need / 2);
    dim3 grid((unsigned)dlc::cdiv((int64_t)H * W, (int64_t)256), (unsigned)frames);
    {
        // gradients into the first half of the workspace, candidates into the second
        int* grad = (int*)resp;
        hipLaunchKernelGGL(harris_grad_kernel, grid, dim3(256), 0, st, gray, H, W, grad);
        dim3 tiles((unsigned)dlc::cdiv((int64_t)W, (int64_t)16), (unsigned)dlc::cdiv((int64_t)H, (int64_t)16), (unsigned)frames);
        hipLaunchKernelGGL(harris_response_nms_kernel, tiles, dim3(256), 0, st, (const int*)grad, H, W, cand);
        DLC_LAUNCH_CHECK(ctx, "harris_response_nms_kernel");
    }
    hipLaunchKernelGGL(harris_select_kernel, dim3((unsigned)frames), dim3(1024), 0, st, cand, H, W, n, (int*)points,
                       (long long*)responses, (int*)counts);
    DLC_LAUNCH_CHECK(ctx, "harris_select_kernel");
    return DLC_OK;
}

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
