// CnnVtl encoder pieces around the conv-as-GEMM (src/cnn_vtl/network/cnn_vtl.py:28-133):
// im2col (NHWC fp64, HWIO column order), 3x3/2 max-pool, and the fused
// per-row min/max -> scale to 0..255 -> int8 cast -> column gather.
// All three are HBM-bound streaming kernels: one element per thread along the
// contiguous (channel / column) axis so that loads and stores coalesce.
#include "gemm_internal.h"

namespace {

__global__ __launch_bounds__(256) void im2col_kernel(const double* __restrict__ x, long long n, int h, int w, int c,
                                                     int kh, int kw, int stride, int pad_top, int pad_left, int oh,
                                                     int ow, double* __restrict__ cols) {
    // one block per output pixel row chunk; thread loops over the K = kh*kw*c columns (c fastest)
    const long long m = (long long)blockIdx.x;                 // output pixel index in [0, n*oh*ow)
    const int K = kh * kw * c;
    const long long img = m / ((long long)oh * ow);
    const int rem = (int)(m - img * oh * ow);
    const int oy = rem / ow, ox = rem - oy * ow;
    const int iy0 = oy * stride - pad_top, ix0 = ox * stride - pad_left;
    const double* xi = x + img * (long long)h * w * c;
    double* out = cols + m * K;
    for (int k = threadIdx.x; k < K; k += 256) {
        const int ch = k % c;
        const int t = k / c;
        const int kx = t % kw, ky = t / kw;
        const int iy = iy0 + ky, ix = ix0 + kx;
        double v = 0.0;
        if (iy >= 0 && iy < h && ix >= 0 && ix < w) v = xi[((long long)iy * w + ix) * c + ch];
        out[k] = v;
    }
}

__global__ __launch_bounds__(256) void maxpool_kernel(const double* __restrict__ x, long long n, int h, int w, int c,
                                                      int oh, int ow, double* __restrict__ y) {
    const long long total = n * oh * ow * c;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int ch = (int)(e % c);
        long long t = e / c;
        const int ox = (int)(t % ow);
        t /= ow;
        const int oy = (int)(t % oh);
        const long long img = t / oh;
        const double* xi = x + ((img * h + oy * 2) * w + ox * 2) * (long long)c + ch;
        double m = xi[0];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) m = fmax(m, xi[((long long)dy * w + dx) * c]);
        y[e] = m;
    }
}

// y[n, h/s, w/s, (dy*s + dx)*c + ch] = x[n, (h/s index)*s + dy, (w/s index)*s + dx, ch]: one element per thread along
// the output's contiguous axis; for a fixed dy the s*c values (dx, ch) are contiguous in the input as well.
__global__ __launch_bounds__(256) void space_to_depth_kernel(const double* __restrict__ x, long long n, int h, int w, int c,
                                                             int s, double* __restrict__ y) {
    const int oh = h / s, ow = w / s, oc = s * s * c, sc = s * c;
    const long long total = n * oh * ow * oc;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int k = (int)(e % oc);
        long long t = e / oc;
        const int ox = (int)(t % ow);
        t /= ow;
        const int oy = (int)(t % oh);
        const long long img = t / oh;
        const int dy = k / sc, rem = k - dy * sc;            // rem = dx * c + ch
        y[e] = x[((img * h + (long long)oy * s + dy) * w + (long long)ox * s) * c + rem];
    }
}

constexpr int MAX_SEGS = 8;
struct Segs {
    const double* ptr[MAX_SEGS];
    long long size[MAX_SEGS];     // per-row width of the segment
    long long start[MAX_SEGS + 1];
    int n;
};

__global__ void minmax_init_kernel(unsigned long long* __restrict__ keys, long long n) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r < n) { keys[r * 2] = ~0ull; keys[r * 2 + 1] = 0ull; }
}

// per-row min / max over all segments: blockIdx.y = row, blockIdx.x = slice of the row, so that a
// handful of rows still fills the chip; partial results meet in ordered-key atomics.
__global__ __launch_bounds__(256) void row_minmax_kernel(Segs s, unsigned long long* __restrict__ keys) {
    __shared__ double smin[4], smax[4];
    const long long r = blockIdx.y;
    double mn = INFINITY, mx = -INFINITY;
    for (int g = 0; g < s.n; ++g) {
        const double* p = s.ptr[g] + r * s.size[g];
        for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < s.size[g]; e += (long long)gridDim.x * 256) {
            const double v = p[e];
            mn = fmin(mn, v);
            mx = fmax(mx, v);
        }
    }
    for (int o = 32; o > 0; o >>= 1) { mn = fmin(mn, __shfl_xor(mn, o)); mx = fmax(mx, __shfl_xor(mx, o)); }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { smin[w] = mn; smax[w] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mn = fmin(fmin(smin[0], smin[1]), fmin(smin[2], smin[3]));
        mx = fmax(fmax(smax[0], smax[1]), fmax(smax[2], smax[3]));
        if (mn <= mx) {                                         // this slice saw at least one element
            atomicMin(&keys[r * 2], dlc_f64_key(mn));
            atomicMax(&keys[r * 2 + 1], dlc_f64_key(mx));
        }
    }
}

__global__ void minmax_decode_kernel(const unsigned long long* __restrict__ keys, long long n, double* __restrict__ minmax) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e < 2 * n) minmax[e] = dlc_f64_unkey(keys[e]);
}

// out[r, j] = int8( trunc( (d[r, cols[j]] - min_r) * (255 / (max_r - min_r)) ) ), wrap mod 256
__global__ __launch_bounds__(256) void quant_gather_kernel(Segs s, const long long* __restrict__ cols, long long n_cols,
                                                           const double* __restrict__ minmax,
                                                           int8_t* __restrict__ out) {
    const long long r = blockIdx.y;
    const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
    if (j >= n_cols) return;
    const long long col = cols[j];
    if (col < 0 || col >= s.start[s.n]) { out[r * n_cols + j] = 0; return; }
    int g = 0;
    while (g + 1 < s.n && col >= s.start[g + 1]) ++g;
    const double v = s.ptr[g][r * s.size[g] + (col - s.start[g])];
    const double mn = minmax[r * 2], mx = minmax[r * 2 + 1];
    const double scaled = (v - mn) * (255.0 / (mx - mn));       // cnn_vtl.py:115, same operation order
    long long t = 0;
    if (scaled == scaled && fabs(scaled) < 9.0e18) t = (long long)scaled;   // truncate toward zero
    out[r * n_cols + j] = (int8_t)(unsigned char)(t & 0xff);
}

}  // namespace

extern "C" int dlc_im2col_nhwc_f64(dlc_ctx* ctx, const double* x, int64_t n, int h, int w, int c, int kh, int kw,
                                   int stride, int pad_top, int pad_left, int oh, int ow, double* cols, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!x || !cols || n < 1 || h < 1 || w < 1 || c < 1 || kh < 1 || kw < 1 || stride < 1 || oh < 1 || ow < 1 ||
        pad_top < 0 || pad_left < 0)
        return dlc::fail(ctx, DLC_ERR_BAD_ARG, "im2col: bad argument");
    const int64_t m = n * oh * ow;
    if (m > 0x7fffffffll) return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "im2col: too many output pixels for one launch");
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)m), dim3(256), 0, (hipStream_t)stream, x, (long long)n, h, w, c, kh,
                       kw, stride, pad_top, pad_left, oh, ow, cols);
    DLC_LAUNCH_CHECK(ctx, "im2col_kernel");
    return DLC_OK;
}

extern "C" int dlc_maxpool3x3s2_nhwc_f64(dlc_ctx* ctx, const double* x, int64_t n, int h, int w, int c, double* y,
                                         void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!x || !y || n < 1 || h < 3 || w < 3 || c < 1) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "maxpool: bad argument");
    const int oh = (h - 3) / 2 + 1, ow = (w - 3) / 2 + 1;
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    const long long total = (long long)n * oh * ow * c;
    long long blocks = dlc::cdiv(total, 256);
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(maxpool_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long long)n, h, w, c,
                       oh, ow, y);
    DLC_LAUNCH_CHECK(ctx, "maxpool_kernel");
    return DLC_OK;
}

extern "C" int dlc_space_to_depth_nhwc_f64(dlc_ctx* ctx, const double* x, int64_t n, int h, int w, int c, int s, double* y,
                                          void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!x || !y || n < 1 || h < 1 || w < 1 || c < 1 || s < 1) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "space_to_depth: bad argument");
    if (h % s || w % s) return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "space_to_depth: %d x %d is not a multiple of the block %d", h, w, s);
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    long long blocks = dlc::cdiv((long long)n * h * w * c, 256);
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(space_to_depth_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long long)n, h, w,
                       c, s, y);
    DLC_LAUNCH_CHECK(ctx, "space_to_depth_kernel");
    return DLC_OK;
}

namespace {
int build_segs(dlc_ctx* ctx, const double* const* segs, const int64_t* seg_sizes, int n_segs, Segs* s) {
    s->n = n_segs;
    s->start[0] = 0;
    for (int g = 0; g < MAX_SEGS; ++g) {
        s->ptr[g] = g < n_segs ? segs[g] : nullptr;
        s->size[g] = g < n_segs ? seg_sizes[g] : 0;
        if (g < n_segs && (!segs[g] || seg_sizes[g] < 1))
            return dlc::fail(ctx, DLC_ERR_BAD_ARG, "minmax_quant_gather: segment %d is null/empty", g);
        s->start[g + 1] = s->start[g] + s->size[g];
    }
    return DLC_OK;
}

// rows of `s` folded into keys (no initialisation): slices of a row over many workgroups
int fold_rows(dlc_ctx* ctx, const Segs& s, int64_t n, unsigned long long* keys, hipStream_t st) {
    const long long width = s.start[s.n];
    long long slices = dlc::cdiv(width, (long long)256 * 16);   // >= 16 elements per thread
    const long long want = dlc::cdiv((long long)256 * 8, (long long)n);      // ~8 workgroups per CU over all rows
    if (slices > want) slices = want;
    if (slices < 1) slices = 1;
    for (int64_t r0 = 0; r0 < n; r0 += 65535) {                 // grid.y
        const int64_t nr = n - r0 < 65535 ? n - r0 : 65535;
        Segs t = s;
        for (int g = 0; g < s.n; ++g) t.ptr[g] = s.ptr[g] + r0 * s.size[g];
        hipLaunchKernelGGL(row_minmax_kernel, dim3((unsigned)slices, (unsigned)nr), dim3(256), 0, st, t, keys + 2 * r0);
    }
    DLC_LAUNCH_CHECK(ctx, "row_minmax_kernel");
    return DLC_OK;
}
}  // namespace

namespace dlc_cnn {
int fold_minmax_f64(dlc_ctx* ctx, const double* x, int64_t n_img, int64_t per_img, unsigned long long* keys, hipStream_t st) {
    Segs s{};
    s.n = 1; s.ptr[0] = x; s.size[0] = per_img; s.start[0] = 0; s.start[1] = per_img;
    return fold_rows(ctx, s, n_img, keys, st);
}
}  // namespace dlc_cnn

extern "C" int dlc_cnnvtl_frame_minmax_init(dlc_ctx* ctx, uint64_t* keys, int64_t n, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!keys || n < 1) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "frame_minmax_init: bad argument");
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    hipLaunchKernelGGL(minmax_init_kernel, dim3((unsigned)dlc::cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       (unsigned long long*)keys, (long long)n);
    DLC_LAUNCH_CHECK(ctx, "minmax_init_kernel");
    return DLC_OK;
}

extern "C" int dlc_quant_gather_i8(dlc_ctx* ctx, const double* const* segs, const int64_t* seg_sizes, int n_segs, int64_t n,
                                   const int64_t* cols, int64_t n_cols, const uint64_t* keys, double* minmax, int8_t* out,
                                   void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!segs || !seg_sizes || n_segs < 1 || n_segs > MAX_SEGS || n < 1 || !cols || n_cols < 1 || !keys || !minmax || !out)
        return dlc::fail(ctx, DLC_ERR_BAD_ARG, "quant_gather: bad argument (1..%d segments)", MAX_SEGS);
    if (n > 65535) return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "quant_gather: more than 65535 rows per call");
    Segs s;
    const int rc = build_segs(ctx, segs, seg_sizes, n_segs, &s);
    if (rc != DLC_OK) return rc;
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(minmax_decode_kernel, dim3((unsigned)dlc::cdiv(2 * n, 256)), dim3(256), 0, st,
                       (const unsigned long long*)keys, (long long)n, minmax);
    hipLaunchKernelGGL(quant_gather_kernel, dim3((unsigned)dlc::cdiv(n_cols, 256), (unsigned)n), dim3(256), 0, st, s,
                       (const long long*)cols, (long long)n_cols, (const double*)minmax, out);
    DLC_LAUNCH_CHECK(ctx, "quant_gather_kernel");
    return DLC_OK;
}

extern "C" int dlc_minmax_quant_gather_i8(dlc_ctx* ctx, const double* const* segs, const int64_t* seg_sizes, int n_segs,
                                          int64_t n, const int64_t* cols, int64_t n_cols, double* minmax, int8_t* out,
                                          void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!segs || !seg_sizes || n_segs < 1 || n_segs > MAX_SEGS || n < 1 || !cols || n_cols < 1 || !minmax || !out)
        return dlc::fail(ctx, DLC_ERR_BAD_ARG, "minmax_quant_gather: bad argument (1..%d segments)", MAX_SEGS);
    if (n > 65535) return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "minmax_quant_gather: more than 65535 rows per call");
    Segs s;
    int rc = build_segs(ctx, segs, seg_sizes, n_segs, &s);
    if (rc != DLC_OK) return rc;
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    hipStream_t st = (hipStream_t)stream;
    // minmax doubles as the key scratch: [n,2] u64 keys first, decoded in place afterwards
    unsigned long long* keys = (unsigned long long*)minmax;
    hipLaunchKernelGGL(minmax_init_kernel, dim3((unsigned)dlc::cdiv(n, 256)), dim3(256), 0, st, keys, (long long)n);
    rc = fold_rows(ctx, s, n, keys, st);
    if (rc != DLC_OK) return rc;
    hipLaunchKernelGGL(minmax_decode_kernel, dim3((unsigned)dlc::cdiv(2 * n, 256)), dim3(256), 0, st, keys, (long long)n,
                       minmax);
    hipLaunchKernelGGL(quant_gather_kernel, dim3((unsigned)dlc::cdiv(n_cols, 256), (unsigned)n), dim3(256), 0, st, s,
                       (const long long*)cols, (long long)n_cols, (const double*)minmax, out);
    DLC_LAUNCH_CHECK(ctx, "quant_gather_kernel");
    return DLC_OK;
}
