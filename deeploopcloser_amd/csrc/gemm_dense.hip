// Dense GEMM with fused bias + activation on the fp64 / fp32-input MFMA of
// gfx950: the SDAV / DA layers (sigmoid(x.W + b)) and the CnnVtl convolutions
// (im2col . HWIO + b, ReLU) in the reference's own arithmetic type.
//
//   fp64: v_mfma_f64_16x16x4_f64   C/D row = (lane>>4) + 4*reg, col = lane&15
//   fp32: v_mfma_f32_16x16x4_f32   C/D row = 4*(lane>>4) + reg, col = lane&15
//   A operand: lane holds A[row = lane&15][k = lane>>4]; B: B[k = lane>>4][col = lane&15]
//
// Workgroup tile 128 x 128, K step 16, 4 waves (2 x 2), each wave 64 x 64 =
// 4 x 4 MFMA tiles.  Operands are staged global -> registers -> LDS (padded
// rows, so arbitrary M/N/K and odd leading dimensions such as 1681 work with
// plain predicated loads); the next K tile's global loads are issued before
// the current tile's MFMAs.
#include "dlc_internal.h"

#include <algorithm>

namespace dlc_gemm {

constexpr int TM = 128, TN = 128, TK = 16;
constexpr int LDA_S = TK + 1;      // A tile [128][17]: conflict-free column-of-rows reads
constexpr int LDB_KN = TN + 16;    // B tile [16][144] for B stored [K,N]
constexpr int LDB_NK = TK + 1;     // B tile [128][17] for B stored [N,K]

template <typename T> struct Mma;
template <> struct Mma<double> {
    typedef f64x4_t acc_t;
    static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <> struct Mma<float> {
    typedef f32x4_t acc_t;
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};

template <typename T>
__device__ __forceinline__ T apply_act(T z, int act) {
    if (act == DLC_ACT_SIGMOID) return (T)1 / ((T)1 + exp(-z));   // tf.nn.sigmoid
    if (act == DLC_ACT_RELU) return z > (T)0 ? z : (T)0;
    return z;
}

// Implicit im2col (tf.layers.conv2d on NHWC, src/cnn_vtl/network/cnn_vtl.py:33-93): row m of the
// A operand is output pixel (img, oy, ox), column k is (ky, kx, c) with c fastest; A then points
// at the NHWC input and lda is unused.  With C % 8 == 0 a thread's 8 consecutive k are 8 channels of
// one input pixel (one 64-byte load); other C (conv1: 3) are loaded element by element.
struct ConvGeom {
    int H, W, C, KW, stride, pad_t, pad_l, OH, OW;
};

template <typename T>
struct Args {
    const T* A; long long lda;
    const T* B; long long ldb;
    const T* bias;
    T* C; long long ldc;
    long long M, N, K;
    int act;
    ConvGeom cv;
    long long kchunk;   // split-K: elements of K per chunk (a multiple of TK; >= K when one pass), chunk = blockIdx.z
    T* P;               // split-K partial results [chunks][M][N] (null when one pass)
};

// CONV: 0 plain GEMM; 1 implicit im2col, 8 consecutive channels of one pixel per thread (C % 8 == 0);
//       2 implicit im2col element by element (any C: conv1's 3 input channels)
template <typename T, int BLAYOUT, int CONV = 0>
__global__ __launch_bounds__(256, 2) void gemm_bias_act_kernel(Args<T> p) {
    constexpr int LDB_S = BLAYOUT == DLC_B_KN ? LDB_KN : LDB_NK;
    constexpr int B_ELEMS = BLAYOUT == DLC_B_KN ? TK * LDB_KN : TN * LDB_NK;
    __shared__ T As[TM * LDA_S];
    __shared__ T Bs[B_ELEMS];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const long long m0 = (long long)blockIdx.x * TM, n0 = (long long)blockIdx.y * TN;
    const long long kb = (long long)blockIdx.z * p.kchunk;          // this workgroup's K range [kb, kend)
    const long long kend = kb + p.kchunk < p.K ? kb + p.kchunk : p.K;

    // staging coordinates
    const int a_row = tid >> 1, a_k = (tid & 1) * 8;            // A: 2 threads per row, 8 k each
    const int bkn_k = tid >> 4, bkn_n = (tid & 15) * 8;         // B[K,N]: 16 threads per k row, 8 n each
    T ra[8], rb[8];

    // interior tiles (the common case) load without per-element predicates
    const bool rows_full = (m0 + TM <= p.M);
    const bool cols_full = (n0 + TN <= p.N);
    // implicit im2col: this thread's output pixel (fixed for the whole K loop)
    long long cv_img_base = 0;
    int cv_iy0 = 0, cv_ix0 = 0;
    if constexpr (CONV != 0) {
        const long long gm = m0 + a_row;
        const long long gmc = gm < p.M ? gm : p.M - 1;
        const long long img = gmc / ((long long)p.cv.OH * p.cv.OW);
        const int rem = (int)(gmc - img * p.cv.OH * p.cv.OW);
        const int oy = rem / p.cv.OW, ox = rem - oy * p.cv.OW;
        cv_iy0 = oy * p.cv.stride - p.cv.pad_t;
        cv_ix0 = ox * p.cv.stride - p.cv.pad_l;
        cv_img_base = img * (long long)p.cv.H * p.cv.W * p.cv.C;
    }
    // (ky, kx, c) of this thread's 8 channels in the NEXT tile to load: load_tile is called with
    // k0 = 0, TK, 2 TK, ... and steps them instead of dividing by C and KW in every K step
    int cv_c = 0, cv_kx = 0, cv_ky = 0;
    if constexpr (CONV != 0) {
        const long long t = (kb + a_k) / p.cv.C;
        cv_c = (int)(kb + a_k - t * p.cv.C);
        cv_ky = (int)(t / p.cv.KW);
        cv_kx = (int)(t - (long long)cv_ky * p.cv.KW);
    }
    auto load_tile = [&](long long k0) {
        const bool k_full = (k0 + TK <= p.K);
        const long long gm = m0 + a_row;
        if constexpr (CONV == 2) {
            // element e of this thread's 8 is (ky, kx, c) stepped e times from the first one
            int c = cv_c, kx = cv_kx, ky = cv_ky;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int iy = cv_iy0 + ky, ix = cv_ix0 + kx;
                const bool ok = gm < p.M && k0 + a_k + e < p.K && iy >= 0 && iy < p.cv.H && ix >= 0 && ix < p.cv.W;
                ra[e] = ok ? p.A[cv_img_base + ((long long)iy * p.cv.W + ix) * p.cv.C + c] : (T)0;
                if (++c == p.cv.C) { c = 0; if (++kx == p.cv.KW) { kx = 0; ++ky; } }
            }
            cv_c += TK;                                          // first element of the next tile
            while (cv_c >= p.cv.C) {
                cv_c -= p.cv.C;
                if (++cv_kx == p.cv.KW) { cv_kx = 0; ++cv_ky; }
            }
        } else if constexpr (CONV == 1) {
            const long long gk = k0 + a_k;                       // multiple of 8; 8 channels of one input pixel
            const int iy = cv_iy0 + cv_ky, ix = cv_ix0 + cv_kx;
            const bool ok = gm < p.M && gk < p.K && iy >= 0 && iy < p.cv.H && ix >= 0 && ix < p.cv.W;
            const int c = cv_c;
            cv_c += TK;
            while (cv_c >= p.cv.C) {
                cv_c -= p.cv.C;
                if (++cv_kx == p.cv.KW) { cv_kx = 0; ++cv_ky; }
            }
            if (ok) {
                const T* src = p.A + cv_img_base + ((long long)iy * p.cv.W + ix) * p.cv.C + c;
#pragma unroll
                for (int e = 0; e < 8; ++e) ra[e] = src[e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) ra[e] = (T)0;
            }
        } else if (rows_full && k_full) {
            const T* src = p.A + gm * p.lda + k0 + a_k;
#pragma unroll
            for (int e = 0; e < 8; ++e) ra[e] = src[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const long long gk = k0 + a_k + e;
                ra[e] = (gm < p.M && gk < p.K) ? p.A[gm * p.lda + gk] : (T)0;
            }
        }
        if constexpr (BLAYOUT == DLC_B_KN) {
            const long long gk = k0 + bkn_k;
            if (cols_full && k_full) {
                const T* src = p.B + gk * p.ldb + n0 + bkn_n;
#pragma unroll
                for (int e = 0; e < 8; ++e) rb[e] = src[e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const long long gn = n0 + bkn_n + e;
                    rb[e] = (gk < p.K && gn < p.N) ? p.B[gk * p.ldb + gn] : (T)0;
                }
            }
        } else {
            const long long gn = n0 + a_row;
            if (cols_full && k_full) {
                const T* src = p.B + gn * p.ldb + k0 + a_k;
#pragma unroll
                for (int e = 0; e < 8; ++e) rb[e] = src[e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const long long gk = k0 + a_k + e;
                    rb[e] = (gn < p.N && gk < p.K) ? p.B[gn * p.ldb + gk] : (T)0;
                }
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int e = 0; e < 8; ++e) As[a_row * LDA_S + a_k + e] = ra[e];
        if constexpr (BLAYOUT == DLC_B_KN) {
#pragma unroll
            for (int e = 0; e < 8; ++e) Bs[bkn_k * LDB_S + bkn_n + e] = rb[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) Bs[a_row * LDB_S + a_k + e] = rb[e];
        }
    };

    typename Mma<T>::acc_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (typename Mma<T>::acc_t){0, 0, 0, 0};

    const int fr = lane & 15, fk = lane >> 4;
    const long long nkt = (kend - kb + TK - 1) / TK;
    load_tile(kb);
    for (long long kt = 0; kt < nkt; ++kt) {
        store_tile();
        __syncthreads();
        if (kt + 1 < nkt) load_tile(kb + (kt + 1) * TK);
#pragma unroll
        for (int kk = 0; kk < TK / 4; ++kk) {
            T a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[(wr * 64 + i * 16 + fr) * LDA_S + kk * 4 + fk];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (BLAYOUT == DLC_B_KN) b[j] = Bs[(kk * 4 + fk) * LDB_S + wc * 64 + j * 16 + fr];
                else b[j] = Bs[(wc * 64 + j * 16 + fr) * LDB_S + kk * 4 + fk];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::run(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }

    if (p.P) {                                  // split-K: this chunk's raw partial tile
        T* part = p.P + (long long)blockIdx.z * p.M * p.N;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long gn = n0 + wc * 64 + j * 16 + fr;
            if (gn >= p.N) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long long gm = m0 + wr * 64 + i * 16 + Mma<T>::row(lane, r);
                    if (gm < p.M) part[gm * p.N + gn] = acc[i][j][r];
                }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long long gn = n0 + wc * 64 + j * 16 + fr;
        if (gn >= p.N) continue;
        const T bv = p.bias ? p.bias[gn] : (T)0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long gm = m0 + wr * 64 + i * 16 + Mma<T>::row(lane, r);
                if (gm < p.M) p.C[gm * p.ldc + gn] = apply_act<T>(acc[i][j][r] + bv, p.act);
            }
    }
}

// Split-K second pass: C = act(sum over chunks, in chunk order, + bias).
template <typename T>
__global__ __launch_bounds__(256) void splitk_bias_act_kernel(const T* __restrict__ P, int chunks, const T* __restrict__ bias,
                                                              T* __restrict__ C, long long ldc, long long M, long long N,
                                                              int act) {
    const long long total = M * N;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long r = e / N, c = e - r * N;
        T z = P[e];
        for (int s_ = 1; s_ < chunks; ++s_) z += P[(long long)s_ * total + e];
        C[r * ldc + c] = apply_act<T>(z + (bias ? bias[c] : (T)0), act);
    }
}

// Chunks of K (latency mode, needs the context's scratch; 1 = one pass).  A K step is 64 MFMAs per
// wave (1.9 us in fp64, 1.0 us in fp32) and one workgroup per CU already keeps the matrix pipes
// busy, so a launch costs about ceil(workgroups / 256) x K steps of that; fewer workgroups than
// CUs, or a little more than a whole number of rounds, waste the difference.  The plan minimises
// rounds x (K steps per chunk x t_step + 6 us) + the partial tiles written and read back + the
// second launch, and splits only for a clear gain.
template <typename T>
int plan_split(const dlc_ctx* ctx, int64_t M, int64_t N, int64_t K, long long* kchunk) {
    const int64_t wgs = dlc::cdiv(M, TM) * dlc::cdiv(N, TN), ksteps = dlc::cdiv(K, TK);
    *kchunk = ksteps * TK;
    if (!ctx->scratch || wgs >= 1024 || ksteps < 16) return 1;
    const double t_step = sizeof(T) == 8 ? 1.9 : 1.0, t_fix = 6.0, t_launch = 5.0, bytes_per_us = 3.0e6;
    const double part_bytes = (double)M * (double)N * sizeof(T);
    const int64_t fit = (int64_t)(ctx->scratch_bytes / ((size_t)M * (size_t)N * sizeof(T)));
    double best_t = (double)dlc::cdiv(wgs, (int64_t)256) * ((double)ksteps * t_step + t_fix);
    int64_t best_steps = ksteps;
    for (int64_t s_ = 2; s_ <= std::min<int64_t>(std::min<int64_t>(ksteps / 8, fit), 64); ++s_) {
        const int64_t steps = dlc::cdiv(ksteps, s_), chunks = dlc::cdiv(ksteps, steps);
        const double t = (double)dlc::cdiv(wgs * chunks, (int64_t)256) * ((double)steps * t_step + t_fix) +
                         2.0 * (double)chunks * part_bytes / bytes_per_us + t_launch;
        if (t < best_t * 0.9) { best_t = t; best_steps = steps; }
    }
    *kchunk = best_steps * TK;
    return (int)dlc::cdiv(ksteps, best_steps);
}

template <typename T>
int launch(dlc_ctx* ctx, int blayout, int act, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
           const void* B, int64_t ldb, const void* bias, void* C, int64_t ldc, hipStream_t st,
           const ConvGeom* cv = nullptr) {
    Args<T> a;
    a.A = (const T*)A; a.lda = lda; a.B = (const T*)B; a.ldb = ldb; a.bias = (const T*)bias;
    a.C = (T*)C; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.act = act;
    a.cv = cv ? *cv : ConvGeom{};
    if (dlc::cdiv(N, TN) > 65535 || dlc::cdiv(M, TM) > 0x7fffffffll)
        return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "gemm: M or N too large for one launch");
    const int chunks = plan_split<T>(ctx, M, N, K, &a.kchunk);
    a.P = chunks > 1 ? (T*)ctx->scratch : nullptr;
    dim3 grid((unsigned)dlc::cdiv(M, TM), (unsigned)dlc::cdiv(N, TN), (unsigned)chunks);
    // bench.py's kernel-only timing (dlc_set_profiling): an event pair around the GEMM kernel on its stream
    const int prof_slot = (int)(ctx->prof_calls % DLC_PROFILE_RING);
    if (ctx->profiling) DLC_HIP_CHECK(ctx, hipEventRecord(ctx->ev_start[prof_slot], st));
    if (cv && cv->C % 8 == 0) hipLaunchKernelGGL((gemm_bias_act_kernel<T, DLC_B_KN, 1>), grid, dim3(256), 0, st, a);
    else if (cv) hipLaunchKernelGGL((gemm_bias_act_kernel<T, DLC_B_KN, 2>), grid, dim3(256), 0, st, a);
    else if (blayout == DLC_B_KN) hipLaunchKernelGGL((gemm_bias_act_kernel<T, DLC_B_KN>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gemm_bias_act_kernel<T, DLC_B_NK>), grid, dim3(256), 0, st, a);
    DLC_LAUNCH_CHECK(ctx, "gemm_bias_act_kernel");
    if (ctx->profiling) {
        DLC_HIP_CHECK(ctx, hipEventRecord(ctx->ev_stop[prof_slot], st));
        ctx->prof_calls++;
    }
    if (chunks > 1) {
        long long blocks = dlc::cdiv(M * N, (int64_t)256);
        if (blocks > 256 * 32) blocks = 256 * 32;
        hipLaunchKernelGGL(splitk_bias_act_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, st, (const T*)a.P, chunks,
                           (const T*)bias, (T*)C, (long long)ldc, (long long)M, (long long)N, act);
        DLC_LAUNCH_CHECK(ctx, "splitk_bias_act_kernel");
    }
    return DLC_OK;
}

int gemm_bias_act(dlc_ctx* ctx, int dtype, int blayout, int act, int64_t M, int64_t N, int64_t K, const void* A,
                  int64_t lda, const void* B, int64_t ldb, const void* bias, void* C, int64_t ldc, hipStream_t st) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!A || !B || !C || M < 1 || N < 1 || K < 1) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "gemm: null/empty operand");
    if (blayout != DLC_B_KN && blayout != DLC_B_NK) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "gemm: blayout %d", blayout);
    if (act < DLC_ACT_NONE || act > DLC_ACT_RELU) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "gemm: act %d", act);
    if (lda < K || ldc < N || (blayout == DLC_B_KN ? ldb < N : ldb < K))
        return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "gemm: leading dimension smaller than the row");
    if (dtype == DLC_F64) return launch<double>(ctx, blayout, act, M, N, K, A, lda, B, ldb, bias, C, ldc, st);
    if (dtype == DLC_F32) return launch<float>(ctx, blayout, act, M, N, K, A, lda, B, ldb, bias, C, ldc, st);
    return dlc::fail(ctx, DLC_ERR_UNSUPPORTED, "gemm: dtype %d (need DLC_F64 or DLC_F32)", dtype);
}

int conv2d_f64(dlc_ctx* ctx, int act, int64_t M, int64_t N, int64_t K, const double* x, const double* w,
               const double* bias, double* out, const ConvGeom& cv, hipStream_t st) {
    return launch<double>(ctx, DLC_B_KN, act, M, N, K, x, 0, w, N, bias, out, N, st, &cv);
}

}  // namespace dlc_gemm

extern "C" int dlc_conv2d_nhwc_f64(dlc_ctx* ctx, const double* x, int64_t n, int h, int w, int c, const double* kernel,
                                   const double* bias, int kh, int kw, int cout, int stride, int pad_top, int pad_left,
                                   int oh, int ow, int act, double* out, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!x || !kernel || !out || n < 1 || h < 1 || w < 1 || c < 1 || kh < 1 || kw < 1 || cout < 1 || stride < 1 ||
        pad_top < 0 || pad_left < 0 || oh < 1 || ow < 1)
        return dlc::fail(ctx, DLC_ERR_BAD_ARG, "conv2d: bad argument");
    if (act < DLC_ACT_NONE || act > DLC_ACT_RELU) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "conv2d: act %d", act);
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    dlc_gemm::ConvGeom cv{h, w, c, kw, stride, pad_top, pad_left, oh, ow};
    return dlc_gemm::conv2d_f64(ctx, act, n * oh * ow, cout, (int64_t)kh * kw * c, x, kernel, bias, out, cv,
                                (hipStream_t)stream);
}

extern "C" int dlc_gemm_bias_act(dlc_ctx* ctx, int dtype, int blayout, int act, int64_t M, int64_t N, int64_t K,
                                 const void* A, int64_t lda, const void* B, int64_t ldb, const void* bias, void* C,
                                 int64_t ldc, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    return dlc_gemm::gemm_bias_act(ctx, dtype, blayout, act, M, N, K, A, lda, B, ldb, bias, C, ldc, (hipStream_t)stream);
}

namespace {
template <typename T>
__global__ __launch_bounds__(256) void bias_act_kernel(const T* __restrict__ A, long long lda, const T* __restrict__ bias,
                                                       T* __restrict__ C, long long ldc, long long M, long long N,
                                                       int act) {
    const long long total = M * N;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long r = e / N, c = e - r * N;
        const T z = A[r * lda + c] + (bias ? bias[c] : (T)0);
        C[r * ldc + c] = dlc_gemm::apply_act<T>(z, act);
    }
}
}  // namespace

extern "C" int dlc_bias_act(dlc_ctx* ctx, int dtype, int act, int64_t M, int64_t N, const void* A, int64_t lda,
                            const void* bias, void* C, int64_t ldc, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!A || !C || M < 1 || N < 1 || lda < N || ldc < N) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "bias_act: bad argument");
    if (act < DLC_ACT_NONE || act > DLC_ACT_RELU) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "bias_act: act %d", act);
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    long long blocks = dlc::cdiv(M * N, 256);
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (dtype == DLC_F64)
        hipLaunchKernelGGL(bias_act_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           (const double*)A, (long long)lda, (const double*)bias, (double*)C, (long long)ldc,
                           (long long)M, (long long)N, act);
    else if (dtype == DLC_F32)
        hipLaunchKernelGGL(bias_act_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           (const float*)A, (long long)lda, (const float*)bias, (float*)C, (long long)ldc, (long long)M,
                           (long long)N, act);
    else
        return dlc::fail(ctx, DLC_ERR_UNSUPPORTED, "bias_act: dtype %d", dtype);
    DLC_LAUNCH_CHECK(ctx, "bias_act_kernel");
    return DLC_OK;
}

static size_t elem_size(int dtype) { return dtype == DLC_F64 ? 8 : (dtype == DLC_F32 ? 4 : 0); }

extern "C" size_t dlc_sdav_encode_workspace_bytes(int64_t rows, const int64_t* dims, int n_layers, int dtype) {
    if (rows < 1 || !dims || n_layers < 1 || elem_size(dtype) == 0) return 0;
    int64_t wmax = 0;
    for (int l = 1; l < n_layers; ++l) wmax = dims[l] > wmax ? dims[l] : wmax;   // widths of the hidden hand-offs
    if (n_layers == 1) return 256;
    return 2 * dlc::align_up((size_t)rows * (size_t)wmax * elem_size(dtype), 256);
}

extern "C" int dlc_sdav_encode(dlc_ctx* ctx, int dtype, int64_t rows, int n_layers, const int64_t* dims, const void* x,
                               const void* const* W, const void* const* b, void* out, void* workspace,
                               size_t workspace_bytes, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!dims || !x || !W || !out || rows < 1 || n_layers < 1)
        return dlc::fail(ctx, DLC_ERR_BAD_ARG, "sdav_encode: null/empty argument");
    if (elem_size(dtype) == 0) return dlc::fail(ctx, DLC_ERR_UNSUPPORTED, "sdav_encode: dtype %d", dtype);
    for (int l = 0; l <= n_layers; ++l)
        if (dims[l] < 1) return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "sdav_encode: dims[%d] = %lld", l, (long long)dims[l]);
    const size_t need = dlc_sdav_encode_workspace_bytes(rows, dims, n_layers, dtype);
    if (n_layers > 1 && (!workspace || workspace_bytes < need))
        return dlc::fail(ctx, DLC_ERR_WORKSPACE, "sdav_encode: workspace %zu < %zu bytes", workspace_bytes, need);
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    char* ping[2] = {(char*)workspace, (char*)workspace + need / 2};
    const void* in = x;
    for (int l = 0; l < n_layers; ++l) {
        if (!W[l]) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "sdav_encode: W[%d] is null", l);
        void* o = (l == n_layers - 1) ? out : (void*)ping[l & 1];
        int rc = dlc_gemm::gemm_bias_act(ctx, dtype, DLC_B_KN, DLC_ACT_SIGMOID, rows, dims[l + 1], dims[l], in, dims[l],
                                         W[l], dims[l + 1], b ? b[l] : nullptr, o, dims[l + 1], (hipStream_t)stream);
        if (rc != DLC_OK) return rc;
        in = o;
    }
    return DLC_OK;
}
