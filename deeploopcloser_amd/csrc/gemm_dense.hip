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
// plain predicated loads).  The LDS image is double-buffered: the next K tile's
// global loads are issued before the current tile's MFMAs and stored into the
// other buffer after them -- one barrier per K tile.
//
// Tile order (one-pass launches): a 1-D grid whose workgroup ids -- dealt round-robin to
// the 8 XCDs, each with its own 4 MiB L2 -- are mapped so that the ~64 workgroups an
// XCD runs at a time (32 CUs x 2) form a compact block of br x bc tiles (8 x 8 when the
// matrix allows).  A row-major (x fastest) 3-D grid makes those 64 workgroups 64 different
// row tiles of ONE column tile: the B panel is shared, every A row tile is private, and A
// is re-streamed from HBM once per column tile (measured r02: 13.4 GB of fabric reads per
// SDAV layer for 0.69 GB of operands, L2 hit rate 60 %).  In a block every A and B K-slice
// is fetched once per 8 users.
#include "gemm_internal.h"

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

// Implicit im2col (ConvGeom, gemm_internal.h): with C % 8 == 0 a thread's 8 consecutive k are 8 channels of
// one input pixel (one 64-byte load); other C (conv1: 3) are loaded element by element.

template <typename T>
struct Args {
    const T* A; long long lda;
    const T* B; long long ldb;
    const T* bias;
    T* C; long long ldc;
    long long M, N, K;
    int act;
    ConvGeom cv;
    long long kchunk;   // split-K: elements of K per chunk (a multiple of TK; >= K when one pass)
    T* P;               // split-K partial results [chunks][M][N] (null when one pass)
    // tile order: tiles_m x tiles_n tiles (x chunks K chunks when split); one-pass launches walk blocks of
    // br x bc tiles (br * bc = 64), nbr x nbc of them, block gb = 8 * (local block of the XCD) + XCD
    long long tiles_m, tiles_n;
    int chunks, br, bc;
    long long nbr, nblocks;
    // triangular skip (the Gram blocks of the SDAV similarity, match_ref.hip): rows / columns are patches of
    // frames of tri_p patches, row r is global patch tri_row0 + r, column c global patch tri_col0 + c; only
    // (row frame < column frame) entries are ever read, so a tile that holds none is not computed.  0 = off.
    int tri_p;
    long long tri_row0, tri_col0;
};

// CONV: 0 plain GEMM; 1 implicit im2col, 8 consecutive channels of one pixel per thread (C % 8 == 0);
//       2 implicit im2col element by element (any C: conv1's 3 input channels)
template <typename T, int BLAYOUT, int CONV = 0>
__global__ __launch_bounds__(256, 2) void gemm_bias_act_kernel(Args<T> p) {
    constexpr int LDB_S = BLAYOUT == DLC_B_KN ? LDB_KN : LDB_NK;
    constexpr int B_ELEMS = BLAYOUT == DLC_B_KN ? TK * LDB_KN : TN * LDB_NK;
    constexpr int A_ELEMS = TM * LDA_S;
    extern __shared__ __attribute__((aligned(16))) char gemm_smem[];
    T* const As = (T*)gemm_smem;                 // [2][A_ELEMS]
    T* const Bs = As + 2 * A_ELEMS;              // [2][B_ELEMS]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    // ---- workgroup id -> (row tile, column tile, K chunk)
    long long tile_m, tile_n;
    int chunk = 0;
    {
        const long long w = blockIdx.x;
        if (p.chunks > 1) {                      // split-K (latency mode, a handful of tiles): plain order
            tile_m = w % p.tiles_m;
            tile_n = (w / p.tiles_m) % p.tiles_n;
            chunk = (int)(w / (p.tiles_m * p.tiles_n));
        } else {
            const long long l = w >> 3;                      // position in the XCD's own queue
            const long long gb = (l >> 6) * 8 + (w & 7);     // block: 8 consecutive ones run side by side, one per XCD
            if (gb >= p.nblocks) return;
            const int i = (int)(l & 63);
            tile_m = (gb % p.nbr) * p.br + (i % p.br);
            tile_n = (gb / p.nbr) * p.bc + (i / p.br);
            if (tile_m >= p.tiles_m || tile_n >= p.tiles_n) return;      // block padding (before any barrier)
        }
    }
    const long long m0 = tile_m * TM, n0 = tile_n * TN;
    if (p.tri_p > 0 && (p.tri_col0 + n0 + TN - 1) / p.tri_p <= (p.tri_row0 + m0) / p.tri_p) return;   // no (row frame < column frame) pair
    const long long kb = (long long)chunk * p.kchunk;               // this workgroup's K range [kb, kend)
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
    // Branch-free loader for interior tiles and whole K tiles (the main loop's form): no per-element predicates,
    // the one per-thread condition there is (a convolution's padding / image border) selects a safe address and
    // zeroes the result instead of branching, so that the iteration stays ONE basic block and its loads, LDS
    // stores and address arithmetic can be scheduled into the gaps of the 64 MFMAs.
    auto load_tile_fast = [&](long long k0) {
        const long long gm = m0 + a_row;
        if constexpr (CONV == 1) {
            const int iy = cv_iy0 + cv_ky, ix = cv_ix0 + cv_kx;
            const bool ok = iy >= 0 && iy < p.cv.H && ix >= 0 && ix < p.cv.W;
            const T* src = p.A + (ok ? cv_img_base + ((long long)iy * p.cv.W + ix) * p.cv.C + cv_c : 0);
            cv_c += TK;
            while (cv_c >= p.cv.C) {
                cv_c -= p.cv.C;
                if (++cv_kx == p.cv.KW) { cv_kx = 0; ++cv_ky; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const T v = src[e];
                ra[e] = ok ? v : (T)0;
            }
        } else {
            const T* src = p.A + gm * p.lda + k0 + a_k;
#pragma unroll
            for (int e = 0; e < 8; ++e) ra[e] = src[e];
        }
        if constexpr (BLAYOUT == DLC_B_KN) {
            const T* src = p.B + (k0 + bkn_k) * p.ldb + n0 + bkn_n;
#pragma unroll
            for (int e = 0; e < 8; ++e) rb[e] = src[e];
        } else {
            const T* src = p.B + (n0 + a_row) * p.ldb + k0 + a_k;
#pragma unroll
            for (int e = 0; e < 8; ++e) rb[e] = src[e];
        }
    };
    auto store_tile = [&](int buf) {
        T* as = As + buf * A_ELEMS;
        T* bs = Bs + buf * B_ELEMS;
#pragma unroll
        for (int e = 0; e < 8; ++e) as[a_row * LDA_S + a_k + e] = ra[e];
        if constexpr (BLAYOUT == DLC_B_KN) {
#pragma unroll
            for (int e = 0; e < 8; ++e) bs[bkn_k * LDB_S + bkn_n + e] = rb[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) bs[a_row * LDB_S + a_k + e] = rb[e];
        }
    };

    typename Mma<T>::acc_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (typename Mma<T>::acc_t){0, 0, 0, 0};

    const int fr = lane & 15, fk = lane >> 4;
    const long long nkt = (kend - kb + TK - 1) / TK;
    // Software pipeline (one barrier per K tile; the two LDS buffers alternate):
    //   iteration kt:  MFMAs of k-slices 0,1 of tile kt | registers (tile kt+1, loaded during iteration kt-1) -> the
    //                  other LDS buffer | global loads of tile kt+2 -> registers | MFMAs of k-slices 2,3 | barrier
    // so the LDS stores and the address arithmetic / issue of the global loads sit between two halves of a tile's 64
    // MFMAs (4096 matrix-pipe cycles in fp64) instead of between two tiles with the pipe idle, and a global load has a
    // whole iteration to land.  Reads of buffer b in iteration kt and writes to it in iteration kt+1 are separated by
    // the barrier at the end of iteration kt.
    auto compute = [&](const T* as, const T* bs, int kk) {
        T a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = as[(wr * 64 + i * 16 + fr) * LDA_S + kk * 4 + fk];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (BLAYOUT == DLC_B_KN) b[j] = bs[(kk * 4 + fk) * LDB_S + wc * 64 + j * 16 + fr];
            else b[j] = bs[(wc * 64 + j * 16 + fr) * LDB_S + kk * 4 + fk];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::run(a[i], b[j], acc[i][j]);
    };
    load_tile(kb);
    store_tile(0);
    if (nkt > 1) load_tile(kb + TK);
    __syncthreads();
    long long kt = 0;
    if (CONV != 2 && rows_full && cols_full) {
        // iterations whose load (K tile kt+2) is a whole tile inside this workgroup's K range: one basic block each
        const long long n_fast = (kend - kb) / TK - 2;
        for (; kt < n_fast; ++kt) {
            const int cur = (int)(kt & 1);
            const T* as = As + cur * A_ELEMS;
            const T* bs = Bs + cur * B_ELEMS;
            compute(as, bs, 0);
            compute(as, bs, 1);
            store_tile(cur ^ 1);
            load_tile_fast(kb + (kt + 2) * TK);
            compute(as, bs, 2);
            compute(as, bs, 3);
            // the schedule asked of the compiler: the LDS stores of the next tile in the gaps of the first 32 MFMAs,
            // the global loads of the one after it (and their address arithmetic) in the gaps of the last 32
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);       // DS read: the fragments of k-slices 0, 1
#pragma unroll
            for (int g_ = 0; g_ < 8; ++g_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // MFMA
                __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);   // DS write
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read: the fragments of k-slices 2, 3, ahead of use
            }
#pragma unroll
            for (int g_ = 0; g_ < 8; ++g_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // MFMA
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // VALU
            }
            __syncthreads();
        }
    }
    for (; kt < nkt; ++kt) {
        const int cur = (int)(kt & 1);
        const T* as = As + cur * A_ELEMS;
        const T* bs = Bs + cur * B_ELEMS;
        compute(as, bs, 0);
        compute(as, bs, 1);
        if (kt + 1 < nkt) store_tile(cur ^ 1);
        if (kt + 2 < nkt) load_tile(kb + (kt + 2) * TK);
        compute(as, bs, 2);
        compute(as, bs, 3);
        __syncthreads();
    }

    if (p.P) {                                  // split-K: this chunk's raw partial tile
        T* part = p.P + (long long)chunk * p.M * p.N;
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

template <typename T, int BLAYOUT, int CONV>
int launch_kernel(dlc_ctx* ctx, const Args<T>& a, long long nwg, hipStream_t st) {
    constexpr int B_ELEMS = BLAYOUT == DLC_B_KN ? TK * LDB_KN : TN * LDB_NK;
    constexpr int LDS = 2 * (TM * LDA_S + B_ELEMS) * (int)sizeof(T);      // 70 KiB in fp64: two workgroups per CU
    auto kern = gemm_bias_act_kernel<T, BLAYOUT, CONV>;
    constexpr int bit = DLC_ATTR_DGEMM_BASE + (sizeof(T) == 8 ? 0 : 4) + (CONV != 0 ? 1 + CONV : (BLAYOUT == DLC_B_KN ? 0 : 1));
    const unsigned long long m = 1ull << bit;
    if (LDS > 48 * 1024 && !(ctx->func_attr_set & m)) {                   // per device: the context's own flag
        DLC_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        ctx->func_attr_set |= m;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), LDS, st, a);
    return DLC_OK;
}

// Few 64-row tiles and a long K (the training step's 300-row products, an encode of a few frames in latency mode): split-K
// on the LDS-DMA kernel -- its K loop is the faster one, and 300 rows pad to 320 there instead of 384 -- then ONE reducing
// pass that sums the chunks in chunk order and applies bias + activation.  A 64-row workgroup keeps a CU's fp64 matrix
// pipes busy on its own, so a launch costs about (its MFMA work) / (the CUs it reaches): the chunk count is the one that
// fills the 512 slots (two such workgroups per CU) once.  Needs the context's scratch (latency mode; the training step
// lends it).  K: the reduction length as A has it (even), Kb <= K as B has it.  DLC_OK, 1 = not taken, < 0 = error.
static int dma_splitk_f64(dlc_ctx* ctx, int blayout, int act, int64_t M, int64_t N, int64_t K, int64_t Kb, const double* A,
                          int64_t lda, const double* B, int64_t ldb, const double* bias, double* C, int64_t ldc, hipStream_t st) {
    if (!ctx->scratch || N <= 96) return 1;
    const int64_t tiles = dlc::cdiv(M, (int64_t)64) * dlc::cdiv(N, (int64_t)128), nkt = dlc::cdiv(K, (int64_t)16);
    if (tiles >= 256) return 1;
    const int64_t fit = (int64_t)(ctx->scratch_bytes / ((size_t)M * (size_t)N * sizeof(double)));
    const int64_t want = std::min<int64_t>(std::min<int64_t>(512 / tiles, nkt / 8), std::min<int64_t>(fit, 64));
    if (want < 2) return 1;
    const int64_t kchunk = dlc::cdiv(nkt, want) * 16, chunks = dlc::cdiv(K, kchunk);
    if (chunks < 2) return 1;
    double* part = (double*)ctx->scratch;
    if (gemm_dma_f64_splitk(ctx, blayout, M, N, K, Kb, A, lda, B, ldb, part, kchunk, st, true) != DLC_OK) return 1;
    const int prof_slot = (int)(ctx->prof_calls % DLC_PROFILE_RING);
    if (ctx->profiling) DLC_HIP_CHECK(ctx, hipEventRecord(ctx->ev_start[prof_slot], st));
    const int rc = gemm_dma_f64_splitk(ctx, blayout, M, N, K, Kb, A, lda, B, ldb, part, kchunk, st, false);
    if (rc != DLC_OK) return rc < 0 ? rc : dlc::fail(ctx, DLC_ERR_HIP, "gemm: the split-K launch was refused after its dry run");
    if (ctx->profiling) {
        DLC_HIP_CHECK(ctx, hipEventRecord(ctx->ev_stop[prof_slot], st));
        ctx->prof_calls++;
    }
    long long blocks = dlc::cdiv(M * N, (int64_t)256);
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(splitk_bias_act_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, st, (const double*)part, (int)chunks, bias, C,
                       (long long)ldc, (long long)M, (long long)N, act);
    DLC_LAUNCH_CHECK(ctx, "splitk_bias_act_kernel");
    return DLC_OK;
}

template <typename T>
int launch(dlc_ctx* ctx, int blayout, int act, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
           const void* B, int64_t ldb, const void* bias, void* C, int64_t ldc, hipStream_t st,
           const ConvGeom* cv = nullptr, const TriSkip* tri = nullptr) {
    Args<T> a;
    a.A = (const T*)A; a.lda = lda; a.B = (const T*)B; a.ldb = ldb; a.bias = (const T*)bias;
    a.C = (T*)C; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.act = act;
    a.cv = cv ? *cv : ConvGeom{};
    a.tri_p = tri ? tri->p : 0;
    a.tri_row0 = tri ? tri->row0 : 0;
    a.tri_col0 = tri ? tri->col0 : 0;
    a.tiles_m = dlc::cdiv(M, TM);
    a.tiles_n = dlc::cdiv(N, TN);
    if constexpr (sizeof(T) == 8) {
        if (!cv && !tri) {
            const int rc_sk = dma_splitk_f64(ctx, blayout, act, M, N, K, K, (const double*)A, lda, (const double*)B, ldb,
                                             (const double*)bias, (double*)C, ldc, st);
            if (rc_sk <= 0) return rc_sk;
        }
    }
    const int chunks = plan_split<T>(ctx, M, N, K, &a.kchunk);
    if constexpr (sizeof(T) == 8) {
        // large aligned fp64 launches: the LDS-DMA kernel (gemm_dma_f64.hip); anything else stays here
        if (chunks == 1 && (!cv || cv->C % 8 == 0)) {
            const int rc_dma = launch_dma_f64(ctx, blayout, act, M, N, K, (const double*)A, lda, (const double*)B, ldb,
                                              (const double*)bias, (double*)C, ldc, st, cv, tri);
            if (rc_dma <= 0) return rc_dma;
        }
    }
    // the kernel below does not fold per-image minima / maxima in its epilogue: a pass over its output does (end of this function)
    unsigned long long* const fold_keys = cv ? cv->mm_keys : nullptr;
    a.chunks = chunks;
    a.P = chunks > 1 ? (T*)ctx->scratch : nullptr;
    // block of 64 tiles: 8 x 8, narrower along a dimension with fewer than 8 tiles (powers of two)
    int bc = 8;
    while (bc > 1 && bc / 2 >= a.tiles_n) bc /= 2;
    int br = 64 / bc;
    if (a.tiles_m < br) {                                    // few row tiles: widen along N instead
        br = 1;
        while (br < a.tiles_m) br *= 2;
        bc = 64 / br;
    }
    a.br = br; a.bc = bc;
    a.nbr = dlc::cdiv(a.tiles_m, (int64_t)br);
    a.nblocks = a.nbr * dlc::cdiv(a.tiles_n, (int64_t)bc);
    const long long nwg = chunks > 1 ? a.tiles_m * a.tiles_n * chunks : dlc::cdiv(a.nblocks, (int64_t)8) * 8 * 64;
    if (nwg > 0x7fffffffll) return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "gemm: M x N too large for one launch");
    // bench.py's kernel-only timing (dlc_set_profiling): an event pair around the GEMM kernel on its stream
    const int prof_slot = (int)(ctx->prof_calls % DLC_PROFILE_RING);
    if (ctx->profiling) DLC_HIP_CHECK(ctx, hipEventRecord(ctx->ev_start[prof_slot], st));
    int rc;
    if (cv && cv->C % 8 == 0) rc = launch_kernel<T, DLC_B_KN, 1>(ctx, a, nwg, st);
    else if (cv) rc = launch_kernel<T, DLC_B_KN, 2>(ctx, a, nwg, st);
    else if (blayout == DLC_B_KN) rc = launch_kernel<T, DLC_B_KN, 0>(ctx, a, nwg, st);
    else rc = launch_kernel<T, DLC_B_NK, 0>(ctx, a, nwg, st);
    if (rc != DLC_OK) return rc;
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
    if constexpr (sizeof(T) == 8) {
        if (fold_keys) {
            if (ldc != N) return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "conv2d: per-image statistics need a contiguous output");
            const int64_t per_img = (int64_t)cv->OH * cv->OW;
            return dlc_cnn::fold_minmax_f64(ctx, (const double*)C, M / per_img, per_img * N, fold_keys, st);
        }
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

// Gram block of the SDAV similarity (match_ref.hip): C = A . B^T in fp64 (B stored [N,K], or its transpose stored
// [K,N]), tiles that hold no (row frame < column frame) entry skipped (their part of C stays unwritten, never read).
int gram_upper_f64(dlc_ctx* ctx, int blayout, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda,
                   const double* B, int64_t ldb, double* C, int64_t ldc, int patches, int64_t row0, int64_t col0,
                   hipStream_t st) {
    TriSkip tri{patches, row0, col0};
    return launch<double>(ctx, blayout, DLC_ACT_NONE, M, N, K, A, lda, B, ldb, nullptr, C, ldc, st, nullptr, &tri);
}

int gemm_bias_act_padded_f64(dlc_ctx* ctx, int act, int64_t M, int64_t N, int64_t K, int64_t Kpad, const double* A,
                             const double* B, int64_t ldb, const double* bias, double* C, int64_t ldc, hipStream_t st) {
    const int rc_sk = dma_splitk_f64(ctx, DLC_B_KN, act, M, N, Kpad, K, A, Kpad, B, ldb, bias, C, ldc, st);
    if (rc_sk <= 0) return rc_sk;
    const int rc = launch_dma_f64(ctx, DLC_B_KN, act, M, N, Kpad, A, Kpad, B, ldb, bias, C, ldc, st, nullptr, nullptr, K);
    if (rc <= 0) return rc;
    return launch<double>(ctx, DLC_B_KN, act, M, N, K, A, Kpad, B, ldb, bias, C, ldc, st);
}

int conv2d_f64(dlc_ctx* ctx, int act, int64_t M, int64_t N, int64_t K, const double* x, const double* w,
               const double* bias, double* out, const ConvGeom& cv, hipStream_t st) {
    return launch<double>(ctx, DLC_B_KN, act, M, N, K, x, 0, w, N, bias, out, N, st, &cv);
}

}  // namespace dlc_gemm

extern "C" int dlc_conv2d_nhwc_f64_stats(dlc_ctx* ctx, const double* x, int64_t n, int h, int w, int c, const double* kernel,
                                         const double* bias, int kh, int kw, int cout, int stride, int pad_top, int pad_left,
                                         int oh, int ow, int act, double* out, uint64_t* frame_keys, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!x || !kernel || !out || n < 1 || h < 1 || w < 1 || c < 1 || kh < 1 || kw < 1 || cout < 1 || stride < 1 ||
        pad_top < 0 || pad_left < 0 || oh < 1 || ow < 1)
        return dlc::fail(ctx, DLC_ERR_BAD_ARG, "conv2d: bad argument");
    if (act < DLC_ACT_NONE || act > DLC_ACT_RELU) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "conv2d: act %d", act);
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    dlc_gemm::ConvGeom cv{h, w, c, kw, stride, pad_top, pad_left, oh, ow, (unsigned long long*)frame_keys};
    return dlc_gemm::conv2d_f64(ctx, act, n * oh * ow, cout, (int64_t)kh * kw * c, x, kernel, bias, out, cv,
                                (hipStream_t)stream);
}

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
    dlc_gemm::ConvGeom cv{h, w, c, kw, stride, pad_top, pad_left, oh, ow, nullptr};
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

// Input rows of an ODD width (SDAV: 1681 = 41 x 41 pixels) are 8-byte aligned only, which keeps layer 0 off the
// LDS-DMA kernel; for batches large enough for that kernel the input is first copied into zero-padded rows of a
// multiple of 16 columns (one extra pass over x, ~1 % of the layer's time).
static int64_t sdav_pad_width(int64_t rows, const int64_t* dims, int dtype) {
    if (dtype != DLC_F64 || (dims[0] & 1) == 0 || (dims[1] & 1) || rows * dims[1] < (int64_t)512 * 256 * 128) return 0;
    return (dims[0] + 15) / 16 * 16;
}

namespace {
__global__ __launch_bounds__(256) void pad_rows_kernel(const double* __restrict__ x, long long rows, long long cols,
                                                       long long ldp, double* __restrict__ out) {
    const long long total = rows * ldp;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long r = e / ldp, c = e - r * ldp;
        out[e] = c < cols ? x[r * cols + c] : 0.0;
    }
}
}  // namespace

extern "C" size_t dlc_sdav_encode_workspace_bytes(int64_t rows, const int64_t* dims, int n_layers, int dtype) {
    if (rows < 1 || !dims || n_layers < 1 || elem_size(dtype) == 0) return 0;
    int64_t wmax = 0;
    for (int l = 1; l < n_layers; ++l) wmax = dims[l] > wmax ? dims[l] : wmax;   // widths of the hidden hand-offs
    const size_t pad = dlc::align_up((size_t)rows * (size_t)sdav_pad_width(rows, dims, dtype) * 8, 256);
    if (n_layers == 1) return 256 + pad;
    return 2 * dlc::align_up((size_t)rows * (size_t)wmax * elem_size(dtype), 256) + pad;
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
    if ((n_layers > 1 || sdav_pad_width(rows, dims, dtype) > 0) && (!workspace || workspace_bytes < need))
        return dlc::fail(ctx, DLC_ERR_WORKSPACE, "sdav_encode: workspace %zu < %zu bytes", workspace_bytes, need);
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    const int64_t kpad = sdav_pad_width(rows, dims, dtype);
    const size_t pad_bytes = dlc::align_up((size_t)rows * (size_t)kpad * 8, 256);
    const size_t half = (need - pad_bytes) / 2;
    char* ping[2] = {(char*)workspace, (char*)workspace + half};
    double* xpad = (double*)((char*)workspace + 2 * half);
    const void* in = x;
    for (int l = 0; l < n_layers; ++l) {
        if (!W[l]) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "sdav_encode: W[%d] is null", l);
        void* o = (l == n_layers - 1) ? out : (void*)ping[l & 1];
        int rc;
        if (l == 0 && kpad > 0) {
            long long blocks = dlc::cdiv(rows * kpad, (int64_t)256);
            if (blocks > 256 * 64) blocks = 256 * 64;
            hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const double*)x,
                               (long long)rows, (long long)dims[0], (long long)kpad, xpad);
            DLC_LAUNCH_CHECK(ctx, "pad_rows_kernel");
            rc = dlc_gemm::gemm_bias_act_padded_f64(ctx, DLC_ACT_SIGMOID, rows, dims[1], dims[0], kpad, xpad, (const double*)W[0],
                                                    dims[1], b ? (const double*)b[0] : nullptr, (double*)o, dims[1],
                                                    (hipStream_t)stream);
        } else {
            rc = dlc_gemm::gemm_bias_act(ctx, dtype, DLC_B_KN, DLC_ACT_SIGMOID, rows, dims[l + 1], dims[l], in, dims[l],
                                         W[l], dims[l + 1], b ? b[l] : nullptr, o, dims[l + 1], (hipStream_t)stream);
        }
        if (rc != DLC_OK) return rc;
        in = o;
    }
    return DLC_OK;
}
