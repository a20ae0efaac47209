// The SDAV similarity's patch matching as an exact-arithmetic FILTER (match_ref.hip, SimilarityCalculator.py:30-37).
//
// What the reference needs from the 31 890 x 31 890 patch products is one thing only: for patch a of frame i, WHICH
// patch b of frame j is nearest (np.argmin of the norms).  The distances themselves never reach the result.  So the
// Gram matrix does not have to be an fp64 product -- it has to decide the arg-min, and say when it cannot:
//
//   u = (x - lo) / (hi - lo) in [0, 1]   (lo / hi: the dataset's extremes; arg-min of |u_a - u_b| = arg-min of |x_a - x_b|)
//   q = floor(u * 2^21) as three 7-bit slices q1 q2 q3 (non-negative int8)
//   acc = C2 + floor((C3 + floor(C4 / 128)) / 128),  Cc = sum over s + t = c of q_s . q_t    (v_mfma_i32_16x16x64_i8: exact)
//
// acc * 2^-14 is a LOWER bound of u_a . u_b that misses at most
//   E = 2^-21 (sum u_a + sum u_b)  [truncation of u]  +  H * 127^2 * (2^-34 + 2^-42)  [the dropped classes 5 and 6]  +  2^-13,
// a rigorous bound with no rounding in it (integer accumulation).  The pair kernel takes the arg-min of
// |u_b|^2 - 2 acc 2^-14 and accepts it when the runner-up is more than 2 E away; otherwise it evaluates the candidates
// inside that window directly in fp64 from the descriptors.  Six int8 products of K = H replace one fp64 product:
// 1/5 of the matrix-pipe time at the int8 rate, and the result is the arg-min of the true distances either way.
//
// Layout: X [rows_pad, 3 Kp] = (q1 | q2 | q3), Y [rows_pad, 3 Kp] = (q3 | q2 | q1), Kp = H rounded up to 128, zero
// padded; then class 4 is X[:, 0:3Kp] . Y[:, 0:3Kp]^T, class 3 is X[:, 0:2Kp] . Y[:, Kp:3Kp]^T, class 2 is
// X[:, 0:Kp] . Y[:, 2Kp:3Kp]^T: one NT int8 GEMM over three K segments with the accumulators shifted right by 7 between
// them.
#include "gemm_internal.h"

namespace dlc_gemm {
namespace {

typedef __attribute__((ext_vector_type(4))) int v4i;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int GI_T = 256;                 // tile: 256 row patches x 256 column patches, 4 waves of 128 x 128
constexpr int GI_KS = 128;                // bytes of K per LDS stage (two MFMA k-steps of 64)
constexpr int GI_HALF = GI_T * GI_KS;     // one operand's part of a stage: 32 KiB
constexpr int GI_STAGE = 2 * GI_HALF;
constexpr int GI_NSTAGE = 2;

struct GramI8Args {
    const char* X;                        // row panel: X + row0 * pitch
    const char* Y;                        // column panel: Y + col0 * pitch
    int* out;                             // [mrows, ldo] accumulators (units of 2^-14 in u . u)
    long long ldo, mrows, ncols;
    int pitch, kp;
    int tiles_m, tiles_n, nsn, nsup;
    int tri_p;
    long long tri_row0, tri_col0;
};

// eight 1 KiB LDS-DMA pieces: four 16-row groups (per-lane offsets o0..o3) x two 64-byte k-steps (a second scalar base:
// an immediate offset would move the LDS destination as well)
__device__ __forceinline__ void dma8(unsigned o0, unsigned o1, unsigned o2, unsigned o3, const char* base, unsigned lds) {
    unsigned keep;
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %6\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %5\n\t"
        "s_add_u32 m0, %6, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %7\n\t"
        "s_add_u32 m0, %6, 0x800\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %5\n\t"
        "s_add_u32 m0, %6, 0xc00\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %7\n\t"
        "s_add_u32 m0, %6, 0x1000\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, %5\n\t"
        "s_add_u32 m0, %6, 0x1400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, %7\n\t"
        "s_add_u32 m0, %6, 0x1800\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %4, %5\n\t"
        "s_add_u32 m0, %6, 0x1c00\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %4, %7\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(base), "s"(lds), "s"(base + 64)
        : "memory", "scc");
}

__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}

// LDS stage: the row panel's 16 groups of 16 rows, each two 1 KiB blocks (k-step 0 / 1) in MFMA operand order -- lane l
// of a block holds row l % 16, bytes (l / 16) * 16 .. + 15 of the k-step -- then the column panel's the same.  A
// fragment read is one ds_read_b128 at block + lane * 16.
__global__ __launch_bounds__(256) void gram_i8_kernel(const GramI8Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem_i8[];
    // workgroup -> tile: ids go round-robin to the 8 XCDs; an XCD's 32 resident workgroups take one 4 x 8 block of
    // tiles (12 panels feed 32 tiles out of that XCD's L2)
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3, local = slot & 31;
    const int sup = (slot >> 5) * 8 + xcd;
    if (sup >= p.nsup) return;
    const int tile_m = (sup / p.nsn) * 4 + (local >> 3), tile_n = (sup % p.nsn) * 8 + (local & 7);
    if (tile_m >= p.tiles_m || tile_n >= p.tiles_n) return;
    const long long m0 = (long long)tile_m * GI_T, n0 = (long long)tile_n * GI_T;
    if ((p.tri_col0 + n0 + GI_T - 1) / p.tri_p <= (p.tri_row0 + m0) / p.tri_p) return;   // no (row frame < column frame) entry

    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wr = w >> 1, wc = w & 1;
    const unsigned lds_base = (unsigned)(unsigned long long)(lptr_t)smem_i8;
    unsigned off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) off[j] = (unsigned)(((w * 4 + j) * 16 + (lane & 15)) * p.pitch + (lane >> 4) * 16);
    const char* xb = p.X + m0 * p.pitch;
    const char* yb = p.Y + n0 * p.pitch;
    const int n128 = p.kp / GI_KS;
    const int nst = 6 * n128;

    auto issue = [&](int t, int stage) {
        int xo, yo;
        if (t < 3 * n128) { xo = t * GI_KS; yo = xo; }
        else if (t < 5 * n128) { xo = (t - 3 * n128) * GI_KS; yo = p.kp + xo; }
        else { xo = (t - 5 * n128) * GI_KS; yo = 2 * p.kp + xo; }
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + stage * GI_STAGE + w * 8192);
        dma8(off[0], off[1], off[2], off[3], uniform_ptr(xb + xo), dst);
        dma8(off[0], off[1], off[2], off[3], uniform_ptr(yb + yo), dst + GI_HALF);
    };

    v4i acc[8][8];                          // [column group j of this wave][row group i]
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = v4i{0, 0, 0, 0};

    // the K stages of one segment (the accumulators stay in place: a branch around the shift inside ONE loop made hipcc
    // copy all 256 of them out of the accumulation registers at the top of every iteration)
    auto run = [&](int t_lo, int t_hi) {
        for (int t = t_lo; t < t_hi; ++t) {
            const int stage = t & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // stage t is visible to every wave; the other stage's readers are through
            asm volatile("" ::: "memory");
            if (t + 1 < nst) issue(t + 1, stage ^ 1);
            const char* sx = smem_i8 + stage * GI_STAGE + (wr * 8) * 2048 + lane * 16;
            const char* sy = smem_i8 + stage * GI_STAGE + GI_HALF + (wc * 8) * 2048 + lane * 16;
#pragma unroll 1
            for (int s = 0; s < 2; ++s) {
                v4i fx[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) fx[i] = *(const v4i*)(sx + i * 2048 + s * 1024);
                v4i fy = *(const v4i*)(sy + s * 1024);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const v4i fyn = j < 7 ? *(const v4i*)(sy + (j + 1) * 2048 + s * 1024) : fy;
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        acc[j][i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fy, fx[i], acc[j][i], 0, 0, 0);
                    fy = fyn;
                }
            }
        }
    };
    auto shift = [&]() {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[j][i] = acc[j][i] >> 7;
    };
    issue(0, 0);
    run(0, 3 * n128);
    shift();
    run(3 * n128, 5 * n128);
    shift();
    run(5 * n128, nst);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // D[m][n] of MFMA (j, i): m = column patch (wc * 8 + j) * 16 + (lane / 16) * 4 + v, n = row patch (wr * 8 + i) * 16 + lane % 16
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const long long r = m0 + (wr * 8 + i) * 16 + (lane & 15);
        if (r >= p.mrows) continue;
        int* orow = p.out + r * p.ldo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const long long c = n0 + (wc * 8 + j) * 16 + (lane >> 4) * 4;
            if (c + 3 < p.ldo) *(v4i*)(orow + c) = acc[j][i];          // ldo is a multiple of 4 >= ncols
        }
    }
}

// ---- range, quantisation -------------------------------------------------------------------------------------------

__global__ void sim_keys_init_kernel(unsigned long long* keys) {
    if (threadIdx.x < 6) keys[threadIdx.x] = threadIdx.x == 0 ? ~0ull : 0ull;     // [4]: direct evaluations (a count for experiments)
}

__global__ __launch_bounds__(256) void sim_range_kernel(const double* __restrict__ x, long long n, unsigned long long* keys) {
    double lo = INFINITY, hi = -INFINITY;
    bool bad = false;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double v = x[i];
        bad |= !(fabs(v) < INFINITY);
        lo = fmin(lo, v); hi = fmax(hi, v);
    }
    for (int o = 32; o > 0; o >>= 1) { lo = fmin(lo, __shfl_xor(lo, o)); hi = fmax(hi, __shfl_xor(hi, o)); }
    const bool any_bad = __ballot(bad) != 0;
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&keys[0], dlc_f64_key(lo));
        atomicMax(&keys[1], dlc_f64_key(hi));
        if (any_bad) atomicMax(&keys[2], 1ull);
    }
}

// One wave per patch row: the three slices into X and Y (a lane packs 4 consecutive k into a word), sum u and |u|^2.
// keys[3]: the largest row sum (ordered key).
__global__ __launch_bounds__(256) void sim_quant_kernel(const double* __restrict__ desc, long long rows, int H, int kp,
                                                        unsigned long long* keys, char* __restrict__ X, char* __restrict__ Y,
                                                        double* __restrict__ nu2) {
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    const double lo = dlc_f64_unkey(keys[0]), hi = dlc_f64_unkey(keys[1]);
    const double range = hi - lo;
    const double inv = range > 0.0 ? 1.0 / range : 0.0;
    const double* x = desc + r * H;
    unsigned* xr = (unsigned*)(X + r * 3ll * kp);
    unsigned* yr = (unsigned*)(Y + r * 3ll * kp);
    const int wpk = kp / 4;                 // words per slice
    double su = 0.0, s2 = 0.0;
    for (int wd = lane; wd < wpk; wd += 64) {
        unsigned w1 = 0, w2 = 0, w3 = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = wd * 4 + e;
            if (k < H) {
                double u = (x[k] - lo) * inv;
                u = u > 0.0 ? (u < 1.0 ? u : 1.0) : 0.0;            // (a NaN lands on 0; such datasets take the fp64 route)
                su += u; s2 = fma(u, u, s2);
                int q = (int)(u * 2097152.0);                       // floor (u >= 0)
                q = q > 2097151 ? 2097151 : q;
                w1 |= (unsigned)(q >> 14) << (8 * e);
                w2 |= (unsigned)((q >> 7) & 127) << (8 * e);
                w3 |= (unsigned)(q & 127) << (8 * e);
            }
        }
        xr[wd] = w1; xr[wpk + wd] = w2; xr[2 * wpk + wd] = w3;
        yr[wd] = w3; yr[wpk + wd] = w2; yr[2 * wpk + wd] = w1;
    }
    for (int o = 32; o > 0; o >>= 1) { su += __shfl_xor(su, o); s2 += __shfl_xor(s2, o); }
    if (lane == 0) {
        nu2[r] = s2;
        atomicMax(&keys[3], dlc_f64_key(su));
    }
}

// NumPy's pairwise summation of n doubles (np.add.reduce along a contiguous axis, as np.linalg.norm uses it:
// SimilarityCalculator.py:34) as a postfix program: a leaf (start, n <= 128) is summed with eight strided accumulators
// -- ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)), then the n % 8 last elements one by one; fewer than 8 elements:
// one by one from 0 -- and (-1, 0) adds the two results before it.  A range of more than 128 splits at n / 2 rounded down
// to a multiple of 8.  The pair kernel runs this program where candidates are too close for anything else.
__global__ void sim_pairwise_program_kernel(int H, int2* prog, unsigned long long* len) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int st_start[32], st_n[32], st_phase[32];
    int sp = 0, L = 0;
    st_start[0] = 0; st_n[0] = H; st_phase[0] = 0; sp = 1;
    while (sp > 0) {
        const int start = st_start[sp - 1], n = st_n[sp - 1];
        if (n <= 128) { prog[L++] = make_int2(start, n); --sp; continue; }
        int n2 = n / 2;
        n2 -= n2 % 8;
        if (st_phase[sp - 1] == 0) { st_phase[sp - 1] = 1; st_start[sp] = start; st_n[sp] = n2; st_phase[sp] = 0; ++sp; }
        else if (st_phase[sp - 1] == 1) { st_phase[sp - 1] = 2; st_start[sp] = start + n2; st_n[sp] = n - n2; st_phase[sp] = 0; ++sp; }
        else { prog[L++] = make_int2(-1, 0); --sp; }
    }
    *len = (unsigned long long)L;
}

}  // namespace

size_t sim_filter_panel_bytes(int64_t rows, int64_t H) {
    const size_t kp = dlc::align_up((size_t)H, (size_t)GI_KS);
    // a tile of the column panel starts at any patch: 256 rows of slack behind the last whole tile
    return (dlc::align_up((size_t)rows, (size_t)GI_T) + GI_T) * 3 * kp;
}

// keys[6]: min key, max key, non-finite flag, max row sum key, direct evaluations (a count), length of prog (up to
// 1023 int2 entries for H <= 32768).  X / Y: sim_filter_panel_bytes each (zeroed here).
int sim_filter_prepare(dlc_ctx* ctx, const double* desc, int64_t rows, int64_t H, unsigned long long* keys, char* X, char* Y,
                       double* nu2, void* prog, hipStream_t st) {
    const size_t pb = sim_filter_panel_bytes(rows, H);
    const int kp = (int)dlc::align_up((size_t)H, (size_t)GI_KS);
    hipLaunchKernelGGL(sim_keys_init_kernel, dim3(1), dim3(64), 0, st, keys);
    // only the padding rows need zeros (the kernel writes every word of a real row, padding columns included)
    const size_t real = (size_t)rows * 3 * kp;
    if (pb > real) {
        DLC_HIP_CHECK(ctx, hipMemsetAsync(X + real, 0, pb - real, st));
        DLC_HIP_CHECK(ctx, hipMemsetAsync(Y + real, 0, pb - real, st));
    }
    hipLaunchKernelGGL(sim_pairwise_program_kernel, dim3(1), dim3(64), 0, st, (int)H, (int2*)prog, keys + 5);
    hipLaunchKernelGGL(sim_range_kernel, dim3(2048), dim3(256), 0, st, desc, (long long)(rows * H), keys);
    DLC_LAUNCH_CHECK(ctx, "sim_range_kernel");
    hipLaunchKernelGGL(sim_quant_kernel, dim3((unsigned)dlc::cdiv(rows, 4)), dim3(256), 0, st, desc, (long long)rows, (int)H, kp,
                       keys, X, Y, nu2);
    DLC_LAUNCH_CHECK(ctx, "sim_quant_kernel");
    return DLC_OK;
}

// out[r, c] = acc of (row patch row0 + r, column patch col0 + c), r < mrows, c < ncols (ldo: a multiple of 4 >= ncols);
// tiles without a (row frame < column frame) entry are skipped
int gram_upper_i8(dlc_ctx* ctx, int64_t mrows, int64_t ncols, int64_t H, const char* X, const char* Y, int* out, int64_t ldo,
                  int patches, int64_t row0, int64_t col0, hipStream_t st) {
    const int kp = (int)dlc::align_up((size_t)H, (size_t)GI_KS);
    GramI8Args a;
    a.pitch = 3 * kp; a.kp = kp;
    a.X = X + row0 * (long long)a.pitch;
    a.Y = Y + col0 * (long long)a.pitch;
    a.out = out; a.ldo = ldo; a.mrows = mrows; a.ncols = ncols;
    a.tiles_m = (int)dlc::cdiv(mrows, (int64_t)GI_T); a.tiles_n = (int)dlc::cdiv(ncols, (int64_t)GI_T);
    const int nsm = (a.tiles_m + 3) / 4;
    a.nsn = (a.tiles_n + 7) / 8;
    a.nsup = nsm * a.nsn;
    a.tri_p = patches; a.tri_row0 = row0; a.tri_col0 = col0;
    const size_t lds = (size_t)GI_NSTAGE * GI_STAGE;
    if (!(ctx->func_attr_set & (1ull << DLC_ATTR_GRAM_I8))) {
        DLC_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)gram_i8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ctx->func_attr_set |= 1ull << DLC_ATTR_GRAM_I8;
    }
    const unsigned grid = (unsigned)(((a.nsup + 7) / 8) * 8 * 32);
    hipLaunchKernelGGL(gram_i8_kernel, dim3(grid), dim3(256), lds, st, a);
    DLC_LAUNCH_CHECK(ctx, "gram_i8_kernel");
    return DLC_OK;
}

}  // namespace dlc_gemm
