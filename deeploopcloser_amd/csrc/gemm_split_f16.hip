// SDAV.transform in a TOLERANCE mode on the 16-bit MFMA (dlc_sdav_encode_split; SDAV.py:126-163,293-302;
// TensorflowWrapper.py:57-78: five times sigmoid(h . W + b)).  The fp64 encoder (gemm_dma_f64.hip) stays the parity mode
// and the default; this one trades bit parity for the matrix cores' 16-bit rate under north_star's own tolerance
// ("descriptor L2 within 1e-4"): measured relative L2 <= 2.1e-5 with the reference's N(0,1) weights, 2e-7 with 1/sqrt(fan_in)
// weights (tests/test_gpu_parity.py; the NumPy emulation that chose the form: scripts/emul_split_encoder.py).
//
// Arithmetic.  Every operand is TWO fp16 pieces of a power-of-two multiple of its value,
//   h 2^11 = h1 + h2 + O(2^-22 h 2^11),   W 2^s = W1 + W2 + O(2^-22 W 2^s)   (s: the layer's largest |W| 2^s lands in [2048, 4096)),
// and a layer is THREE fp16 MFMA products into one fp32 accumulator, z 2^(11+s) = h1.W1 + h1.W2 + h2.W1 (the dropped h2.W2 is
// 2^-22 of a term).  fp16 pieces carry 11 significant bits each where bf16 pieces carry 8: the bf16 split needs SIX products
// (three pieces per operand) to come under 1e-4 with N(0,1) weights -- three bf16 products leave 2.7e-4 (same emulation) --
// at the same MFMA rate per product.  The products of two fp16 values are exact in fp32; what is left is the fp32
// accumulation over K = 2500 (1.7e-5 of the descriptor norm through five saturating layers) and the fp32 sigmoid.
//
// Kernel.  The three products are ONE plain GEMM over K' = 3 K: K tile 3 t pairs (h1, W1) at k-tile t, 3 t + 1 (h1, W2),
// 3 t + 2 (h2, W1) -- only the DMA's base pointers know; a piece's k-tile is read again one or two K tiles after its first
// use, out of L2 (segment by segment -- all of (h1, W1), then all of (h1, W2), ... -- every re-read went back over the
// fabric: 4 % slower, at 1.5e-5 instead of 2.1e-5 of error: the small products then reach the accumulator last).
// Main loop = the cosine match's score GEMM (cosine_topk.hip: 256 x 256 tile, BK = 64, 8 waves of 128 x 64, LDS-DMA rings, snake order of eight mini-phases per K
// tile), with the roles turned: MFMA A operand = 256 weight columns (fragment rows permuted so that a lane's sixteen
// accumulators of a half are sixteen CONSECUTIVE output columns n), B operand = 256 activation rows (a lane holds ONE
// row m).  The epilogue therefore writes, per lane, runs of sixteen consecutive outputs of one row: bias + sigmoid in fp32,
// then either the next layer's two fp16 pieces (32 bytes each) or, for the last layer, sixteen fp64 values.  The
// activations never exist as fp64 between the layers.
// Both operands are re-read (neither is a once-only stream): an XCD's 32 resident workgroups take one br x bc block of
// tiles, so that br + bc operand panels feed br * bc tiles out of that XCD's L2.
#include "gemm_internal.h"

namespace dlc_gemm {
namespace {

constexpr int SP_BM = 256;                     // weight columns per tile (MFMA A operand)
constexpr int SP_BN = 256;                     // activation rows per tile (MFMA B operand)
constexpr int SP_THREADS = 512;
constexpr int SP_TILE = 256 * 64 * 2;          // one operand's K tile: 256 rows x 128 B = 32 KiB
constexpr int SP_HALF = 128 * 128;             // a half tile of 128 rows
constexpr int SP_STAGES = 2;
constexpr int SP_B_RING = SP_STAGES * SP_TILE;
constexpr int SP_LDS = 2 * SP_STAGES * SP_TILE;            // 128 KiB
constexpr int SP_X_SHIFT = 11;                 // activations are carried as h * 2^11

typedef __attribute__((address_space(3))) void* lptr_t;

struct SplitArgs {
    const char* W[2];           // weight pieces, TRANSPOSED: [Np, ldw] fp16, row n = column n of W (rows >= N, k >= K: zeros)
    const char* X[2];           // activation pieces [M, ldx] fp16
    long long ldw_b, ldx_b;     // row strides in bytes
    long long M;                // activation rows
    int N;                      // output columns
    int nk;                     // K tiles of 64 per piece
    const double* bias;         // [N] or null
    const float* wscale;        // device: 2^s of this layer's weights
    char* O[2];                 // next layer's pieces [M, ldo] fp16 (null for the last layer) ...
    long long ldo_b;
    double* C;                  // ... whose output is fp64 [M, ldc]
    long long ldc;
    int br, bc;                 // an XCD's block: br tiles of activation rows x bc tiles of weight columns, br * bc <= 32
    long long nbr, nblocks;
    long long tiles_m;          // activation row tiles
    int tiles_n;                // weight column tiles
};

__device__ __forceinline__ int swz_a(int r) { return ((r >> 1) & 1) | (((r >> 4) & 3) << 1); }
__device__ __forceinline__ int swz_b(int r) { return (r >> 1) & 7; }

// Four LDS-DMA wave-instructions (4 x 1 KiB: 32 rows of one half tile); inline asm so that hipcc does not count them in
// vmcnt, M0 saved and restored (cosine_topk.hip: dma4).
__device__ __forceinline__ void sp_dma4(const unsigned (&voff)[4], const char* sbase, unsigned lds0) {
    unsigned keep;
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %6\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %5\n\t"
        "s_add_u32 m0, %6, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %5\n\t"
        "s_add_u32 m0, %6, 0x800\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, %5\n\t"
        "s_add_u32 m0, %6, 0xc00\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %4, %5\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(sbase), "s"(lds0)
        : "memory", "scc");
}
__device__ __forceinline__ const char* sp_uniform_ptr(const char* p) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}

#define SP_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define SP_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
__device__ __forceinline__ void sp_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <bool FINAL>
__global__ __launch_bounds__(SP_THREADS, 2) void gemm_split_f16_kernel(SplitArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_sp[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2;   // weight-column half (128 columns)
    const int wc = wid & 3;    // activation-row block (64 rows)
    // workgroup -> tile: ids go round-robin to the 8 XCDs; 32 consecutive ids of an XCD take one br x bc block
    long long tile_m;
    int tile_n;
    {
        const long long id = blockIdx.x;
        const long long l = id >> 3;
        const long long blk = (l >> 5) * 8 + (id & 7);
        const int i = (int)(l & 31);
        if (blk >= p.nblocks || i >= p.br * p.bc) return;
        tile_m = (blk % p.nbr) * p.br + (i % p.br);
        tile_n = (int)(blk / p.nbr) * p.bc + (i / p.br);
        if (tile_m >= p.tiles_m || tile_n >= p.tiles_n) return;
    }
    const unsigned lds_base = (unsigned)(unsigned long long)(lptr_t)smem_sp;

    // ---- DMA roles: waves 0-3 stage the weight (A) halves, waves 4-7 the activation (B) halves
    const bool is_a = wid < 4;
    const int ridx = wid & 3;                               // this wave stages rows 32*ridx .. +31 of a half
    unsigned voff[2][4];                                    // [half][dma]: byte offset of this lane's 16 B
    {
        const int slot = lane & 7;
        const long long brows = p.M - tile_m * SP_BN;       // valid activation rows in this tile (>= 1)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 32 * ridx + 8 * j + (lane >> 3);  // row inside the half
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const long long ar = (r >> 6) * 128 + h * 64 + (r & 63);         // (the weight panels hold whole tiles)
                long long br = (r >> 5) * 64 + h * 32 + (r & 31);
                if (br > brows - 1) br = brows - 1;
                voff[h][j] = is_a ? (unsigned)(ar * p.ldw_b + ((slot ^ swz_a(r)) << 4))
                                  : (unsigned)(br * p.ldx_b + ((slot ^ swz_b(r)) << 4));
            }
        }
    }
    // this wave's two piece bases, and the segment of a k-tile's three products (0: h1 W1, 1: h1 W2, 2: h2 W1) that takes its second piece
    const char* base0 = sp_uniform_ptr(is_a ? p.W[0] + (long long)tile_n * SP_BM * p.ldw_b : p.X[0] + tile_m * SP_BN * p.ldx_b);
    const char* base1 = sp_uniform_ptr(is_a ? p.W[1] + (long long)tile_n * SP_BM * p.ldw_b : p.X[1] + tile_m * SP_BN * p.ldx_b);
    const int seg1 = is_a ? 1 : 2;
    const unsigned lds_stage = lds_base + (unsigned)(32 * ridx) * 128 + (is_a ? 0u : (unsigned)SP_B_RING);
    const int nk3 = 3 * p.nk;

    // ---- fragment read offsets (bytes inside a half)
    const int i = lane & 15;
    const int kq = lane >> 4;
    const int fa = ((i >> 1) & 1) | ((i >> 2) << 1);
    const int fb = (i >> 1) & 7;
    typedef const __attribute__((address_space(3))) u32x4_t* lds_u4p;
    typedef const __attribute__((address_space(3))) char* lds_cp;
    const lds_cp lbase = (lds_cp)(lptr_t)smem_sp;
    const unsigned rdA0_l = (wr * 64 + 16 * (i >> 2) + (i & 3)) * 128 + (((0 + kq) ^ fa) << 4);   // + tt*512
    const unsigned rdA1_l = (wr * 64 + 16 * (i >> 2) + (i & 3)) * 128 + (((4 + kq) ^ fa) << 4);
    const unsigned rdB0_l = SP_B_RING + (wc * 32 + i) * 128 + (((0 + kq) ^ fb) << 4);             // + c*2048
    const unsigned rdB1_l = SP_B_RING + (wc * 32 + i) * 128 + (((4 + kq) ^ fb) << 4);
    unsigned aoff = 0, boff = 0;
    unsigned rdA0 = rdA0_l, rdA1 = rdA1_l, rdB0 = rdB0_l, rdB1 = rdB1_l;

    f32x4_t acc[8][4];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[t][c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    u32x4_t faX[4], faY[4], fbX[2], fbY[2];

#define SP_READ_A(DST, RD, OFF) _Pragma("unroll") for (int tt = 0; tt < 4; ++tt) DST[tt] = *(lds_u4p)(lbase + (RD) + (OFF) + tt * 512)
#define SP_READ_B(DST, RD, OFF) _Pragma("unroll") for (int c = 0; c < 2; ++c) DST[c] = *(lds_u4p)(lbase + (RD) + (OFF) + c * 2048)
#define SP_MFMA(FA, FB, AH, BH)                                                                                  \
    do {                                                                                                         \
        __builtin_amdgcn_s_setprio(1);                                                                           \
        _Pragma("unroll") for (int tt = 0; tt < 4; ++tt) _Pragma("unroll") for (int c = 0; c < 2; ++c)           \
            acc[(AH) * 4 + tt][(BH) * 2 + c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(                           \
                __builtin_bit_cast(f16x8_t, FA[tt]), __builtin_bit_cast(f16x8_t, FB[c]), acc[(AH) * 4 + tt][(BH) * 2 + c], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                           \
    } while (0)
#define SP_RELEASE() SP_WAIT_LGKM0(); sp_barrier()
    // DMA of half H of K tile t2 into ring position POS (past the end: clamped -- the redundant DMA lands in a dead half
    // and keeps the vmcnt bookkeeping uniform).  The K tile's segment picks the piece.
#define SP_ISSUE(POS, H, t2)                                                                                     \
    do {                                                                                                         \
        const int kk_ = (t2) < nk3 ? (t2) : nk3 - 1;                                                             \
        const int kt_ = kk_ / 3, seg_ = kk_ - 3 * kt_;                                                           \
        const char* src_ = (seg_ == seg1 ? base1 : base0) + (long long)kt_ * 128;                                \
        sp_dma4(voff[H], src_, lds_stage + (POS) + (H) * SP_HALF);                                               \
    } while (0)
#define SP_ISSUE_A(POS, H, t2) do { if (is_a) SP_ISSUE(POS, H, t2); } while (0)
#define SP_ISSUE_B(POS, H, t2) do { if (!is_a) SP_ISSUE(POS, H, t2); } while (0)

    // ---- prologue: K tiles 0 and 1 of both operands issued (per-tile order A1, A0 / B0, B1, as the steady state issues them)
#pragma unroll
    for (int s_ = 0; s_ < SP_STAGES; ++s_) {
        SP_ISSUE_A(s_ * SP_TILE, 1, s_);
        SP_ISSUE_A(s_ * SP_TILE, 0, s_);
    }
    SP_ISSUE_B(0 * SP_TILE, 0, 0);
    SP_ISSUE_B(0 * SP_TILE, 1, 0);
    SP_ISSUE_B(1 * SP_TILE, 0, 1);
    SP_ISSUE_B(1 * SP_TILE, 1, 1);
    SP_WAIT_VMCNT(8);                                          // K tile 0 landed (this wave's share)
    sp_barrier();
    SP_READ_A(faX, rdA0, 0);
    SP_READ_B(fbX, rdB0, 0);

    // One K tile = 8 mini-phases of 8 MFMAs (cosine_topk.hip: the same schedule):
    //   m1 (A0,B0,k0) m2 (A0,B1,k0) m3 (A1,B1,k0) m4 (A1,B0,k0)  m5 (A1,B0,k1) m6 (A1,B1,k1) m7 (A0,B1,k1) m8 (A0,B0,k1)
    // A half is dead once its k1 slice has been read (A1 after m3, B0 after m4, B1 after m5, A0 after m6): a barrier
    // there, then the DMA that refills it with K tile t + 2.
    for (int t = 0; t < nk3; ++t) {
        SP_READ_B(fbY, rdB0, SP_HALF);
        SP_MFMA(faX, fbX, 0, 0);                                   // m1
        SP_READ_A(faY, rdA0, SP_HALF);
        SP_MFMA(faX, fbY, 0, 1);                                   // m2
        SP_READ_A(faX, rdA1, SP_HALF);
        SP_MFMA(faY, fbY, 1, 1);                                   // m3
        SP_RELEASE();                                              // A1 read by everyone
        SP_READ_B(fbY, rdB1, 0);
        SP_ISSUE_A(aoff, 1, t + SP_STAGES);
        SP_MFMA(faY, fbX, 1, 0);                                   // m4
        SP_RELEASE();                                              // B0
        SP_READ_B(fbX, rdB1, SP_HALF);
        SP_ISSUE_B(boff, 0, t + 2);
        SP_MFMA(faX, fbY, 1, 0);                                   // m5
        SP_RELEASE();                                              // B1
        SP_READ_A(faY, rdA1, 0);
        SP_ISSUE_B(boff, 1, t + 2);
        SP_MFMA(faX, fbX, 1, 1);                                   // m6
        if (is_a) SP_WAIT_VMCNT(4); else SP_WAIT_VMCNT(8);         // K tile t+1 landed (this wave's share)
        SP_RELEASE();                                              // A0; and t+1 visible to all
        SP_ISSUE_A(aoff, 0, t + SP_STAGES);
        aoff ^= SP_TILE;
        boff ^= SP_TILE;
        rdA0 = rdA0_l + aoff; rdA1 = rdA1_l + aoff; rdB0 = rdB0_l + boff; rdB1 = rdB1_l + boff;
        SP_READ_A(faX, rdA0, 0);
        SP_MFMA(faY, fbX, 0, 1);                                   // m7
        SP_READ_B(fbX, rdB0, 0);
        SP_MFMA(faY, fbY, 0, 0);                                   // m8
    }
    SP_WAIT_VMCNT(0);    // the clamped tail DMAs must not outlive the workgroup's LDS
    SP_WAIT_LGKM0();
#undef SP_ISSUE_A
#undef SP_ISSUE_B
#undef SP_ISSUE
#undef SP_READ_A
#undef SP_READ_B
#undef SP_MFMA
#undef SP_RELEASE

    // ---- epilogue.  C/D layout: column = lane & 15 -> activation row, row = 4 * (lane >> 4) + reg -> weight column; with
    // the permuted A rows, acc[th * 4 + tt][c][r] is output (m, n0 + 4 tt + r): sixteen consecutive columns per (c, th).
    const int lg = lane >> 4;
    const float inv = 1.0f / ((float)(1 << SP_X_SHIFT) * p.wscale[0]);        // a power of two: exact
    const float xs = (float)(1 << SP_X_SHIFT);
#pragma unroll
    for (int th = 0; th < 2; ++th) {
        const int n0 = tile_n * SP_BM + wr * 128 + th * 64 + 16 * lg;
        float bv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) bv[j] = (p.bias && n0 + j < p.N) ? (float)p.bias[n0 + j] : 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const long long m = tile_m * SP_BN + wc * 64 + c * 16 + i;
            if (m >= p.M) continue;
            float hv[16];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float z = fmaf(acc[th * 4 + tt][c][r], inv, bv[4 * tt + r]);
                    hv[4 * tt + r] = __frcp_rn(1.0f + __expf(-z));                  // sigmoid (TensorflowWrapper.py:77-78)
                }
            if constexpr (FINAL) {
                double* dst = p.C + m * p.ldc + n0;
                if (n0 + 16 <= p.N && ((p.ldc & 1) == 0)) {
#pragma unroll
                    for (int j = 0; j < 16; j += 2) *(double2*)(dst + j) = make_double2((double)hv[j], (double)hv[j + 1]);
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j)
                        if (n0 + j < p.N) dst[j] = (double)hv[j];
                }
            } else {
                // the next layer's two pieces of h * 2^11; columns past N are zeros there (its weights' k rows past N too)
                unsigned w1[8], w2[8];
#pragma unroll
                for (int j = 0; j < 16; j += 2) {
                    const float v0 = n0 + j < p.N ? hv[j] * xs : 0.0f, v1 = n0 + j + 1 < p.N ? hv[j + 1] * xs : 0.0f;
                    const _Float16 a0 = (_Float16)v0, a1 = (_Float16)v1;
                    const _Float16 b0 = (_Float16)(v0 - (float)a0), b1 = (_Float16)(v1 - (float)a1);
                    w1[j >> 1] = (unsigned)__builtin_bit_cast(unsigned short, a0) | ((unsigned)__builtin_bit_cast(unsigned short, a1) << 16);
                    w2[j >> 1] = (unsigned)__builtin_bit_cast(unsigned short, b0) | ((unsigned)__builtin_bit_cast(unsigned short, b1) << 16);
                }
                char* o1 = p.O[0] + m * p.ldo_b + (long long)n0 * 2;
                char* o2 = p.O[1] + m * p.ldo_b + (long long)n0 * 2;
                *(uint4*)o1 = make_uint4(w1[0], w1[1], w1[2], w1[3]);
                *(uint4*)(o1 + 16) = make_uint4(w1[4], w1[5], w1[6], w1[7]);
                *(uint4*)o2 = make_uint4(w2[0], w2[1], w2[2], w2[3]);
                *(uint4*)(o2 + 16) = make_uint4(w2[4], w2[5], w2[6], w2[7]);
            }
        }
    }
}

// ---- operand preparation --------------------------------------------------------------------------------------------

// largest |w| of a weight matrix -> bits of a non-negative double (they order like the values)
__global__ __launch_bounds__(256) void sp_absmax_kernel(const double* __restrict__ w, long long n, unsigned long long* out) {
    double m = 0.0;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) m = fmax(m, fabs(w[e]));
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(m));     // (a NaN never wins: fmax drops it)
}
// 2^s with the layer's largest |W| 2^s in [2048, 4096) (1 for a zero matrix), as the float the GEMM's epilogue divides by
__global__ void sp_scale_kernel(const unsigned long long* amax, float* wscale) {
    if (threadIdx.x != 0) return;
    const double m = __longlong_as_double((long long)*amax);
    int e = 0;
    if (m > 0.0 && m < INFINITY) (void)frexp(m, &e);                   // m = f 2^e, f in [0.5, 1)
    int s = (m > 0.0 && m < INFINITY) ? 12 - e : 0;                     // m 2^s = f 2^12 in [2048, 4096)
    s = s > 100 ? 100 : (s < -100 ? -100 : s);
    *wscale = (float)ldexp(1.0, s);
}
// W [K, N] fp64 -> its two fp16 pieces transposed, [Np, ldw] each (k contiguous), zeros past K / N: 64 x 64 tiles through LDS
__global__ __launch_bounds__(256) void sp_split_weights_kernel(const double* __restrict__ w, long long K, long long N, const float* wscale,
                                                               unsigned short* __restrict__ p1, unsigned short* __restrict__ p2,
                                                               long long ldw) {
    __shared__ float t1[64][65], t2[64][65];
    const long long n0 = (long long)blockIdx.x * 64, k0 = (long long)blockIdx.y * 64;
    const double sc = (double)wscale[0];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int kk = ty; kk < 64; kk += 4) {
        const long long k = k0 + kk, n = n0 + tx;
        const double v = (k < K && n < N) ? w[k * N + n] * sc : 0.0;
        const _Float16 a = (_Float16)v;
        const _Float16 b = (_Float16)(v - (double)a);
        t1[kk][tx] = (float)a;
        t2[kk][tx] = (float)b;
    }
    __syncthreads();
    for (int nn = ty; nn < 64; nn += 4) {
        const long long o = (n0 + nn) * ldw + k0 + tx;
        p1[o] = __builtin_bit_cast(unsigned short, (_Float16)t1[tx][nn]);
        p2[o] = __builtin_bit_cast(unsigned short, (_Float16)t2[tx][nn]);
    }
}
// x [rows, K] fp64 -> the two fp16 pieces of x 2^11, [rows, ldx] each, zeros past K
__global__ __launch_bounds__(256) void sp_split_rows_kernel(const double* __restrict__ x, long long rows, long long K,
                                                            unsigned short* __restrict__ p1, unsigned short* __restrict__ p2,
                                                            long long ldx) {
    const long long total = rows * ldx;
    const double sc = (double)(1 << SP_X_SHIFT);
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long r = e / ldx, k = e - r * ldx;
        const double v = k < K ? x[r * K + k] * sc : 0.0;
        const _Float16 a = (_Float16)v;
        p1[e] = __builtin_bit_cast(unsigned short, a);
        p2[e] = __builtin_bit_cast(unsigned short, (_Float16)(v - (double)a));
    }
}

struct PanelLayout {
    size_t scale, amax, p1, p2, total;          // byte offsets inside one layer's panel block
    long long np, kp;
};
PanelLayout panel_layout(int64_t K, int64_t N) {
    PanelLayout L;
    L.np = (long long)dlc::align_up((size_t)N, (size_t)SP_BM);
    L.kp = (long long)dlc::align_up((size_t)K, 64);
    L.scale = 0; L.amax = 8;
    L.p1 = 256;
    L.p2 = L.p1 + dlc::align_up((size_t)L.np * L.kp * 2, 256);
    L.total = L.p2 + dlc::align_up((size_t)L.np * L.kp * 2, 256);
    return L;
}

}  // namespace

size_t split_panels_bytes(int n_layers, const int64_t* dims) {
    size_t t = 0;
    for (int l = 0; l < n_layers; ++l) t += panel_layout(dims[l], dims[l + 1]).total;
    return t;
}

int split_prepare(dlc_ctx* ctx, int n_layers, const int64_t* dims, const double* const* W, char* panels, hipStream_t st) {
    size_t off = 0;
    for (int l = 0; l < n_layers; ++l) {
        const PanelLayout L = panel_layout(dims[l], dims[l + 1]);
        char* base = panels + off;
        DLC_HIP_CHECK(ctx, hipMemsetAsync(base, 0, 256, st));
        const long long n = dims[l] * dims[l + 1];
        hipLaunchKernelGGL(sp_absmax_kernel, dim3(1024), dim3(256), 0, st, W[l], n, (unsigned long long*)(base + L.amax));
        hipLaunchKernelGGL(sp_scale_kernel, dim3(1), dim3(64), 0, st, (const unsigned long long*)(base + L.amax), (float*)(base + L.scale));
        hipLaunchKernelGGL(sp_split_weights_kernel, dim3((unsigned)(L.np / 64), (unsigned)(L.kp / 64)), dim3(256), 0, st, W[l],
                           (long long)dims[l], (long long)dims[l + 1], (const float*)(base + L.scale), (unsigned short*)(base + L.p1),
                           (unsigned short*)(base + L.p2), L.kp);
        DLC_LAUNCH_CHECK(ctx, "sp_split_weights_kernel");
        off += L.total;
    }
    return DLC_OK;
}

// the activation pieces' row pitch (elements) for a layer input of width K: whole output tiles of the layer before
static long long split_pitch(int64_t K) { return (long long)dlc::align_up((size_t)K, (size_t)SP_BM); }

size_t split_encode_workspace_bytes(int64_t rows, const int64_t* dims, int n_layers) {
    long long wmax = 0;
    for (int l = 0; l < n_layers; ++l) wmax = split_pitch(dims[l]) > wmax ? split_pitch(dims[l]) : wmax;
    return 4 * dlc::align_up((size_t)rows * (size_t)wmax * 2, 256);          // two ping-pong buffers of two pieces
}

int split_encode(dlc_ctx* ctx, int64_t rows, int n_layers, const int64_t* dims, const double* x, const char* panels,
                 const double* const* b, double* out, char* ws, hipStream_t st) {
    long long wmax = 0;
    for (int l = 0; l < n_layers; ++l) wmax = split_pitch(dims[l]) > wmax ? split_pitch(dims[l]) : wmax;
    const size_t piece = dlc::align_up((size_t)rows * (size_t)wmax * 2, 256);
    char* buf[2][2] = {{ws, ws + piece}, {ws + 2 * piece, ws + 3 * piece}};
    {
        const long long ldx = split_pitch(dims[0]);
        long long blocks = dlc::cdiv(rows * ldx, (int64_t)256);
        if (blocks > 256 * 32) blocks = 256 * 32;
        hipLaunchKernelGGL(sp_split_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, (long long)rows, (long long)dims[0],
                           (unsigned short*)buf[0][0], (unsigned short*)buf[0][1], ldx);
        DLC_LAUNCH_CHECK(ctx, "sp_split_rows_kernel");
    }
    if (!(ctx->func_attr_set & (1ull << DLC_ATTR_SPLIT_F16))) {
        DLC_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)gemm_split_f16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, SP_LDS));
        DLC_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)gemm_split_f16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SP_LDS));
        ctx->func_attr_set |= 1ull << DLC_ATTR_SPLIT_F16;
    }
    size_t off = 0;
    for (int l = 0; l < n_layers; ++l) {
        const PanelLayout L = panel_layout(dims[l], dims[l + 1]);
        const char* base = panels + off;
        SplitArgs a;
        a.W[0] = base + L.p1; a.W[1] = base + L.p2; a.ldw_b = L.kp * 2;
        a.X[0] = buf[l & 1][0]; a.X[1] = buf[l & 1][1]; a.ldx_b = split_pitch(dims[l]) * 2;
        a.M = rows; a.N = (int)dims[l + 1]; a.nk = (int)(L.kp / 64);
        a.bias = b ? b[l] : nullptr;
        a.wscale = (const float*)(base + L.scale);
        const bool fin = l == n_layers - 1;
        a.O[0] = fin ? nullptr : buf[(l + 1) & 1][0]; a.O[1] = fin ? nullptr : buf[(l + 1) & 1][1];
        a.ldo_b = fin ? 0 : split_pitch(dims[l + 1]) * 2;
        a.C = fin ? out : nullptr; a.ldc = dims[l + 1];
        a.tiles_m = dlc::cdiv(rows, (int64_t)SP_BN);
        a.tiles_n = (int)(L.np / SP_BM);
        // an XCD's block of <= 32 tiles: the shape with the fewest operand panels per computed tile, idle slots counted
        int best_br = 1, best_bc = 1;
        double best = 1e30;
        for (int bc = 1; bc <= 32 && bc <= a.tiles_n; ++bc) {
            int br = 32 / bc;
            if ((long long)br > a.tiles_m) br = (int)a.tiles_m;
            const double cover = (double)(dlc::cdiv((int64_t)a.tiles_n, (int64_t)bc) * bc) / a.tiles_n *
                                 (double)(dlc::cdiv(a.tiles_m, (int64_t)br) * br) / (double)a.tiles_m;
            const double cost = (double)(br + bc) / (br * bc) * cover * (32.0 / (br * bc));
            if (cost < best) { best = cost; best_br = br; best_bc = bc; }
        }
        a.br = best_br; a.bc = best_bc;
        a.nbr = dlc::cdiv(a.tiles_m, (int64_t)a.br);
        a.nblocks = a.nbr * dlc::cdiv((int64_t)a.tiles_n, (int64_t)a.bc);
        const long long nwg = dlc::cdiv(a.nblocks, (int64_t)8) * 8 * 32;
        if (nwg > 0x7fffffffll) return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "sdav_encode_split: too many tiles");
        const int prof_slot = (int)(ctx->prof_calls % DLC_PROFILE_RING);
        if (ctx->profiling) DLC_HIP_CHECK(ctx, hipEventRecord(ctx->ev_start[prof_slot], st));
        if (fin) hipLaunchKernelGGL(gemm_split_f16_kernel<true>, dim3((unsigned)nwg), dim3(SP_THREADS), SP_LDS, st, a);
        else hipLaunchKernelGGL(gemm_split_f16_kernel<false>, dim3((unsigned)nwg), dim3(SP_THREADS), SP_LDS, st, a);
        DLC_LAUNCH_CHECK(ctx, "gemm_split_f16_kernel");
        if (ctx->profiling) {
            DLC_HIP_CHECK(ctx, hipEventRecord(ctx->ev_stop[prof_slot], st));
            ctx->prof_calls++;
        }
        off += L.total;
    }
    return DLC_OK;
}

}  // namespace dlc_gemm

extern "C" size_t dlc_sdav_split_panels_bytes(int n_layers, const int64_t* dims) {
    if (!dims || n_layers < 1) return 0;
    for (int l = 0; l <= n_layers; ++l)
        if (dims[l] < 1) return 0;
    return dlc_gemm::split_panels_bytes(n_layers, dims);
}

extern "C" int dlc_sdav_split_prepare(dlc_ctx* ctx, int n_layers, const int64_t* dims, const double* const* W, void* panels,
                                      size_t panels_bytes, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!dims || !W || !panels || n_layers < 1) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "sdav_split_prepare: null/empty argument");
    const size_t need = dlc_sdav_split_panels_bytes(n_layers, dims);
    if (need == 0) return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "sdav_split_prepare: bad dims");
    if (panels_bytes < need) return dlc::fail(ctx, DLC_ERR_WORKSPACE, "sdav_split_prepare: panels %zu < %zu bytes", panels_bytes, need);
    if ((uintptr_t)panels & 255) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "sdav_split_prepare: panels must be 256-byte aligned");
    for (int l = 0; l < n_layers; ++l)
        if (!W[l]) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "sdav_split_prepare: W[%d] is null", l);
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    return dlc_gemm::split_prepare(ctx, n_layers, dims, W, (char*)panels, (hipStream_t)stream);
}

extern "C" size_t dlc_sdav_encode_split_workspace_bytes(int64_t rows, const int64_t* dims, int n_layers) {
    if (rows < 1 || !dims || n_layers < 1) return 0;
    return dlc_gemm::split_encode_workspace_bytes(rows, dims, n_layers);
}

extern "C" int dlc_sdav_encode_split(dlc_ctx* ctx, int64_t rows, int n_layers, const int64_t* dims, const double* x,
                                     const void* panels, const double* const* b, double* out, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!dims || !x || !panels || !out || rows < 1 || n_layers < 1)
        return dlc::fail(ctx, DLC_ERR_BAD_ARG, "sdav_encode_split: null/empty argument");
    for (int l = 0; l <= n_layers; ++l)
        if (dims[l] < 1) return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "sdav_encode_split: dims[%d] = %lld", l, (long long)dims[l]);
    if (rows * 4096 > 0x7fffffffll * 64) return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "sdav_encode_split: too many rows");
    const size_t need = dlc_sdav_encode_split_workspace_bytes(rows, dims, n_layers);
    if (!workspace || workspace_bytes < need)
        return dlc::fail(ctx, DLC_ERR_WORKSPACE, "sdav_encode_split: workspace %zu < %zu bytes", workspace_bytes, need);
    if (((uintptr_t)workspace & 255) || ((uintptr_t)panels & 255))
        return dlc::fail(ctx, DLC_ERR_BAD_ARG, "sdav_encode_split: workspace and panels must be 256-byte aligned");
    // per-lane DMA offsets are 32-bit: 256 rows of the widest operand
    for (int l = 0; l < n_layers; ++l)
        if ((long long)dlc::align_up((size_t)dims[l], 256) * 2 * 256 > 0x7fffffffll)
            return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "sdav_encode_split: dims[%d] too wide", l);
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    return dlc_gemm::split_encode(ctx, rows, n_layers, dims, x, (const char*)panels, b, out, (char*)workspace, (hipStream_t)stream);
}
