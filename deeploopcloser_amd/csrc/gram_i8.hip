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
// inside that window directly in fp64 from the descriptors, near-ties in NumPy's own summation order (match_ref.hip).
// Six int8 products of K = H replace one fp64 product: a sixth of its time as measured (6 ms against 36.5 at 1063
// frames), and the result is the arg-min of the true distances either way.
// Test infrastructure never enters: the oracle (oracle/similarity.py) only checks the outcome in tests/.
//
// Layout: X = (q1 | q2 | q3), Y = (q3 | q2 | q1) along K (Kp = H rounded up to 256, zero padded); then class 4 is
// X[:, 0:3Kp] . Y[:, 0:3Kp]^T, class 3 is X[:, 0:2Kp] . Y[:, Kp:3Kp]^T, class 2 is X[:, 0:Kp] . Y[:, 2Kp:3Kp]^T: one NT
// int8 GEMM over three K segments with the accumulators shifted right by 7 between them.  In memory both are tiled the
// way the MFMA reads them: [16-row group][k-step of 64 bytes][lane l: row l % 16, bytes (l / 16) * 16 .. + 15] -- every
// LDS-DMA piece is then 1 KiB of consecutive bytes, eight whole cache lines.  (Row-major slices made each piece 16
// half lines; every line crossed the L2 -> L1 path twice, once per k-step, and the kernel sat at 12 B / clock / CU.)
#include "gemm_internal.h"

namespace dlc_gemm {
namespace {

typedef __attribute__((ext_vector_type(4))) int v4i;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int GI_T = 256;                 // tile: 256 row patches x 256 column patches, 8 waves of 128 x 64
constexpr int GI_KS = 64;                 // bytes of K per LDS stage = one MFMA k-step
constexpr int GI_KPAD = 256;              // the slices' padded length is a multiple of this: every segment an even number of stages, at least 4
constexpr int GI_HALF = GI_T * GI_KS;     // one operand's part of a stage: 16 KiB
constexpr int GI_STAGE = 2 * GI_HALF;
constexpr int GI_NSTAGE = 4;              // 128 KiB: three stages in flight behind the one being read

struct GramI8Args {
    const char* X;                        // row panel: the group of row patch row0 (a multiple of 16)
    const char* Y;                        // column panel: the group of column patch col0 (a multiple of 16)
    int* out;                             // [mrows, ldo] accumulators (units of 2^-14 in u . u)
    long long ldo, mrows, ncols;
    long long gpitch;                     // bytes of one 16-row group: 3 Kp / 64 k-steps of 1 KiB
    int kp;
    int tiles_m, tiles_n, nsm, nsn, nsup;
    int tri_p;
    long long tri_row0, tri_col0;
};

// two 1 KiB LDS-DMA pieces (two 16-row groups: wave-uniform bases b0, b1; lane l fetches bytes l * 16 .. + 15).  Inline
// asm so that hipcc does not count them in vmcnt; M0 carries the wave-uniform LDS destination and is saved / restored
// because the compiler owns it.
__device__ __forceinline__ void dma2(unsigned voff, const char* b0, const char* b1, unsigned lds) {
    unsigned keep;
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %4\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_add_u32 m0, %4, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %3\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(b0), "s"(b1), "s"(lds)
        : "memory", "scc");
}

// ... and one piece, the one the K loop issues.  Here M0 is declared clobbered instead of saved and restored, and the
// leading s_nop is gone (base and destination are scalar-ALU results): 6.25 -> 6.11 ms.  hipcc warns that M0 is a
// reserved register it may not preserve across the statement -- nothing else in this kernel reads or writes M0 (checked
// in the ISA: every M0 access lies inside these asm statements), which is what makes it safe HERE and only here.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma1(unsigned voff, const char* b0, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(b0), "s"(lds) : "memory", "m0");
}
#pragma clang diagnostic pop

__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}

// LDS stage: the row panel's 16 groups of 16 rows, a 1 KiB block each in MFMA operand order -- lane l of a block holds
// row l % 16, bytes (l / 16) * 16 .. + 15 of the k-step -- then the column panel's the same.  A fragment read is one
// ds_read_b128 at block + lane * 16.
//
// Pipeline: four stages; iteration t multiplies the fragments of stage t, which it read from LDS during iteration t-1,
// while it reads those of stage t+1 -- one wave per SIMD (256 accumulators per lane), so nothing else hides the LDS
// latency.  Barrier t therefore says "stage t+1 has landed everywhere and everybody is through reading stage t", and
// behind it stage t+4 goes into stage t's slot: three k-steps for a piece to arrive.
__global__ __launch_bounds__(512) void gram_i8_kernel(const GramI8Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem_i8[];
    // workgroup -> tile: ids go round-robin to the 8 XCDs; an XCD's 32 resident workgroups take one 4 x 8 block of
    // tiles (12 panels feed 32 tiles out of that XCD's L2).  Only blocks with a wanted tile are numbered, row by row, and
    // dealt to the XCDs in turn: dealt by block column, the triangle gave XCD 7 2.4 times the work of XCD 0 (a third of
    // the chip's wave time idle, profiles/r02i).
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3, local = slot & 31;
    int want = (slot >> 5) * 8 + xcd;                 // index among the wanted blocks
    int si = 0, sj = 0;
    bool found = false;
    for (; si < p.nsm; ++si) {
        // first block column of block row si with a tile that holds a (row frame < column frame) entry
        const long long row_frame = (p.tri_row0 + (long long)si * 4 * GI_T) / p.tri_p;
        long long c_need = (row_frame + 1) * p.tri_p - p.tri_col0;      // first wanted column patch (relative)
        if (c_need < 0) c_need = 0;
        const int sj_min = (int)(c_need / (8 * GI_T));
        const int cnt = p.nsn - sj_min;
        if (cnt <= 0) continue;
        if (want < cnt) { sj = sj_min + want; found = true; break; }
        want -= cnt;
    }
    if (!found) return;
    const int tile_m = si * 4 + (local >> 3), tile_n = sj * 8 + (local & 7);
    if (tile_m >= p.tiles_m || tile_n >= p.tiles_n) return;
    const long long m0 = (long long)tile_m * GI_T, n0 = (long long)tile_n * GI_T;
    if ((p.tri_col0 + n0 + GI_T - 1) / p.tri_p <= (p.tri_row0 + m0) / p.tri_p) return;   // no (row frame < column frame) entry

    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wr = w >> 2, wc = w & 3;              // 128 row patches x 64 column patches per wave
    const unsigned lds_base = (unsigned)(unsigned long long)(lptr_t)smem_i8;
    const unsigned voff = lane * 16;
    // this wave's two groups of each panel (groups 2 w, 2 w + 1 of the tile's 16)
    const char* xb = uniform_ptr(p.X + (m0 / 16 + w * 2) * p.gpitch);
    const char* yb = uniform_ptr(p.Y + (n0 / 16 + w * 2) * p.gpitch);
    const int n64 = p.kp / GI_KS;
    const int nst = 6 * n64;

    // The next stage to fetch and where it lies, kept as running scalars: worked out from the stage number (two
    // compares, selects, a multiply, a readfirstlane) the address arithmetic of an issue was ~30 scalar and 2 vector
    // instructions in front of its four DMA instructions, in the gap of every iteration.
    const char* xb1 = xb + p.gpitch;
    const char* yb1 = yb + p.gpitch;
    const unsigned lds_w = __builtin_amdgcn_readfirstlane(lds_base + w * 2048);
    int is_t = 0, is_x = 0, is_y = 0, is_wrap = 3 * n64, is_ybase = n64 * 1024;
    auto issue = [&]() {
        const unsigned dst = lds_w + (is_t & (GI_NSTAGE - 1)) * GI_STAGE;
        dma2(voff, xb + is_x, xb1 + is_x, dst);
        dma2(voff, yb + is_y, yb1 + is_y, dst + GI_HALF);
        ++is_t; is_x += 1024; is_y += 1024;                                 // k-step blocks of 1 KiB
        if (is_t == is_wrap) {                                              // the next segment: X from its start, Y one slice on
            is_x = 0; is_y = is_ybase;
            is_ybase += n64 * 1024; is_wrap += 2 * n64;                     // 3 n64, then 5 n64
        }
    };

    // the same, a piece at a time (piece 0..3: X group 0, X group 1, Y group 0, Y group 1): between the MFMAs of an iteration
    auto issue_piece = [&](int piece) {
        const unsigned dst = lds_w + (is_t & (GI_NSTAGE - 1)) * GI_STAGE + (piece >> 1) * GI_HALF + (piece & 1) * 1024;
        const char* src = (piece >> 1) ? ((piece & 1) ? yb1 : yb) + is_y : ((piece & 1) ? xb1 : xb) + is_x;
        dma1(voff, src, dst);
        if (piece == 3) {
            ++is_t; is_x += 1024; is_y += 1024;
            if (is_t == is_wrap) {
                is_x = 0; is_y = is_ybase;
                is_ybase += n64 * 1024; is_wrap += 2 * n64;
            }
        }
    };

    v4i acc[4][8];                          // [column group j of this wave][row group i]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = v4i{0, 0, 0, 0};
    v4i fx[2][8], fy[2][4];

    const char* sx = smem_i8 + (wr * 8) * 1024 + lane * 16;
    const char* sy = smem_i8 + GI_HALF + (wc * 4) * 1024 + lane * 16;
    // (Measured and not kept: the two waves of a SIMD issuing their DMA at opposite ends of the iteration -- 6.35 -> 6.58 ms;
    // a fifth LDS stage, all 160 KiB, four k-steps for a piece to arrive -- 6.15 vs 6.17 ms: its latency is covered.)
    // s_waitcnt immediates (gfx9: vmcnt [3:0] + [15:14], expcnt [6:4] left at 7, lgkmcnt [11:8]) with lgkmcnt(0); the
    // builtin, not inline asm, so that hipcc's own wait insertion knows the LDS reads are done
    constexpr int GI_WAIT_VM8 = 0x0078, GI_WAIT_VM4 = 0x0074, GI_WAIT_VM0 = 0x0070;
#define GI_STAMP(K)
    // (a macro: through a generic lambda hipcc kept the fragment and accumulator arrays in scratch memory; and the steady
    // state has no branch in it -- with the tail's conditions inside, hipcc moved accumulators between register files
    // in every iteration)
#define GI_BODY(T, CUR, NXT, WAIT, ISSUE)                                                                                  \
    {                                                                                                                      \
        const int t_ = (T);                                                                                                \
        /* this wave's pieces of stage t+1 (4 instructions per stage; stages up to t+3 are in flight) and its LDS reads */ \
        /* of stage t, whose slot is about to be overwritten */                                                            \
        GI_STAMP(0)                                                                                                        \
        __builtin_amdgcn_s_waitcnt(WAIT);                                                                                  \
        asm volatile("" ::: "memory");                                                                                     \
        GI_STAMP(1)                                                                                                        \
        __builtin_amdgcn_s_barrier();                                                                                      \
        asm volatile("" ::: "memory");                                                                                     \
        GI_STAMP(2)                                                                                                        \
        const int so = ((t_ + 1) & (GI_NSTAGE - 1)) * GI_STAGE;                                                            \
        /* MFMAs first (their fragments are in registers), the 12 reads of stage t+1 one per pair of MFMAs, the DMA of  */ \
        /* stage t+4 behind them: all eight waves leave the barrier together, and whatever stands between the barrier    */ \
        /* and a wave's first MFMA is time both matrix-pipe users of a SIMD spend idle (reads + DMA first: 6.8 ms)       */ \
        /* (the last iteration reads a slot nobody writes any more: unused) */                                            \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                    \
            fx[NXT][2 * j] = *(const v4i*)(sx + so + (2 * j) * 1024);                                                      \
            fx[NXT][2 * j + 1] = *(const v4i*)(sx + so + (2 * j + 1) * 1024);                                              \
            fy[NXT][j] = *(const v4i*)(sy + so + j * 1024);                                                                \
            _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                                  \
                acc[j][i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fy[CUR][j], fx[CUR][i], acc[j][i], 0, 0, 0);           \
            _Pragma("unroll") for (int g_ = 0; g_ < 3; ++g_) {                                                             \
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                         \
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                         \
            }                                                                                                              \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                             \
            /* one DMA piece of stage t+4 behind each quarter of the MFMAs (all four behind the last MFMA measured the   */  \
            /* same: cycle stamps -- GI_EXP_STAMPS -- put a stage at 1490 cycles for 1024 of MFMA per SIMD, none of it    */  \
            /* waiting for DMA data; the first wave of a SIMD is through its body after 920 and stands 530 at the barrier, */  \
            /* the second takes 1285: about 60 cycles per DMA piece that neither wave of the SIMD issues MFMAs in)        */  \
            if (ISSUE) issue_piece(j);                                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                             \
        }                                                                                                                  \
    }
    // the K stages of one segment, two per trip (the fragment buffers alternate)
#define GI_RUN(T_LO, T_HI)                                                                                                 \
    for (int t = (T_LO); t < (T_HI); t += 2) {                                                                             \
        GI_BODY(t, 0, 1, GI_WAIT_VM8, true)                                                          \
        GI_BODY(t + 1, 1, 0, GI_WAIT_VM8, true)                                                       \
    }
    auto shift = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[j][i] = acc[j][i] >> 7;
    };
#pragma unroll
    for (int t = 0; t < GI_NSTAGE; ++t) issue();                         // nst >= 24
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                    // stage 0
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 8; ++j) fx[0][j] = *(const v4i*)(sx + j * 1024);
#pragma unroll
    for (int j = 0; j < 4; ++j) fy[0][j] = *(const v4i*)(sy + j * 1024);
    // (iteration 0's barrier finds stage 0's readers done only because every wave reads it before that barrier)
    // (one copy of the steady-state loop for the three segments: as three loops in a row, hipcc gave the first one a
    // register assignment that moved 200 accumulators between the register files in every trip)
#pragma unroll 1
    for (int seg = 0; seg < 3; ++seg) {
        const int t_lo = seg == 0 ? 0 : seg == 1 ? 3 * n64 : 5 * n64;
        const int t_hi = seg == 0 ? 3 * n64 : seg == 1 ? 5 * n64 : nst - GI_NSTAGE;     // n64 >= 4: the last segment holds the tail
        GI_RUN(t_lo, t_hi)
        if (seg < 2) shift();
    }
    GI_BODY(nst - 4, 0, 1, GI_WAIT_VM8, false)       // stages nst-3 .. nst-1 in flight
    GI_BODY(nst - 3, 1, 0, GI_WAIT_VM4, false)
    GI_BODY(nst - 2, 0, 1, GI_WAIT_VM0, false)
    GI_BODY(nst - 1, 1, 0, GI_WAIT_VM0, false)
#undef GI_STAMP
#undef GI_RUN
#undef GI_BODY
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // D[m][n] of MFMA (j, i): m = column patch (wc * 4 + j) * 16 + (lane / 16) * 4 + v, n = row patch (wr * 8 + i) * 16 + lane % 16
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const long long r = m0 + (wr * 8 + i) * 16 + (lane & 15);
        if (r >= p.mrows) continue;
        int* orow = p.out + r * p.ldo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long c = n0 + (wc * 4 + j) * 16 + (lane >> 4) * 4;
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
    auto take = [&](double v) { bad |= !(fabs(v) < INFINITY); lo = fmin(lo, v); hi = fmax(hi, v); };
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x, nth = (long long)gridDim.x * 256;
    long long head = ((16 - ((unsigned long long)x & 15)) & 15) / 8;      // doubles in front of the first 16-byte boundary
    if (head > n) head = n;
    const double2* x2 = (const double2*)(x + head);
    const long long n2 = (n - head) / 2;
    long long i = tid;
    for (; i + 3 * nth < n2; i += 4 * nth) {                               // (eight loads per thread on 4096 workgroups: 0.24 -> 0.41 ms)
        const double2 a = x2[i], b = x2[i + nth], c = x2[i + 2 * nth], d = x2[i + 3 * nth];
        take(a.x); take(a.y); take(b.x); take(b.y); take(c.x); take(c.y); take(d.x); take(d.y);
    }
    for (; i < n2; i += nth) { const double2 a = x2[i]; take(a.x); take(a.y); }
    if (tid < head) take(x[tid]);
    if (tid == 0 && head + 2 * n2 < n) take(x[n - 1]);
    for (int o = 32; o > 0; o >>= 1) { lo = fmin(lo, __shfl_xor(lo, o)); hi = fmax(hi, __shfl_xor(hi, o)); }
    const bool any_bad = __ballot(bad) != 0;
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&keys[0], dlc_f64_key(lo));
        atomicMax(&keys[1], dlc_f64_key(hi));
        if (any_bad) atomicMax(&keys[2], 1ull);
    }
}

// u = clamp((x - lo) / (hi - lo)) and its 21-bit fixed-point value (a NaN lands on 0; such datasets take the fp64 route)
__device__ __forceinline__ double sim_unit(double x, double lo, double inv) {
    const double u = (x - lo) * inv;
    return u > 0.0 ? (u < 1.0 ? u : 1.0) : 0.0;
}
__device__ __forceinline__ int sim_fixed(double u) {
    const int q = (int)(u * 2097152.0);                                 // floor (u >= 0)
    return q > 2097151 ? 2097151 : q;
}

// A 128-bit content hash of a row: two sums over the elements of a strong 64-bit mix of (bit pattern, position) -- order
// independent, so lanes and waves add their parts in any grouping.  Rows with equal hashes are taken to be the same patch
// (a key-point found twice, a blank patch repeated): their distances to anything are equal however they are summed and
// np.argmin keeps the first of them, so the pair kernel drops the later ones from its undecided candidates without
// reading a row.  (Different rows collide with probability 2^-128 per comparison.)
__device__ __forceinline__ unsigned long long sim_mix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

// The pass over the descriptors that both forms of the similarity share, a workgroup per group of 16 patch rows, a wave
// per k-step of 64 elements (w, w + 4, ..), lane l on row l % 16, elements ks * 64 + (l / 16) * 16 .. + 15 -- 128
// contiguous bytes per lane:
//   |x|^2 (the fp64 Gram form's norms) and p = dot(score, row) for every row -- per lane in k order, then the row's four
//   lanes (xor 16, 32), then its four waves in order: one fixed summation order for both forms, whose projections must
//   agree bit for bit;
//   QUANT (the filter): sum u (its largest value into keys[3], an ordered key), |u|^2, and the three 7-bit slices of
//   the 21-bit fixed-point value, 16 bytes of each per lane at lane * 16 of that (group, slice, k-step) block -- 1 KiB per
//   store instruction.  Rows past the last one and k >= H are zeros there and take no part in the sums.
template <bool QUANT>
__global__ __launch_bounds__(256) void sim_rows_kernel(const double* __restrict__ desc, long long rows, int H, int kp,
                                                       const double* __restrict__ score, unsigned long long* keys,
                                                       char* __restrict__ X, char* __restrict__ Y, double* __restrict__ nrm2,
                                                       double* __restrict__ nu2, double* __restrict__ proj,
                                                       unsigned long long* __restrict__ rowhash) {
    __shared__ double red[4][16][4];
    __shared__ unsigned long long redh[4][16][2];
    unsigned long long h1 = 0, h2 = 0;
    const long long g = blockIdx.x;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, rr = lane & 15, chunk = lane >> 4;
    double lo = 0.0, inv = 0.0;
    if constexpr (QUANT) {
        const double hi = dlc_f64_unkey(keys[1]);
        lo = dlc_f64_unkey(keys[0]);
        inv = hi - lo > 0.0 ? 1.0 / (hi - lo) : 0.0;
    }
    const long long r = g * 16 + rr;
    const bool row_ok = r < rows;
    const int nks = QUANT ? kp / 64 : (H + 63) / 64;
    const double* x = desc + (row_ok ? r : 0) * H;
    const bool vec = (H & 1) == 0 && ((unsigned long long)desc & 15) == 0 && ((unsigned long long)score & 15) == 0;
    char* xg = X + g * (3ll * nks * 1024) + lane * 16;
    char* yg = Y + g * (3ll * nks * 1024) + lane * 16;
    double n2 = 0.0, pr = 0.0, su = 0.0, s2 = 0.0;
    for (int ks = w; ks < nks; ks += 4) {
        const int k0 = ks * 64 + chunk * 16;
        unsigned w1[4] = {0, 0, 0, 0}, w2[4] = {0, 0, 0, 0}, w3[4] = {0, 0, 0, 0};
        auto put = [&](int e, double v, double sc) {
            n2 = fma(v, v, n2);
            pr = fma(sc, v, pr);
            if (rowhash) {
                const unsigned long long bits = (unsigned long long)__double_as_longlong(v), pos = (unsigned long long)(k0 + e);
                h1 += sim_mix64(bits + pos * 0x9e3779b97f4a7c15ull);
                h2 += sim_mix64((bits ^ 0xd6e8feb86659fd93ull) + pos * 0xc2b2ae3d27d4eb4full);
            }
            if constexpr (QUANT) {
                const double u = sim_unit(v, lo, inv);
                su += u; s2 = fma(u, u, s2);
                const int q = sim_fixed(u);
                w1[e >> 2] |= (unsigned)(q >> 14) << (8 * (e & 3));
                w2[e >> 2] |= (unsigned)((q >> 7) & 127) << (8 * (e & 3));
                w3[e >> 2] |= (unsigned)(q & 127) << (8 * (e & 3));
            }
        };
        if (row_ok) {
            if (vec && k0 + 16 <= H) {                                  // 128 contiguous, 16-byte aligned bytes per lane
                double2 v[8], c[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { v[e] = *(const double2*)(x + k0 + 2 * e); c[e] = *(const double2*)(score + k0 + 2 * e); }
#pragma unroll
                for (int e = 0; e < 8; ++e) { put(2 * e, v[e].x, c[e].x); put(2 * e + 1, v[e].y, c[e].y); }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (k0 + e < H) put(e, x[k0 + e], score[k0 + e]);
            }
        }
        if constexpr (QUANT) {
            const uint4 v1 = make_uint4(w1[0], w1[1], w1[2], w1[3]), v2 = make_uint4(w2[0], w2[1], w2[2], w2[3]),
                        v3 = make_uint4(w3[0], w3[1], w3[2], w3[3]);
            *(uint4*)(xg + (0ll * nks + ks) * 1024) = v1;
            *(uint4*)(xg + (1ll * nks + ks) * 1024) = v2;
            *(uint4*)(xg + (2ll * nks + ks) * 1024) = v3;
            *(uint4*)(yg + (0ll * nks + ks) * 1024) = v3;
            *(uint4*)(yg + (1ll * nks + ks) * 1024) = v2;
            *(uint4*)(yg + (2ll * nks + ks) * 1024) = v1;
        }
    }
    for (int o = 16; o <= 32; o <<= 1) {
        n2 += __shfl_xor(n2, o); pr += __shfl_xor(pr, o);
        if constexpr (QUANT) { su += __shfl_xor(su, o); s2 += __shfl_xor(s2, o); }
        h1 += __shfl_xor(h1, o); h2 += __shfl_xor(h2, o);
    }
    if (chunk == 0) {
        red[w][rr][0] = n2; red[w][rr][1] = pr; red[w][rr][2] = su; red[w][rr][3] = s2;
        redh[w][rr][0] = h1; redh[w][rr][1] = h2;
    }
    __syncthreads();
    if (w == 0 && chunk == 0 && row_ok) {
        double t[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) t[c] = ((red[0][rr][c] + red[1][rr][c]) + red[2][rr][c]) + red[3][rr][c];
        if (nrm2) nrm2[r] = t[0];
        proj[r] = t[1];
        if constexpr (QUANT) {
            nu2[r] = t[3];
            atomicMax(&keys[3], dlc_f64_key(t[2]));
        }
        if (rowhash) {
            rowhash[2 * r] = redh[0][rr][0] + redh[1][rr][0] + redh[2][rr][0] + redh[3][rr][0];
            rowhash[2 * r + 1] = redh[0][rr][1] + redh[1][rr][1] + redh[2][rr][1] + redh[3][rr][1];
        }
    }
}

// NumPy's pairwise summation of n doubles (np.add.reduce along a contiguous axis, as np.linalg.norm uses it:
// SimilarityCalculator.py:34) as a postfix program: a leaf (start, n <= 128) is summed with eight strided accumulators
// -- ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)), then the n % 8 last elements one by one; fewer than 8 elements:
// one by one from 0 -- and (-1, 0) adds the two results before it.  A range of more than 128 splits at n / 2 rounded down
// to a multiple of 8.  The pair kernel runs this program where candidates are too close for anything else.
__global__ void sim_pairwise_program_kernel(int H, int2* prog, unsigned long long* len) {
    __shared__ int st_start[32], st_n[32], st_phase[32];    // (in scratch memory this one thread took 0.2 ms)
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
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

// rows of a panel: whole tiles, 256 rows of slack behind the last one (a tile of the column panel starts at any group)
static int64_t sim_panel_rows(int64_t rows) { return (int64_t)dlc::align_up((size_t)rows, (size_t)GI_T) + GI_T + 16; }

size_t sim_filter_panel_bytes(int64_t rows, int64_t H) {
    const size_t kp = dlc::align_up((size_t)H, (size_t)GI_KPAD);
    return (size_t)sim_panel_rows(rows) * 3 * kp;
}

// keys[6]: min key, max key, non-finite flag, max row sum key, direct evaluations (a count), length of prog (up to
// 1023 int2 entries for H <= 32768).  X / Y: sim_filter_panel_bytes each.
int sim_filter_prepare(dlc_ctx* ctx, const double* desc, int64_t rows, int64_t H, const double* score, unsigned long long* keys,
                       char* X, char* Y, double* nu2, double* proj, unsigned long long* rowhash, void* prog, hipStream_t st) {
    const int kp = (int)dlc::align_up((size_t)H, (size_t)GI_KPAD);
    hipLaunchKernelGGL(sim_keys_init_kernel, dim3(1), dim3(64), 0, st, keys);
    hipLaunchKernelGGL(sim_pairwise_program_kernel, dim3(1), dim3(64), 0, st, (int)H, (int2*)prog, keys + 5);
    hipLaunchKernelGGL(sim_range_kernel, dim3(2048), dim3(256), 0, st, desc, (long long)(rows * H), keys);
    DLC_LAUNCH_CHECK(ctx, "sim_range_kernel");
    hipLaunchKernelGGL(sim_rows_kernel<true>, dim3((unsigned)(sim_panel_rows(rows) / 16)), dim3(256), 0, st, desc, (long long)rows,
                       (int)H, kp, score, keys, X, Y, (double*)nullptr, nu2, proj, rowhash);
    DLC_LAUNCH_CHECK(ctx, "sim_rows_kernel");
    return DLC_OK;
}

// |x|^2, dot(score, x) and the content hash of every patch row (the fp64 Gram form; the filter's prepare computes the same
// projections and hashes), and NumPy's pairwise-summation program for rows of H elements (prog: sim_pairwise_program_bytes,
// its length to *prog_len) -- what the pair kernels need to decide the arg-mins the Gram matrix cannot.
size_t sim_pairwise_program_bytes(int64_t H) { return dlc::align_up((size_t)(H / 32 + 64) * 8, 256); }

int sim_row_sums(dlc_ctx* ctx, const double* desc, int64_t rows, int64_t H, const double* score, double* nrm2, double* proj,
                 unsigned long long* rowhash, void* prog, unsigned long long* prog_len, hipStream_t st) {
    hipLaunchKernelGGL(sim_pairwise_program_kernel, dim3(1), dim3(64), 0, st, (int)H, (int2*)prog, prog_len);
    hipLaunchKernelGGL(sim_rows_kernel<false>, dim3((unsigned)dlc::cdiv(rows, (int64_t)16)), dim3(256), 0, st, desc, (long long)rows,
                       (int)H, 0, score, (unsigned long long*)nullptr, (char*)nullptr, (char*)nullptr, nrm2, (double*)nullptr, proj,
                       rowhash);
    DLC_LAUNCH_CHECK(ctx, "sim_rows_kernel");
    return DLC_OK;
}

// out[r, c] = acc of (row patch row0 + r, column patch col0 + c), r < mrows, c < ncols; row0 and col0 multiples of 16
// (the panels' groups), ldo a multiple of 4 >= ncols; tiles without a (row frame < column frame) entry are skipped
int gram_upper_i8(dlc_ctx* ctx, int64_t mrows, int64_t ncols, int64_t H, const char* X, const char* Y, int* out, int64_t ldo,
                  int patches, int64_t row0, int64_t col0, hipStream_t st) {
    const int kp = (int)dlc::align_up((size_t)H, (size_t)GI_KPAD);
    if ((row0 & 15) || (col0 & 15)) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "gram_upper_i8: panel origins must be multiples of 16");
    GramI8Args a;
    a.gpitch = 3ll * kp * 16; a.kp = kp;
    a.X = X + (row0 / 16) * a.gpitch;
    a.Y = Y + (col0 / 16) * a.gpitch;
    a.out = out; a.ldo = ldo; a.mrows = mrows; a.ncols = ncols;
    a.tiles_m = (int)dlc::cdiv(mrows, (int64_t)GI_T); a.tiles_n = (int)dlc::cdiv(ncols, (int64_t)GI_T);
    a.nsm = (a.tiles_m + 3) / 4;
    a.nsn = (a.tiles_n + 7) / 8;
    a.nsup = 0;                               // blocks with a wanted tile (the kernel numbers them the same way)
    for (int si = 0; si < a.nsm; ++si) {
        const long long row_frame = (row0 + (long long)si * 4 * GI_T) / patches;
        long long c_need = (row_frame + 1) * patches - col0;
        if (c_need < 0) c_need = 0;
        const int cnt = a.nsn - (int)(c_need / (8 * GI_T));
        if (cnt > 0) a.nsup += cnt;
    }
    if (a.nsup == 0) return DLC_OK;
    a.tri_p = patches; a.tri_row0 = row0; a.tri_col0 = col0;
    const size_t lds = (size_t)GI_NSTAGE * GI_STAGE;
    if (!(ctx->func_attr_set & (1ull << DLC_ATTR_GRAM_I8))) {
        DLC_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)gram_i8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ctx->func_attr_set |= 1ull << DLC_ATTR_GRAM_I8;
    }
    const unsigned grid = (unsigned)(((a.nsup + 7) / 8) * 8 * 32);
    // bench.py's kernel-only timing (dlc_set_profiling): an event pair around the kernel on its stream
    const int prof_slot = (int)(ctx->prof_calls % DLC_PROFILE_RING);
    if (ctx->profiling) DLC_HIP_CHECK(ctx, hipEventRecord(ctx->ev_start[prof_slot], st));
    hipLaunchKernelGGL(gram_i8_kernel, dim3(grid), dim3(512), lds, st, a);
    DLC_LAUNCH_CHECK(ctx, "gram_i8_kernel");
    if (ctx->profiling) {
        DLC_HIP_CHECK(ctx, hipEventRecord(ctx->ev_stop[prof_slot], st));
        ctx->prof_calls++;
    }
    return DLC_OK;
}

}  // namespace dlc_gemm
