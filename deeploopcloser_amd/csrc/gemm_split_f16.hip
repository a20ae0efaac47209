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
// Kernel.  The three products of a k-slice share ONE set of operands (r04 ran them as one plain GEMM over K' = 3 K: six
// operand tiles staged and six sets of fragments read per 64-deep k-tile, where four are distinct -- DMA issue and LDS
// fragment reads were worth 17 % of the kernel each, docs/LAB.md 10.2).  Per 32-deep k-slice (one MFMA of K) the four
// tiles W1, W2 (256 weight columns), h1, h2 (256 activation rows) are staged once (64 KiB by LDS-DMA), each wave reads its
// fragments of the four once (W1, W2: 8 x 16 B per lane; h1, h2: 4 x 16 B) and issues 96 MFMAs from registers:
//   P1 = h1 . W1,  P2 = h2 . W1,  P3 = h1 . W2       (8 x 4 tiles of 16 x 16 each, v_mfma_f32_16x16x32_f16)
// with ONE workgroup barrier per slice (r04: four per 64 MFMAs), behind P1: by then every wave holds all four fragment
// sets of the slice (W1, h1 were read during P3 of the slice before, h2 and W2 between P1's MFMAs) and the slice's LDS is
// dead.  The LDS is a ring of 2 1/2 slices, all 160 KiB: two slots for (W2, h2) pairs, three for (W1, h1) pairs.  Behind the
// barrier of slice s the loaders issue (W2, h2) of slice s + 2 -- wanted one period later -- and then (W1, h1) of slice
// s + 3, which has TWO periods to land: the wait in front of a barrier leaves the youngest (W1, h1) pair in flight, so the
// L2 -> LDS stream never drains (with two whole-slice stages every refill had to be issued behind one barrier and land in
// front of the next: 64 KiB take ~2900 cycles round trip on a CU, the slice's MFMAs 3072 -- the period was their sum's
// better part, 3900 cycles; in-kernel stamps, docs/LAB.md 11.3).
//
// Operand format ("K32-major", private to this file -- both operands are written by kernels here): element (row r, k) of a
// piece lives at byte ((k / 32) * rows_pad + r) * 64 + (k % 32) * 2, so that the 256 rows x 32 k of one tile and slice are
// ONE contiguous 16 KiB block: sixteen LDS-DMA wave-instructions of 1 KiB, every one a fully coalesced read (a row-major
// operand would hand each of them sixteen half lines).  The LDS image is that block with the four 16-byte slots of a row
// XOR-ed by 2 * bit 3 of the row (on the DMA's per-lane source offset; the destination is lane-linear), which makes the
// fragment reads -- lane (i, kq) takes slot kq of row i of a 16-row block, ds_read_b128 -- conflict-free in each of the
// instruction's four 16-lane groups.
// Roles: MFMA A operand = 256 weight columns; the weight panel's rows are PERMUTED inside every 64 (column 16 g + 4 t + r
// sits at row 16 t + 4 g + r) so that sixteen consecutive panel rows are one A fragment whose accumulators are, per lane,
// sixteen CONSECUTIVE output columns.  B operand = 256 activation rows (a lane holds ONE row m).  The epilogue therefore
// holds, per lane, runs of sixteen consecutive outputs of one row: bias + sigmoid in fp32 (one fma, v_exp_f32, v_rcp_f32 per
// element), then either the next layer's two fp16 pieces (32 bytes each, K32-major; lanes l and l + 32 trade halves so that a
// store instruction writes one whole slice: 1 KiB in a piece) or, for the last layer, fp64 values that leave through LDS a
// row at a time.  The activations never exist as fp64 between the layers.
// Both operands are re-read (neither is a once-only stream).  Tile order: the tiles in "super-rows" of four activation row
// tiles, column by column inside a super-row, cut into EIGHT CONTIGUOUS RANGES, one per XCD (workgroup ids go round-robin
// to the XCDs; an XCD starts its ids in order, one workgroup per CU): the 32 tiles an XCD runs at a time are 4 rows x 8
// columns of tiles -- 12 operand panels out of that XCD's L2 -- and every XCD gets its 1 / 8 of the tiles to within one.
// (r04 .. the first r05 form dealt whole blocks of 6 x 5 tiles to the XCDs: 1250 tiles = 42 blocks = SIX rounds on two of
// the XCDs where 1250 / 256 = 4.9 fit in five -- a sixth of the kernel's time, found through the in-kernel stamps: the k
// loop accounted for 140 of a tile's 186 us.)
#include "gemm_internal.h"

namespace dlc_gemm {
namespace {

constexpr int SP_BM = 256;                     // weight columns per tile (MFMA A operand)
constexpr int SP_BN = 256;                     // activation rows per tile (MFMA B operand)
constexpr int SP_THREADS = 512;
constexpr int SP_KS = 32;                      // k per slice: one v_mfma_f32_16x16x32_f16
constexpr int SP_ROWB = SP_KS * 2;             // 64 bytes of a row per slice
constexpr int SP_OP = 256 * SP_ROWB;           // one operand tile of one slice: 16 KiB, contiguous in memory and in LDS
constexpr int SP_PAIR_B = 2 * SP_OP;           // a slot: one weight tile + one activation tile of a slice, 32 KiB
constexpr int SP_RING_B = 0;                   // (W2, h2) slots 0, 1
constexpr int SP_RING_A = 2 * SP_PAIR_B;       // (W1, h1) slots 0, 1, 2
constexpr int SP_LDS = 5 * SP_PAIR_B;          // 160 KiB: all of a CU's LDS
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
constexpr int SP_X_SHIFT = 11;                 // activations are carried as h * 2^11
constexpr int SP_SUPER = 4;                    // activation row tiles per super-row of the tile order

typedef __attribute__((address_space(3))) void* lptr_t;

struct SplitArgs {
    const char* W[2];           // weight pieces, K32-major: [K/32][Np rows (permuted inside 64s)][32] fp16 (rows >= N, k >= K: zeros)
    const char* X[2];           // activation pieces, K32-major: [slices][Mp rows][32] fp16
    long long wslice_b, xslice_b;   // bytes from one k-slice to the next: Np * 64, Mp * 64
    long long M;                // activation rows
    int N;                      // output columns
    int ns;                     // k-slices of 32
    const double* bias;         // [N] or null
    const float* wscale;        // device: 2^s of this layer's weights
    char* O[2];                 // next layer's pieces, K32-major with oslice_b per slice (null for the last layer) ...
    long long oslice_b;
    double* C;                  // ... whose output is fp64 [M, ldc]
    long long ldc;
    long long tiles_m;          // activation row tiles
    int tiles_n;                // weight column tiles
    long long per_xcd, extra;   // tiles per XCD: per_xcd (+ 1 for the first `extra` XCDs), in the order of the file header
};

// LDS-DMA wave-instructions of 1 KiB each: inline asm so that hipcc does not count them in vmcnt, M0 saved and restored
// (cosine_topk.hip: dma4).  The instruction's immediate offset is
// added to the global address AND to the LDS address (M0 + offset + 16 * lane), and a run is laid out alike on both
// sides: one address register and one M0 value serve every piece of it.
// Four pieces (a 4 KiB run) in one statement, SKIPPED by a wave whose `on` is 0: the jump is inside the statement, so that
// the compiler sees straight-line code (a C-level `if` around the statement -- basic blocks inside the k loop -- cost the
// register allocator its footing at 252 of 256 registers: spills).
__device__ __forceinline__ void sp_dma4(unsigned on, unsigned voff, const char* sbase, unsigned lds0) {
    unsigned keep;
    asm volatile(
        "s_cmp_eq_u32 %4, 0\n\t"
        "s_cbranch_scc1 .Lsp_skip_%=\n\t"
        "s_nop 4\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:3072\n\t"
        "s_mov_b32 m0, %0\n"
        ".Lsp_skip_%=:"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds0), "s"(on)
        : "memory", "scc");
}
// s_waitcnt vmcnt(N0) for a wave whose `sel` is 0, vmcnt(N1) otherwise -- the choice inside the statement (see sp_dma4)
template <int N0, int N1>
__device__ __forceinline__ void sp_wait_vmcnt_by(unsigned sel) {
    asm volatile(
        "s_cmp_eq_u32 %0, 0\n\t"
        "s_cbranch_scc1 .Lsp_w0_%=\n\t"
        "s_waitcnt vmcnt(%2)\n\t"
        "s_branch .Lsp_w1_%=\n"
        ".Lsp_w0_%=:\n\t"
        "s_waitcnt vmcnt(%1)\n"
        ".Lsp_w1_%=:"
        :
        : "s"(sel), "n"(N0), "n"(N1)
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
    // workgroup -> tile: ids go round-robin to the 8 XCDs; XCD x walks its own contiguous range of the tile order
    long long tile_m;
    int tile_n;
    {
        const long long id = blockIdx.x;
        const long long x = id & 7, l = id >> 3;
        if (l >= p.per_xcd + (x < p.extra ? 1 : 0)) return;
        const long long t = x * p.per_xcd + (x < p.extra ? x : p.extra) + l;
        const long long full = (p.tiles_m / SP_SUPER) * SP_SUPER * p.tiles_n;       // tiles in whole super-rows
        if (t < full) {
            const long long w = t % ((long long)SP_SUPER * p.tiles_n);
            tile_m = t / ((long long)SP_SUPER * p.tiles_n) * SP_SUPER + w % SP_SUPER;
            tile_n = (int)(w / SP_SUPER);
        } else {
            const long long rr = p.tiles_m % SP_SUPER, w = t - full;
            tile_m = p.tiles_m / SP_SUPER * SP_SUPER + w % rr;
            tile_n = (int)(w / rr);
        }
    }
    const unsigned lds_base = (unsigned)(unsigned long long)(lptr_t)smem_sp;

    // ---- DMA roles.  A piece costs its wave ~100 cycles of issue, 64 pieces a slice: all eight waves carry eight each, the
    // two waves of a SIMD (w and w + 4) in ANTIPHASE.  Waves 0-3 ("early") stage the (W2, h2) pairs -- wanted within one
    // period -- right behind the barrier, between P2's first MFMAs; waves 4-7 ("late") stage the (W1, h1) pairs, which have
    // two periods, between P3's: while one wave of a SIMD is inside its DMA statements (no matrix instruction for hundreds
    // of cycles) the matrix pipe has the other's.  Wave w stages half (w & 1) of the weight tile (w & 2 == 0) or of the
    // activation tile of its class: eight pieces a slice.
    const unsigned late = __builtin_amdgcn_readfirstlane(wid >> 2);            // 0: (W2, h2) loader, 1: (W1, h1) loader
    const unsigned early = late ^ 1u;
    const int lw = wid & 3;
    unsigned voff;                                          // this lane's 16 B inside a piece
    {
        const int row = lane >> 2, slot = lane & 3;         // a piece = 16 rows x 4 slots; LDS slot s of row r holds source slot s ^ 2 * bit3(r)
        voff = (unsigned)(row * SP_ROWB + ((slot ^ (((row >> 3) & 1) << 1)) << 4));
    }
    const long long slice_b = lw < 2 ? p.wslice_b : p.xslice_b;
    const long long tile_off = (lw < 2 ? (long long)tile_n * SP_BM * SP_ROWB : tile_m * SP_BN * SP_ROWB) + (lw & 1) * (SP_OP / 2);
    const int piece_sel = late ? 0 : 1;                     // late waves stage the first pieces (W1 / h1), early ones the second
    const char* src = sp_uniform_ptr((lw < 2 ? p.W[piece_sel] : p.X[piece_sel]) + tile_off);
    const unsigned lds_mine = lds_base + (late ? (unsigned)SP_RING_A : (unsigned)SP_RING_B) + (unsigned)(lw >> 1) * SP_OP +
                              (unsigned)(lw & 1) * (SP_OP / 2);                // inside slot 0 of this wave's ring
    const int ns = p.ns;

    // ---- fragment read addresses: lane (i, kq) -> row i, slot kq ^ 2 * bit3(i) of a 16-row block
    const int i = lane & 15;
    const int kq = lane >> 4;
    typedef const __attribute__((address_space(3))) u32x4_t* lds_u4p;
    typedef const __attribute__((address_space(3))) char* lds_cp;
    const lds_cp lbase = (lds_cp)(lptr_t)smem_sp;
    const unsigned rd_l = (unsigned)(i * SP_ROWB + ((kq ^ (((i >> 3) & 1) << 1)) << 4));
    const unsigned rdW = rd_l + (unsigned)wr * (SP_OP / 2);                      // weight fragment T at + T KiB inside a slot
    const unsigned rdH = rd_l + (unsigned)SP_OP + (unsigned)wc * (SP_OP / 4);    // activation fragment c at + c KiB
    unsigned sb = 0;          // byte offset of the (W2, h2) slot of the current slice: s % 2
    unsigned sa_next = SP_PAIR_B;   // ... of the (W1, h1) slot of slice s + 1: (s + 1) % 3
    unsigned sa_free = 0;           // ... of slice s's own (W1, h1) slot, refilled with slice s + 3: s % 3

    f32x4_t acc[8][4];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[t][c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    u32x4_t fW1[8], fW2[8], fhA[4], fhB[4];

#define SP_READ_W(DST, BASE) _Pragma("unroll") for (int tt = 0; tt < 8; ++tt) DST[tt] = *(lds_u4p)(lbase + (BASE) + rdW + tt * 1024)
#define SP_READ_H(DST, BASE) _Pragma("unroll") for (int c = 0; c < 4; ++c) DST[c] = *(lds_u4p)(lbase + (BASE) + rdH + c * 1024)
    // MFMAs E0 .. E0 + NE - 1 of one product, in the order e -> (tile e / 4, row block e % 4)
#define SP_MFMA(FA, FB, E0, NE)                                                                                  \
    do {                                                                                                         \
        _Pragma("unroll") for (int e = (E0); e < (E0) + (NE); ++e)                                               \
            acc[e >> 2][e & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(                                         \
                __builtin_bit_cast(f16x8_t, FA[e >> 2]), __builtin_bit_cast(f16x8_t, FB[e & 3]), acc[e >> 2][e & 3], 0, 0, 0); \
    } while (0)
    // "N times one MFMA, then one LDS read" for the scheduler: the fragment reads ride in the issue slots an MFMA leaves
    // (it holds the SIMD's vector issue for 8 of its 16 cycles), not in a block of their own with the matrix pipe idle
#define SP_PAIR(N) _Pragma("unroll") for (int z_ = 0; z_ < (N); ++z_) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#define SP_ONLY_MFMA(N) __builtin_amdgcn_sched_group_barrier(0x008, (N), 0)
    // pieces 4 Q .. 4 Q + 3 (Q = 0, 1) of slice S2's pair into the slot at byte SLOT of this wave's ring, issued by the
    // waves whose ON is set (past the end: clamped -- the redundant DMA lands in a dead slot and keeps the vmcnt
    // bookkeeping uniform); the other waves jump over the statement
#define SP_ISSUE4(ON, SLOT, S2, Q)                                                                               \
    do {                                                                                                         \
        const int ss_ = (S2) < ns ? (S2) : ns - 1;                                                               \
        sp_dma4(ON, voff, src + (long long)ss_ * slice_b + (Q) * 4096, lds_mine + (SLOT) + (Q) * 4096);          \
    } while (0)

    // One slice, entered with W1 and HCUR = h1 of it in registers; leaves W1 and HNXT = h1 of the next.
#define SP_STEP(S, HCUR, HNXT)                                                                                   \
    do {                                                                                                         \
        SP_READ_H(HNXT, SP_RING_B + sb);                           /* h2 (s) */                                   \
        SP_READ_W(fW2, SP_RING_B + sb);                            /* W2 (s) */                                   \
        SP_MFMA(fW1, HCUR, 0, 32);                                 /* P1 = h1 . W1, the twelve reads between */   \
        SP_PAIR(12); SP_ONLY_MFMA(20);                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        sp_wait_vmcnt_by<0, 8>(late);                              /* early: (W2, h2) of s + 1 landed; late: all but its youngest (W1, h1) pair */ \
        SP_WAIT_LGKM0();                                           /* this wave's reads of slice s are done */     \
        sp_barrier();                                              /* slice s is dead; slice s + 1 is visible */  \
        SP_ISSUE4(early, sb, (S) + 2, 0);                          /* (W2, h2) of s + 2: wanted in one period */  \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        SP_MFMA(fW1, HNXT, 0, 4);                                  /* P2 = h2 . W1 */                            \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        SP_ISSUE4(early, sb, (S) + 2, 1);                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        SP_MFMA(fW1, HNXT, 4, 28);                                                                               \
        SP_READ_H(HNXT, SP_RING_A + sa_next);                      /* h1, W1 (s + 1) */                           \
        SP_READ_W(fW1, SP_RING_A + sa_next);                                                                     \
        SP_MFMA(fW2, HCUR, 0, 12);                                 /* P3 = h1 . W2 */                            \
        SP_ONLY_MFMA(28); SP_PAIR(12);                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        SP_ISSUE4(late, sa_free, (S) + 3, 0);                      /* (W1, h1) of s + 3: two periods to land */  \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        SP_MFMA(fW2, HCUR, 12, 4);                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        SP_ISSUE4(late, sa_free, (S) + 3, 1);                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        SP_MFMA(fW2, HCUR, 16, 16);                                                                              \
        sb ^= SP_PAIR_B;                                                                                         \
        sa_free = sa_next;                                                                                       \
        sa_next = sa_next == 2 * SP_PAIR_B ? 0u : sa_next + SP_PAIR_B;                                           \
    } while (0)

    // this thread's bias of the tile (threads 0 .. 255), folded for the epilogue's one fma (see there); fetched here, a
    // register across the k loop, because behind the loop its miss was 2 000 cycles that every wave of the tile waited for
    float bias_l = 0.0f;
    if (tid < SP_BM) {
        const int n = tile_n * SP_BM + tid;
        bias_l = (float)(-1.4426950408889634 * ((p.bias && n < p.N) ? p.bias[n] : 0.0) - (FINAL ? 0.0 : (double)SP_X_SHIFT));
    }
    // ---- prologue: early waves issue (W2, h2) of slices 0, 1; late waves (W1, h1) of slices 0, 1, 2; slice 0 landed and
    // visible; W1, h1 (0) read
    SP_ISSUE4(early, 0, 0, 0); SP_ISSUE4(early, 0, 0, 1);
    SP_ISSUE4(early, SP_PAIR_B, 1, 0); SP_ISSUE4(early, SP_PAIR_B, 1, 1);
    SP_ISSUE4(late, 0, 0, 0); SP_ISSUE4(late, 0, 0, 1);
    SP_ISSUE4(late, SP_PAIR_B, 1, 0); SP_ISSUE4(late, SP_PAIR_B, 1, 1);
    SP_ISSUE4(late, 2 * SP_PAIR_B, 2, 0); SP_ISSUE4(late, 2 * SP_PAIR_B, 2, 1);
    sp_wait_vmcnt_by<8, 16>(late);                             // slice 0 landed (this wave's share)
    sp_barrier();
    SP_READ_W(fW1, SP_RING_A + 0);
    SP_READ_H(fhA, SP_RING_A + 0);
    for (int s = 0; s < ns; s += 2) {
        SP_STEP(s, fhA, fhB);
        if (s + 1 < ns) SP_STEP(s + 1, fhB, fhA);
    }
    SP_WAIT_VMCNT(0);    // the clamped tail DMAs must not outlive the workgroup's LDS
    SP_WAIT_LGKM0();
#undef SP_STEP
#undef SP_ISSUE4
#undef SP_PAIR
#undef SP_ONLY_MFMA
#undef SP_READ_W
#undef SP_READ_H
#undef SP_MFMA

    // ---- epilogue.  C/D layout: column = lane & 15 -> activation row, row = 4 * (lane >> 4) + reg -> weight column; with
    // the permuted panel rows, acc[th * 4 + tt][c][r] is output (m, n0 + 4 tt + r): sixteen consecutive columns per (c, th).
    const int lg = lane >> 4;
    // sigmoid(z) = 1 / (1 + 2^(-z log2 e)), z = acc * inv + b with inv = 1 / (2^11 * the weights' scale), a power of two.
    // The constants are folded into the ONE fma in front of v_exp_f32: zl = acc * (-inv log2 e) + (-b log2 e); a hidden
    // layer wants h * 2^11 = 1 / (2^-11 + 2^(zl - 11)), so its -11 rides in the staged bias too.  The reciprocal is
    // v_rcp_f32 (1 ulp): `1.0f / x` and __frcp_rn are the IEEE division here -- ten instructions per element, 1280 per lane
    // and tile, a third of the epilogue (the tolerance of this mode is 2^-15 of the descriptors, not the last bit).
    const float ninv = -1.4426950408889634f / ((float)(1 << SP_X_SHIFT) * p.wscale[0]);
    const float one = FINAL ? 1.0f : 1.0f / (float)(1 << SP_X_SHIFT);
    // the tile's 256 biases through LDS (the ring is dead): ONE global load per thread; a lane's sixteen then come as four
    // 16-byte LDS reads (read straight from memory they were 32 dependent 8-byte loads per lane: a tenth of a tile's time)
    {
        sp_barrier();                                          // every wave is past its last fragment read
        float* bl = (float*)smem_sp;
        if (tid < SP_BM) bl[tid] = bias_l;
        __syncthreads();
    }
    constexpr int SP_TP = 68;                                 // floats per row of a wave's 64 x 64 block: 16 rows of float4 writes / a row of float2 reads each sweep the 64 banks once
    float* tl = (float*)(smem_sp + 4096) + wid * 64 * SP_TP;  // (behind the biases; 8 waves x 17 KiB)
#pragma unroll
    for (int th = 0; th < 2; ++th) {
        const int n0 = tile_n * SP_BM + wr * 128 + th * 64 + 16 * lg;
        const bool whole = tile_n * SP_BM + wr * 128 + th * 64 + 64 <= p.N;      // all 64 columns of this wave's half exist
        float bv[16];
        {
            const float4* b4 = (const float4*)((const float*)smem_sp + wr * 128 + th * 64 + 16 * lg);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 v = b4[j];
                bv[4 * j] = v.x; bv[4 * j + 1] = v.y; bv[4 * j + 2] = v.z; bv[4 * j + 3] = v.w;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const long long m = tile_m * SP_BN + wc * 64 + c * 16 + i;
            if (m >= p.M) continue;
            float hv[16];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float zl = fmaf(acc[th * 4 + tt][c][r], ninv, bv[4 * tt + r]);
                    hv[4 * tt + r] = __builtin_amdgcn_rcpf(one + __builtin_amdgcn_exp2f(zl));   // sigmoid (TensorflowWrapper.py:77-78) [* 2^11]
                }
            if constexpr (FINAL) {
                // into the wave's LDS block, row (c, i), columns 16 lg .. + 15: the stores below want whole rows
                float* row = tl + (c * 16 + i) * SP_TP + 16 * lg;
#pragma unroll
                for (int j = 0; j < 16; j += 4) *(float4*)(row + j) = make_float4(hv[j], hv[j + 1], hv[j + 2], hv[j + 3]);
            } else {
                // the next layer's two pieces of h * 2^11; columns past N are zeros there (its weights' k rows past N too)
                unsigned w1[8], w2[8];
                if (!whole) {                                  // (wave-uniform: only the last column tile has such columns)
#pragma unroll
                    for (int j = 0; j < 16; ++j) hv[j] = n0 + j < p.N ? hv[j] : 0.0f;
                }
#pragma unroll
                for (int j = 0; j < 16; j += 2) {
                    const f32x2_t v = {hv[j], hv[j + 1]};
                    const f16x2_t a = __builtin_convertvector(v, f16x2_t);                 // v_cvt_pk_f16_f32
                    const f16x2_t b = __builtin_convertvector(v - __builtin_convertvector(a, f32x2_t), f16x2_t);
                    w1[j >> 1] = __builtin_bit_cast(unsigned, a);
                    w2[j >> 1] = __builtin_bit_cast(unsigned, b);
                }
                // K32-major: columns n0 .. n0 + 15 are half of row m's 64 bytes in slice n0 / 32 -- the wave's 64 columns are two
                // slices, lanes 0-31 hold the first one's.  Stored as they lie, an instruction wrote 16 of every row's 64 bytes
                // in each slice; v_permlane32_swap trades the second 16 bytes of the lower lanes for the first 16 of the upper
                // ones, and every instruction writes ONE slice: sixteen rows x 64 bytes = 1 KiB in a piece.
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const u32x2_t s1 = __builtin_amdgcn_permlane32_swap(w1[e], w1[4 + e], false, false);
                    const u32x2_t s2 = __builtin_amdgcn_permlane32_swap(w2[e], w2[4 + e], false, false);
                    w1[e] = s1[0]; w1[4 + e] = s1[1];
                    w2[e] = s2[0]; w2[4 + e] = s2[1];
                }
                const int nb = tile_n * SP_BM + wr * 128 + th * 64;                      // the wave's 64 columns: slices nb / 32, + 1
                const long long o = (long long)(nb >> 5) * p.oslice_b + m * SP_ROWB + (lg & 1) * 32 + (lg >> 1) * 16;
                char* o1 = p.O[0] + o;
                char* o2 = p.O[1] + o;
                *(uint4*)o1 = make_uint4(w1[0], w1[1], w1[2], w1[3]);
                *(uint4*)(o1 + p.oslice_b) = make_uint4(w1[4], w1[5], w1[6], w1[7]);
                *(uint4*)o2 = make_uint4(w2[0], w2[1], w2[2], w2[3]);
                *(uint4*)(o2 + p.oslice_b) = make_uint4(w2[4], w2[5], w2[6], w2[7]);
            }
        }
        if constexpr (FINAL) {
            // fp64 out, a ROW at a time: the wave's 64 rows x 64 columns come back from LDS with 32 lanes on one row's 512
            // bytes, two rows per store instruction.  (Stored from the accumulators' layout every lane wrote 16 bytes of a
            // different cache line: 64 lines per instruction, 0.14 ms of this layer's 0.99.)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const int nb = tile_n * SP_BM + wr * 128 + th * 64, cj = 2 * (lane & 31);
            const bool pairs = (p.ldc & 1) == 0 && ((unsigned long long)p.C & 15) == 0;
#pragma unroll 4
            for (int it = 0; it < 32; ++it) {
                const int r = 2 * it + (lane >> 5);
                const long long m = tile_m * SP_BN + wc * 64 + r;
                const float2 v = *(const float2*)(tl + r * SP_TP + cj);
                if (m >= p.M) continue;
                double* dst = p.C + m * p.ldc + nb + cj;
                if (pairs && nb + cj + 2 <= p.N) *(double2*)dst = make_double2((double)v.x, (double)v.y);
                else {
                    if (nb + cj < p.N) dst[0] = (double)v.x;
                    if (nb + cj + 1 < p.N) dst[1] = (double)v.y;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the next half's rows overwrite the block
            __builtin_amdgcn_wave_barrier();
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
// panel row of weight column n: inside every 64 columns, column 16 g + 4 t + r sits at row 16 t + 4 g + r (the kernel's header)
__device__ __forceinline__ long long sp_panel_row(long long n) {
    const int c = (int)(n & 63);
    return (n & ~63ll) | (long long)((((c >> 2) & 3) << 4) | ((c >> 4) << 2) | (c & 3));
}
// W [K, N] fp64 -> its two fp16 pieces, K32-major [kp / 32][Np rows][32] (k contiguous inside a row's 64 bytes), zeros past
// K / N: 64 x 64 tiles through LDS
__global__ __launch_bounds__(256) void sp_split_weights_kernel(const double* __restrict__ w, long long K, long long N, const float* wscale,
                                                               unsigned short* __restrict__ p1, unsigned short* __restrict__ p2,
                                                               long long np) {
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
        const long long k = k0 + tx;
        const long long o = ((k >> 5) * np + sp_panel_row(n0 + nn)) * SP_KS + (k & 31);
        p1[o] = __builtin_bit_cast(unsigned short, (_Float16)t1[tx][nn]);
        p2[o] = __builtin_bit_cast(unsigned short, (_Float16)t2[tx][nn]);
    }
}
// x [rows, K] fp64 -> the two fp16 pieces of x 2^11, K32-major [kp / 32][mp rows][32], zeros past K (rows past `rows` are
// never read into a stored output and stay unwritten).  645 MB in 170 us for 1063 frames = 3.8 TB/s, the rate of a device copy
// on this chip; two other forms were measured and dropped: eight consecutive k per thread (64 cache lines per load instruction:
// 258 us) and 32 rows x 256 k per workgroup turned through LDS (whole-KiB stores, 256-byte reads: 211 us).
__global__ __launch_bounds__(256) void sp_split_rows_kernel(const double* __restrict__ x, long long rows, long long K, long long kp,
                                                            unsigned short* __restrict__ p1, unsigned short* __restrict__ p2,
                                                            long long mp) {
    const long long total = rows * kp;
    const double sc = (double)(1 << SP_X_SHIFT);
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long r = e / kp, k = e - r * kp;
        const double v = k < K ? x[r * K + k] * sc : 0.0;
        const _Float16 a = (_Float16)v;
        const long long o = ((k >> 5) * mp + r) * SP_KS + (k & 31);
        p1[o] = __builtin_bit_cast(unsigned short, a);
        p2[o] = __builtin_bit_cast(unsigned short, (_Float16)(v - (double)a));
    }
}

struct PanelLayout {
    size_t scale, amax, p1, p2, total;          // byte offsets inside one layer's panel block
    long long np, kp;
};
PanelLayout panel_layout(int64_t K, int64_t N) {
    PanelLayout L;
    L.np = (long long)dlc::align_up((size_t)N, (size_t)SP_BM);
    L.kp = (long long)dlc::align_up((size_t)K, 64);           // (whole 64 x 64 tiles of the preparation kernel; the GEMM walks align(K, 32))
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
                           (unsigned short*)(base + L.p2), L.np);
        DLC_LAUNCH_CHECK(ctx, "sp_split_weights_kernel");
        off += L.total;
    }
    return DLC_OK;
}

// the k-slices an activation piece holds for a layer input of width K: whole 256-column output tiles of the layer before
static long long split_slices(int64_t K) { return (long long)dlc::align_up((size_t)K, (size_t)SP_BM) / SP_KS; }
// ... and its rows: whole 256-row tiles (the GEMM stages full tiles; rows past the last are never stored)
static long long split_rows_pad(int64_t rows) { return (long long)dlc::align_up((size_t)rows, (size_t)SP_BN); }

size_t split_encode_workspace_bytes(int64_t rows, const int64_t* dims, int n_layers) {
    long long smax = 0;
    for (int l = 0; l < n_layers; ++l) smax = split_slices(dims[l]) > smax ? split_slices(dims[l]) : smax;
    return 4 * dlc::align_up((size_t)split_rows_pad(rows) * (size_t)smax * SP_ROWB, 256);          // two ping-pong buffers of two pieces
}

int split_encode(dlc_ctx* ctx, int64_t rows, int n_layers, const int64_t* dims, const double* x, const char* panels,
                 const double* const* b, double* out, char* ws, hipStream_t st) {
    long long smax = 0;
    for (int l = 0; l < n_layers; ++l) smax = split_slices(dims[l]) > smax ? split_slices(dims[l]) : smax;
    const long long mp = split_rows_pad(rows);
    const size_t piece = dlc::align_up((size_t)mp * (size_t)smax * SP_ROWB, 256);
    char* buf[2][2] = {{ws, ws + piece}, {ws + 2 * piece, ws + 3 * piece}};
    {
        const long long kp = (long long)dlc::align_up((size_t)dims[0], (size_t)SP_KS);
        long long blocks = dlc::cdiv(rows * kp, (int64_t)256);
        if (blocks > 256 * 32) blocks = 256 * 32;
        hipLaunchKernelGGL(sp_split_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, (long long)rows, (long long)dims[0], kp,
                           (unsigned short*)buf[0][0], (unsigned short*)buf[0][1], mp);
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
        a.W[0] = base + L.p1; a.W[1] = base + L.p2; a.wslice_b = L.np * SP_ROWB;
        a.X[0] = buf[l & 1][0]; a.X[1] = buf[l & 1][1]; a.xslice_b = mp * SP_ROWB;
        a.M = rows; a.N = (int)dims[l + 1]; a.ns = (int)dlc::cdiv(dims[l], (int64_t)SP_KS);
        a.bias = b ? b[l] : nullptr;
        a.wscale = (const float*)(base + L.scale);
        const bool fin = l == n_layers - 1;
        a.O[0] = fin ? nullptr : buf[(l + 1) & 1][0]; a.O[1] = fin ? nullptr : buf[(l + 1) & 1][1];
        a.oslice_b = fin ? 0 : mp * SP_ROWB;
        a.C = fin ? out : nullptr; a.ldc = dims[l + 1];
        a.tiles_m = dlc::cdiv(rows, (int64_t)SP_BN);
        a.tiles_n = (int)(L.np / SP_BM);
        const long long tiles = a.tiles_m * a.tiles_n;
        a.per_xcd = tiles / 8; a.extra = tiles % 8;
        const long long nwg = (a.per_xcd + (a.extra ? 1 : 0)) * 8;
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
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    return dlc_gemm::split_encode(ctx, rows, n_layers, dims, x, (const char*)panels, b, out, (char*)workspace, (hipStream_t)stream);
}
