// Cosine-similarity + top-k match for gfx950 (MI355X).
//
// Pipeline of one dlc_cosine_topk() call (all on the caller's stream):
//   1. score_gemm_kernel   S~ = DB . Q^T on the bf16/f16 MFMA (fp32 accumulate).
//      The score tile never leaves the accumulators: the epilogue keeps, per
//      query, the maximum of every aligned block of 8 database rows
//      ("group", gmax) and of every 128 rows ("half tile", tmax).  The
//      database is streamed from HBM exactly once.
//   2. finish_topk_kernel  one workgroup per query:
//      a. the kg = k + SLACK half tiles with the largest tmax, then the kg groups
//         with the largest gmax inside them.  Every member of the exact top-k
//         lies in one of those groups (the k-th largest group maximum is a
//         lower bound of the k-th largest score);
//      b. exact fp32 dot products for the 8 rows of each selected group (a
//         gather of kg*8 rows per query);
//      c. top-k of the kg*8 candidates, score descending, ties toward the
//         lower database index.
// dlc_topk_merge() is step 2c on an all-gather of per-shard results.
//
// MFMA operand roles: A = database rows, B = queries, so that in the 16x16 C/D
// layout (col = lane&15, row = 4*(lane>>4)+reg) a lane holds ONE query and
// FOUR database rows per tile: the per-group maximum is an in-lane v_max chain.
// The A fragment's row i of MFMA tile tt is mapped to database row
// 16*(i>>2) + 4*tt + (i&3) of the wave's 64-row half, which makes each lane's
// 16 accumulators (4 tiles x 4 regs) one CONTIGUOUS block of 16 database rows
// (two groups of 8).
#include "dlc_internal.h"

#include <algorithm>

namespace {

constexpr int BM = 256;          // database rows per workgroup tile
constexpr int BNQ = 256;         // queries per workgroup tile
constexpr int BK = 64;           // K step (elements); rows of the LDS image are 128 B
constexpr int NTHREADS = 512;    // 8 waves: 2 (database halves of 128 rows) x 4 (64 queries)
constexpr int TILE_BYTES = 256 * BK * 2;   // 32 KiB: one operand tile
constexpr int A_TILE = TILE_BYTES;         // ring strides
constexpr int B_TILE = TILE_BYTES;
#ifndef DLC_A_STAGES
#define DLC_A_STAGES 2                     // depth of the database-operand ring (2 -> 128 KiB LDS, 3 -> 160 KiB)
#endif
constexpr int A_STAGES = DLC_A_STAGES;
constexpr int B_RING = A_STAGES * A_TILE;  // A ring first, then the B ring of 2 K tiles (64 KiB)
constexpr int LDS_BYTES = A_STAGES * A_TILE + 2 * B_TILE;
#ifndef DLC_STAGGER_SLEEP
#define DLC_STAGGER_SLEEP 15     // s_sleep units of 64 cycles per stagger step (~0.5 us); x stagger_mult
#endif
#ifndef DLC_STAGGER_PHASES
#define DLC_STAGGER_PHASES 32    // launches of >= 3 dispatch rounds (r02, with the non-temporal database stream: 32 phases 1-2 %
#endif                           // faster than 16, 64 the same as 32; no stagger at all now costs 1 %, it was 8 % before `nt`:
                                 // scripts/exp_gemm.py, three rounds); smaller launches keep 16 phases of half the length

constexpr int GROUP = 8;         // database rows per group
constexpr int HALF = 128;        // database rows per half tile
constexpr int GROUPS_PER_HALF = HALF / GROUP;
constexpr int SLACK = 4;         // extra groups kept beyond k (fp32 re-score order vs MFMA order)

template <typename Tag> struct Mfma16;
template <> struct Mfma16<dlc_bf16_tag> {
    static __device__ __forceinline__ f32x4_t run(u32x4_t a, u32x4_t b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                       __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ float to_f32(unsigned short h) { return dlc_bf16_bits_to_f32(h); }
};
template <> struct Mfma16<dlc_f16_tag> {
    static __device__ __forceinline__ f32x4_t run(u32x4_t a, u32x4_t b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a),
                                                      __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ float to_f32(unsigned short h) { return dlc_f16_bits_to_f32(h); }
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

// LDS image of an operand tile: 256 rows x 128 B, 16-B chunk `ch` of row `r`
// lives in slot ch ^ f(r).  f is chosen so that each ds_read_b128 lane group of
// a fragment read covers 16 distinct slots of the 256-B bank row:
//   A (database, permuted fragment rows): f = bit1(r) | bits5:4(r) << 1
//   B (queries, consecutive fragment rows): f = bits3:1(r)
__device__ __forceinline__ int swz_a(int r) { return ((r >> 1) & 1) | (((r >> 4) & 3) << 1); }
__device__ __forceinline__ int swz_b(int r) { return (r >> 1) & 7; }

struct GemmArgs {
    const char* Q;      // [q, ldq] elements of 2 bytes
    const char* DB;     // [n, lddb]
    long long ldq_b;    // row strides in BYTES
    long long lddb_b;
    int q;
    long long n;
    int nk;             // d / 64
    float* gmax;        // [q, ldg]
    long long ldg;
    float* tmax;        // [q, ldt]
    long long ldt;
    long long ng;       // ceil(n/16)
    long long nh;       // ceil(n/128)
    float* S;           // dense mode: [q, lds]
    long long lds;
    int stagger_mult;   // first-round start stagger: phase * mult * DLC_STAGGER_SLEEP * 64 cycles
    int stagger_phases;
    int nqb;            // query blocks of 256 (grid mapping below)
    long long ntiles;   // database tiles of 256 rows
    int nsplit;         // split-K: chunks of kchunk K tiles, one workgroup each (1 = whole K in one)
    int kchunk;
    float* P;           // split-K partial scores [nsplit][q][ldp], ldp = ntiles * 256
    long long ldp;
};

constexpr int GEMM_GROUPS = 0;   // epilogue: group / half-tile maxima (top-k path)
constexpr int GEMM_DENSE = 1;    //           the score tile itself
constexpr int GEMM_PARTIAL = 2;  //           this K chunk's partial score tile (split-K)

// ---- LDS image (128 KiB): a ring of A_STAGES = 2 K tiles of the database operand (A, streamed
// from HBM; a 3-deep / 160 KiB ring measured the same) followed by a ring of 2 K tiles of the
// query operand (B, re-read from L2).  A K tile of an
// operand is two 16-KiB half tiles of 128 rows x 128 B:
//   A half h, row r = wr*64 + rr   <->  tile database row  wr*128 + h*64 + rr
//   B half h, row r = wc*32 + rr   <->  query row          wc*64  + h*32 + rr
// A wave computes its 128 x 64 block as four 64 x 32 quadrants (A half, B half),
// each in two 32-wide k-slices: 8 mini-phases of 8 MFMAs per K tile.
constexpr int HALF_BYTES = 128 * 128;

// Four LDS-DMA wave-instructions (4 x 1 KiB: 32 rows of one half tile).  Issued from inline asm
// so that hipcc does not count them: it would otherwise put s_waitcnt vmcnt(0) in front of every
// ds_read and serialise the pipeline.  M0 carries the LDS destination (wave-uniform); saved and
// restored because the compiler owns it.  s_nop 4 covers an SGPR operand freshly written by a VALU.
// Cache policy of the DATABASE stream's loads.  The rows are read exactly once, at 4 TB/s through 8 L2s of 4 MiB,
// while the 2 MiB query block is re-read by every workgroup: with the default policy the stream keeps pushing
// query lines out.  `nt` (non-temporal) marks the stream's lines for early replacement: 2.06 -> 1.86 ms per
// 1 M-row launch, A/B/A/B on one device (scripts/exp_policy.sh; sc1 / sc0 sc1 made no difference).
#ifndef DLC_A_CACHE_POLICY
#define DLC_A_CACHE_POLICY " nt"
#endif
#define DLC_DMA4_BODY(POLICY)                                \
    asm volatile(                                            \
        "s_nop 4\n\t"                                        \
        "s_mov_b32 %0, m0\n\t"                               \
        "s_mov_b32 m0, %6\n\t"                               \
        "s_nop 0\n\t"                                        \
        "global_load_lds_dwordx4 %1, %5" POLICY "\n\t"       \
        "s_add_u32 m0, %6, 0x400\n\t"                        \
        "s_nop 0\n\t"                                        \
        "global_load_lds_dwordx4 %2, %5" POLICY "\n\t"       \
        "s_add_u32 m0, %6, 0x800\n\t"                        \
        "s_nop 0\n\t"                                        \
        "global_load_lds_dwordx4 %3, %5" POLICY "\n\t"       \
        "s_add_u32 m0, %6, 0xc00\n\t"                        \
        "s_nop 0\n\t"                                        \
        "global_load_lds_dwordx4 %4, %5" POLICY "\n\t"       \
        "s_mov_b32 m0, %0"                                   \
        : "=&s"(keep)                                        \
        : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(sbase), "s"(lds0) \
        : "memory", "scc")
__device__ __forceinline__ void dma4(const unsigned (&voff)[4], const char* sbase, unsigned lds0) {
    unsigned keep;
    DLC_DMA4_BODY("");
}
// the database operand's form: streamed once, so its cache policy is a separate knob
__device__ __forceinline__ void dma4_stream(const unsigned (&voff)[4], const char* sbase, unsigned lds0) {
    unsigned keep;
    DLC_DMA4_BODY(DLC_A_CACHE_POLICY);
}

#define DLC_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define DLC_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
__device__ __forceinline__ void wg_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// MASKQ: instantiation for a last query block with fewer than 193 queries -- waves whose 64-query
// column block lies entirely past q skip their fragment reads and MFMAs (they still stage and
// synchronise), so 5..192 queries do not pay, at the power cap, for 256 queries' matrix work.
template <typename Tag, int MODE, bool MASKQ>
__global__ __launch_bounds__(NTHREADS, 2) void score_gemm_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2;   // database half (128 rows)
    const int wc = wid & 3;    // query block (64 queries)
    // Workgroup id -> (tile, query block).  Ids go round-robin to the 8 XCDs, each with its own
    // L2: within a run of 8*nqb ids, id j works on tile 8g + (j & 7) for query block j >> 3, so the
    // nqb workgroups that stream the same database tile sit on ONE XCD, 8 ids apart in dispatch
    // order -- the tile comes from HBM once and the others hit that XCD's L2.  (A tile-major grid
    // re-read the whole database per query block: Q = 512 cost exactly 2x Q = 256.)
    // Split-K (MODE == GEMM_PARTIAL; few tiles, long rows): the K chunk is the fastest index.
    const unsigned chunk = MODE == GEMM_PARTIAL ? blockIdx.x % (unsigned)p.nsplit : 0u;
    const unsigned wg = MODE == GEMM_PARTIAL ? blockIdx.x / (unsigned)p.nsplit : blockIdx.x;
    const unsigned run = 8u * (unsigned)p.nqb;
    const int j_ = (int)(wg % run);
    const long long tile = (long long)(wg / run) * 8 + (j_ & 7);
    const int qblk = j_ >> 3;
    if (tile >= p.ntiles) return;      // padding of the last run (before any barrier)
    const unsigned lds_base = (unsigned)(unsigned long long)(lptr_t)smem;

    // ---- DMA roles: waves 0-3 stream the database (A) halves from HBM, waves 4-7 the query
    // (B) halves from L2.  vmcnt counts per wave and in order, so with one stream per wave the
    // depth of the A prefetch does not depend on the B traffic.
    const bool is_a = wid < 4;
    const int ridx = wid & 3;                               // this wave stages rows 32*ridx .. +31 of a half
    const int k_begin = MODE == GEMM_PARTIAL ? (int)chunk * p.kchunk : 0;
    const char* a_base = p.DB + tile * BM * p.lddb_b + (long long)k_begin * 128;
    const char* b_base = p.Q + (long long)qblk * BNQ * p.ldq_b + (long long)k_begin * 128;
    unsigned voff[2][4];                                    // [half][dma]: byte offset of this lane's 16 B
    {
        const int slot = lane & 7;
        const long long arows = p.n - tile * BM;            // valid rows in this tile (>= 1)
        const long long brows = (long long)p.q - (long long)qblk * BNQ;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 32 * ridx + 8 * j + (lane >> 3);  // row inside the half
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                long long ar = (r >> 6) * 128 + h * 64 + (r & 63);
                if (ar > arows - 1) ar = arows - 1;
                long long br = (r >> 5) * 64 + h * 32 + (r & 31);
                if (br > brows - 1) br = brows - 1;
                voff[h][j] = is_a ? (unsigned)(ar * p.lddb_b + ((slot ^ swz_a(r)) << 4))
                                  : (unsigned)(br * p.ldq_b + ((slot ^ swz_b(r)) << 4));
            }
        }
    }
    const unsigned lds_stage = lds_base + (unsigned)(32 * ridx) * 128;   // this wave's rows of a half
    const int nk = MODE == GEMM_PARTIAL ? min(p.kchunk, p.nk - k_begin) : p.nk;   // K tiles of this workgroup

    // ---- fragment read offsets (bytes inside a half)
    const int i = lane & 15;
    const int kq = lane >> 4;
    const int fa = ((i >> 1) & 1) | ((i >> 2) << 1);
    const int fb = (i >> 1) & 7;
    typedef const __attribute__((address_space(3))) u32x4_t* lds_u4p;
    typedef const __attribute__((address_space(3))) char* lds_cp;
    const lds_cp lbase = (lds_cp)(lptr_t)smem;
    const unsigned rdA0_l = (wr * 64 + 16 * (i >> 2) + (i & 3)) * 128 + (((0 + kq) ^ fa) << 4);   // + tt*512
    const unsigned rdA1_l = (wr * 64 + 16 * (i >> 2) + (i & 3)) * 128 + (((4 + kq) ^ fa) << 4);
    const unsigned rdB0_l = B_RING + (wc * 32 + i) * 128 + (((0 + kq) ^ fb) << 4);                // + c*2048
    const unsigned rdB1_l = B_RING + (wc * 32 + i) * 128 + (((4 + kq) ^ fb) << 4);
    // ring positions of the CURRENT K tile (bytes) in the A ring (A_STAGES tiles) and the B ring (2)
    unsigned aoff = 0, boff = 0;
    unsigned rdA0 = rdA0_l, rdA1 = rdA1_l, rdB0 = rdB0_l, rdB1 = rdB1_l;

    f32x4_t acc[8][4];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[t][c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // Fragment registers: one k-slice (32 wide) of one half: A 4 tiles, B 2 tiles.
    u32x4_t faX[4], faY[4], fbX[2], fbY[2];

    const bool active = !MASKQ || (qblk * BNQ + wc * 64 < p.q);
#define DLC_READ_A(DST, RD, OFF)                      \
    if (!MASKQ || active) _Pragma("unroll") for (int tt = 0; tt < 4; ++tt)  \
        DST[tt] = *(lds_u4p)(lbase + (RD) + (OFF) + tt * 512)
#define DLC_READ_B(DST, RD, OFF)                      \
    if (!MASKQ || active) _Pragma("unroll") for (int c = 0; c < 2; ++c)     \
        DST[c] = *(lds_u4p)(lbase + (RD) + (OFF) + c * 2048)
#define DLC_MFMA(FA, FB, AH, BH)                                                                     \
    if (!MASKQ || active) do {                                                                       \
        __builtin_amdgcn_s_setprio(1);                                                               \
        _Pragma("unroll") for (int tt = 0; tt < 4; ++tt) _Pragma("unroll") for (int c = 0; c < 2; ++c) \
            acc[(AH) * 4 + tt][(BH) * 2 + c] =                                                       \
                Mfma16<Tag>::run(FA[tt], FB[c], acc[(AH) * 4 + tt][(BH) * 2 + c]);                   \
        __builtin_amdgcn_s_setprio(0);                                                               \
    } while (0)
#define DLC_RELEASE()    \
    DLC_WAIT_LGKM0();    \
    wg_barrier()
    // DMA of half H of K tile t2 into ring position POS (K offset clamped past the end: the
    // redundant DMA lands in a dead half and keeps the vmcnt bookkeeping uniform)
#define DLC_ISSUE_A(POS, H, t2)                                                                      \
    do {                                                                                             \
        if (is_a) {                                                                                  \
            int kk_ = (t2) < nk ? (t2) : nk - 1;                                                     \
            dma4_stream(voff[H], a_base + (long long)kk_ * 128, lds_stage + (POS) + (H) * HALF_BYTES); \
        }                                                                                            \
    } while (0)
    // wave 4 + j stages exactly the queries of column block j: nobody reads them when that block is idle
    const bool b_on = !MASKQ || (qblk * BNQ + ridx * 64 < p.q);
#define DLC_ISSUE_B(POS, H, t2)                                                                      \
    do {                                                                                             \
        if (!is_a && b_on) {                                                                         \
            int kk_ = (t2) < nk ? (t2) : nk - 1;                                                     \
            dma4(voff[H], b_base + (long long)kk_ * 128, lds_stage + B_RING + (POS) + (H) * HALF_BYTES); \
        }                                                                                            \
    } while (0)

    // Workgroups of one round would otherwise walk K in lockstep, all CUs touching the same
    // 128-byte column of their rows at the same time; a small start stagger of the FIRST round
    // (later rounds inherit it) spreads them over K without changing any result.
    if (MODE != GEMM_PARTIAL && wg < 256) {
        const int steps = (int)((tile >> 3) % p.stagger_phases) * p.stagger_mult;
        for (int s_ = 0; s_ < steps; ++s_) __builtin_amdgcn_s_sleep(DLC_STAGGER_SLEEP);
    }
    // ---- prologue: A tiles 0..A_STAGES-1 and B tiles 0,1 issued (per-tile order A1,A0 / B0,B1, as the
    // steady state issues them); then wait for tile 0.
#pragma unroll
    for (int s_ = 0; s_ < A_STAGES; ++s_) {
        DLC_ISSUE_A(s_ * A_TILE, 1, s_);
        DLC_ISSUE_A(s_ * A_TILE, 0, s_);
    }
    DLC_ISSUE_B(0 * B_TILE, 0, 0);
    DLC_ISSUE_B(0 * B_TILE, 1, 0);
    DLC_ISSUE_B(1 * B_TILE, 0, 1);
    DLC_ISSUE_B(1 * B_TILE, 1, 1);
    if (is_a) { if constexpr (A_STAGES == 3) DLC_WAIT_VMCNT(16); else DLC_WAIT_VMCNT(8); }
    else DLC_WAIT_VMCNT(8);                                 // K tile 0 landed (this wave's share)
    wg_barrier();
    DLC_READ_A(faX, rdA0, 0);
    DLC_READ_B(fbX, rdB0, 0);

    // One K tile = 8 mini-phases of 8 MFMAs, k-slice outer, quadrants in snake order so that
    // exactly one operand changes per step and every LDS fragment is read once:
    //   m1 (A0,B0,k0) m2 (A0,B1,k0) m3 (A1,B1,k0) m4 (A1,B0,k0)
    //   m5 (A1,B0,k1) m6 (A1,B1,k1) m7 (A0,B1,k1) m8 (A0,B0,k1)
    // Mini-phase m issues the reads of m+1 first, then its own MFMAs.  A half of the current
    // tile is dead once its k1 slice has been read (A1 after m3, B0 after m4, B1 after m5, A0
    // after m6): a barrier there, then the DMA that refills it -- A halves with K tile t+A_STAGES, B
    // halves with t+2.  At the end of m6 each wave waits for its own stream: an A wave leaves
    // A1(t+2) in flight (vmcnt 4; three halves / vmcnt 12 with a 3-deep ring), a B wave B0(t+2), B1(t+2) (vmcnt 8); K
    // tile t+1, whose first slices are read in m7 / m8, has then landed.
    for (int t = 0; t < nk; ++t) {
        DLC_READ_B(fbY, rdB0, HALF_BYTES);
        DLC_MFMA(faX, fbX, 0, 0);                                  // m1
        DLC_READ_A(faY, rdA0, HALF_BYTES);
        DLC_MFMA(faX, fbY, 0, 1);                                  // m2
        DLC_READ_A(faX, rdA1, HALF_BYTES);
        DLC_MFMA(faY, fbY, 1, 1);                                  // m3
        DLC_RELEASE();                                             // A1 read by everyone
        DLC_READ_B(fbY, rdB1, 0);
        DLC_ISSUE_A(aoff, 1, t + A_STAGES);
        DLC_MFMA(faY, fbX, 1, 0);                                  // m4
        DLC_RELEASE();                                             // B0
        DLC_READ_B(fbX, rdB1, HALF_BYTES);
        DLC_ISSUE_B(boff, 0, t + 2);
        DLC_MFMA(faX, fbY, 1, 0);                                  // m5
        DLC_RELEASE();                                             // B1
        DLC_READ_A(faY, rdA1, 0);
        DLC_ISSUE_B(boff, 1, t + 2);
        DLC_MFMA(faX, fbX, 1, 1);                                  // m6
        if (is_a) { if constexpr (A_STAGES == 3) DLC_WAIT_VMCNT(12); else DLC_WAIT_VMCNT(4); }
        else DLC_WAIT_VMCNT(8);                                    // K tile t+1 landed (this wave's share)
        DLC_RELEASE();                                             // A0; and t+1 visible to all
        DLC_ISSUE_A(aoff, 0, t + A_STAGES);
        aoff = aoff == (A_STAGES - 1) * A_TILE ? 0u : aoff + A_TILE;   // ring positions of K tile t+1
        boff ^= B_TILE;
        rdA0 = rdA0_l + aoff; rdA1 = rdA1_l + aoff; rdB0 = rdB0_l + boff; rdB1 = rdB1_l + boff;
        DLC_READ_A(faX, rdA0, 0);
        DLC_MFMA(faY, fbX, 0, 1);                                  // m7
        DLC_READ_B(fbX, rdB0, 0);
        DLC_MFMA(faY, fbY, 0, 0);                                  // m8
    }
    DLC_WAIT_VMCNT(0);   // the clamped tail DMAs must not outlive the workgroup's LDS
    DLC_WAIT_LGKM0();
#undef DLC_ISSUE_A
#undef DLC_ISSUE_B
#undef DLC_READ_A
#undef DLC_READ_B
#undef DLC_MFMA
#undef DLC_RELEASE

    // ---- epilogue
    const int lg = lane >> 4;
    if constexpr (MODE == GEMM_PARTIAL) {
        float* part = p.P + (long long)chunk * p.q * p.ldp;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int qidx = qblk * BNQ + wc * 64 + c * 16 + i;
            if (qidx >= p.q) continue;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const long long row0 = tile * BM + wr * 128 + (t >> 2) * 64 + 16 * lg + 4 * (t & 3);
                *(f32x4_t*)(part + (long long)qidx * p.ldp + row0) = acc[t][c];   // ldp covers whole tiles
            }
        }
    } else if constexpr (MODE == GEMM_DENSE) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int qidx = qblk * BNQ + wc * 64 + c * 16 + i;
            if (qidx >= p.q) continue;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const long long row0 = tile * BM + wr * 128 + (t >> 2) * 64 + 16 * lg + 4 * (t & 3);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (row0 + r < p.n) p.S[(long long)qidx * p.lds + row0 + r] = acc[t][c][r];
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int qidx = qblk * BNQ + wc * 64 + c * 16 + i;
            float hm[2];
#pragma unroll
            for (int th = 0; th < 2; ++th) {
                // this lane's 16 rows of the half are two groups of 8: MFMA tiles {0,1} and {2,3}
                float m0 = acc[th * 4][c][0], m1 = acc[th * 4 + 2][c][0];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        m0 = fmaxf(m0, acc[th * 4 + tt][c][r]);
                        m1 = fmaxf(m1, acc[th * 4 + 2 + tt][c][r]);
                    }
                hm[th] = fmaxf(m0, m1);
                const long long g = tile * (BM / GROUP) + wr * 16 + th * 8 + lg * 2;
                if (qidx < p.q) {
                    float* dst = p.gmax + (long long)qidx * p.ldg + g;
                    if (g + 1 < p.ng) *(float2*)dst = make_float2(m0, m1);
                    else if (g < p.ng) dst[0] = m0;
                }
            }
            float h = fmaxf(hm[0], hm[1]);
            h = fmaxf(h, __shfl_xor(h, 16));
            h = fmaxf(h, __shfl_xor(h, 32));
            const long long ht = tile * 2 + wr;
            if (lg == 0 && qidx < p.q && ht < p.nh) p.tmax[(long long)qidx * p.ldt + ht] = h;
        }
    }
}

// ---------------------------------------------------------------------------
// score_gemv_kernel: the score pass for a HANDFUL of queries (q <= 4: a single camera frame).
// The 256-query MFMA tile would do 64-256x the needed matrix work -- at the power cap, where
// that work is what sets the time -- so few queries take a bandwidth kernel instead:
// queries resident in LDS, every wave streams 8 database rows at a time with 16-byte loads (8 KiB
// in flight per wave), v_dot2 accumulation in fp32, wave reduction, the same gmax / tmax outputs.
// One workgroup (4 waves) per half tile of 128 rows; wave w takes its groups 4w .. 4w+3.
// ---------------------------------------------------------------------------
template <typename Tag> struct Dot2;
template <> struct Dot2<dlc_bf16_tag> {
    typedef __attribute__((ext_vector_type(2))) __bf16 v2_t;
    static __device__ __forceinline__ float run(unsigned a, unsigned b, float c) {
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2_t, a), __builtin_bit_cast(v2_t, b), c, false);
    }
};
template <> struct Dot2<dlc_f16_tag> {
    typedef __attribute__((ext_vector_type(2))) _Float16 v2_t;
    static __device__ __forceinline__ float run(unsigned a, unsigned b, float c) {
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(v2_t, a), __builtin_bit_cast(v2_t, b), c, false);
    }
};

constexpr int GEMV_MAX_Q = 4;
constexpr int GEMV_MAX_LDS = 64 * 1024;

template <typename Tag, int QB>
__global__ __launch_bounds__(256) void score_gemv_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [QB][d] stored queries
    __shared__ float wmax[4][QB];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const long long half = blockIdx.x;
    const int d_bytes = p.nk * 128;
    for (int off = tid * 16; off < QB * d_bytes; off += 256 * 16) {
        const int qi = off / d_bytes, within = off - qi * d_bytes;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (qi < p.q) v = *(const uint4*)(p.Q + (long long)qi * p.ldq_b + within);
        *(uint4*)(smem + off) = v;
    }
    __syncthreads();
    float hmax[QB];
#pragma unroll
    for (int qq = 0; qq < QB; ++qq) hmax[qq] = -INFINITY;
    for (int gi = 0; gi < 4; ++gi) {
        const long long g = half * GROUPS_PER_HALF + w * 4 + gi;
        const char* rows[GROUP];
#pragma unroll
        for (int r = 0; r < GROUP; ++r) {
            long long rr = g * GROUP + r;
            if (rr > p.n - 1) rr = p.n - 1;                          // duplicates of the last row, as the GEMM's clamp
            rows[r] = p.DB + rr * p.lddb_b;
        }
        float acc[QB][GROUP];
#pragma unroll
        for (int qq = 0; qq < QB; ++qq)
#pragma unroll
            for (int r = 0; r < GROUP; ++r) acc[qq][r] = 0.f;
        for (int c = lane * 16; c < d_bytes; c += 1024) {
            uint4 rv[GROUP];
#pragma unroll
            for (int r = 0; r < GROUP; ++r) rv[r] = *(const uint4*)(rows[r] + c);
#pragma unroll
            for (int qq = 0; qq < QB; ++qq) {
                const uint4 qv = *(const uint4*)(smem + qq * d_bytes + c);
#pragma unroll
                for (int r = 0; r < GROUP; ++r) {
                    float a = acc[qq][r];
                    a = Dot2<Tag>::run(rv[r].x, qv.x, a);
                    a = Dot2<Tag>::run(rv[r].y, qv.y, a);
                    a = Dot2<Tag>::run(rv[r].z, qv.z, a);
                    a = Dot2<Tag>::run(rv[r].w, qv.w, a);
                    acc[qq][r] = a;
                }
            }
        }
#pragma unroll
        for (int qq = 0; qq < QB; ++qq) {
            float m = -INFINITY;
#pragma unroll
            for (int r = 0; r < GROUP; ++r) {
                float v = acc[qq][r];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
                m = fmaxf(m, v);
            }
            hmax[qq] = fmaxf(hmax[qq], m);
            if (lane == 0 && qq < p.q && g < p.ng) p.gmax[(long long)qq * p.ldg + g] = m;
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int qq = 0; qq < QB; ++qq) wmax[w][qq] = hmax[qq];
    }
    __syncthreads();
    if (tid < QB && tid < p.q && half < p.nh)
        p.tmax[(long long)tid * p.ldt + half] = fmaxf(fmaxf(wmax[0][tid], wmax[1][tid]), fmaxf(wmax[2][tid], wmax[3][tid]));
}

// ---------------------------------------------------------------------------
// selection on packed 64-bit keys: (monotone score key << 32) | ~id32, so "larger key" ==
// "higher score, then lower id".  key 0 = empty.
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned f32_key(float x) {   // monotone: larger float -> larger key
    unsigned u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ unsigned long long pack_key(float score, unsigned id) {
    return ((unsigned long long)f32_key(score) << 32) | (unsigned long long)(~id);
}
__device__ __forceinline__ unsigned key_id(unsigned long long key) { return ~(unsigned)key; }

// wave-wide unsigned max through DPP (row_shr 1,2,4,8 + row_bcast 15 / 31), result in every lane
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#define DLC_DPP_MAX(ctrl, rmask) \
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rmask, 0xf, false))
    DLC_DPP_MAX(0x111, 0xf);
    DLC_DPP_MAX(0x112, 0xf);
    DLC_DPP_MAX(0x114, 0xf);
    DLC_DPP_MAX(0x118, 0xf);
    DLC_DPP_MAX(0x142, 0xa);
    DLC_DPP_MAX(0x143, 0xc);
#undef DLC_DPP_MAX
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

constexpr int FIN_THREADS = 512;            // merge workgroup; finish workgroup of the stand-alone variant
// finish kernel LDS carve (bytes) for kg selected groups: keys | re-scored rows | two index lists
__host__ __device__ inline size_t fin_lds_fixed(int kg) { return (size_t)kg * (GROUPS_PER_HALF * 8 + GROUP * 4 + 8); }

// Rank-by-counting on unique keys in LDS: out[rank] = element index, for rank < k.
// Every thread walks the keys in the same order (LDS broadcast reads).
__device__ __forceinline__ void rank_select(const unsigned long long* keys, int m, int k, int* out) {
    for (int e = threadIdx.x; e < m; e += blockDim.x) {
        const unsigned long long ke = keys[e];
        if (ke == 0ull) continue;
        int rank = 0;
#pragma unroll 8
        for (int j = 0; j < m; ++j) rank += keys[j] > ke ? 1 : 0;
        if (rank < k) out[rank] = e;
    }
}

// One workgroup per query: half-tile selection -> group selection -> exact fp32 re-score of
// the selected groups' rows -> final top-k.
// THREADS / RS_UNROLL pick the footprint: (512, 4) is the fastest stand-alone form (246 VGPRs);
// (256, 1) needs ~70 VGPRs and a few KiB of LDS so that its workgroups can share a CU with a
// resident score-GEMM workgroup (128 KiB LDS, 2 x 198 VGPRs per SIMD) when the two run on
// different streams.
// MODE: FIN_FUSED  all of it;
//       FIN_GROUPS stops after the group selection and writes the kg selected groups of every query
//                  (grp_ids, -1 = none; grp_max, their maxima, in rank order);
//       FIN_RESCORE starts from such a list.  With parts > 0 it first drops every own group that
//                  cannot be among the kg best groups of the WHOLE database: all_max holds the
//                  lists of all `parts` shards ([parts, q, kg], an all-gather of grp_max), and a
//                  group with kg or more strictly larger maxima anywhere is out.  Each shard then
//                  re-scores ~kg/parts groups per query instead of kg.
enum { FIN_FUSED = 0, FIN_GROUPS = 1, FIN_RESCORE = 2 };
struct FinishExtra {
    int* grp_ids;             // [q, kg]
    float* grp_max;           // [q, kg]
    const float* all_max;     // [parts, q, kg] or null
    int parts;
    long long nq;
    int gparts;               // FIN_RESCORE: > 1 = grid.y workgroups per query, workgroup y re-scores the listed
                              // groups e with e % gparts == y and writes its top-k at [y][q][k]
    const float* dense_S;     // small-database plan: the score matrix itself [q, ld_s] is in the workspace, so the
    long long ld_s;           // selected groups' scores are READ from it instead of re-computed from gathered rows
};

// The 256-thread forms are meant to sit beside a resident score-GEMM workgroup (2 x 200 VGPRs per
// SIMD): 5 waves per SIMD caps them at 96 VGPRs.  (Left to itself the compiler chose 116-118 for two
// of them -- more loads in flight, but no room beside the GEMM.)
template <typename Tag, int THREADS, int RS_UNROLL, int MODE>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(THREADS == 256 ? 5 : 1)))
void finish_topk_kernel(
    float* __restrict__ tmax, long long ldt, int nh, int tv_in_lds, const float* __restrict__ gmax, long long ldg,
    long long ng, int kg, const char* __restrict__ Q, long long ldq_b, const char* __restrict__ DB, long long lddb_b,
    long long n, int d, int k, long long row_offset, float* __restrict__ out_s, long long* __restrict__ out_i,
    FinishExtra x) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    constexpr int FIN_WAVES = THREADS / 64;
    constexpr int FIN_THREADS = THREADS;      // shadows the namespace constant inside this kernel
    constexpr int GPH = GROUPS_PER_HALF;
    unsigned long long* ckey = (unsigned long long*)dsm;                       // [kg * GPH]
    float* cval = (float*)(dsm + (size_t)kg * GPH * 8);                        // [kg * GROUP] re-scored rows
    int* sel = (int*)(dsm + (size_t)kg * (GPH * 8 + GROUP * 4));               // [kg] half tiles; later the winners
    int* sel2 = sel + kg;                                                      // [kg] selected groups
    const int qi = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // [nh] half-tile maxima of this query: staged in LDS, or consumed in place in the workspace
    // row (which the next GEMM rewrites anyway) when LDS is to be kept small or the shard is huge
    float* tv = tv_in_lds ? (float*)(dsm + fin_lds_fixed(kg)) : tmax + (long long)qi * ldt;
    if (tv_in_lds)
        for (int e = tid; e < nh; e += FIN_THREADS) tv[e] = tmax[(long long)qi * ldt + e];
    for (int e = tid; e < kg; e += FIN_THREADS) { sel[e] = -1; sel2[e] = -1; }
    __syncthreads();
    int kg2;
    if constexpr (MODE != FIN_RESCORE) {
    // ---- level 1: the kt half tiles with the largest maximum (ties -> lower tile).
    // Each wave extracts the kt best of its slice by repeated wave arg-max (DPP, no barrier);
    // the FIN_WAVES * kt survivors are ranked together.
    const int kt = min(kg, nh);
    {
        const int chunk = (nh + FIN_WAVES - 1) / FIN_WAVES;
        const int lo_e = w * chunk, hi_e = min(nh, lo_e + chunk);
        unsigned bh = 0, bl = 0;               // this lane's best key (hi, lo); 0,0 = none
        constexpr int LV = 16;                 // a lane's values live in registers when its slice has <= LV of them
        // (not in the small-footprint fused form, which has to stay under ~110 VGPRs to sit beside a GEMM workgroup)
        if ((THREADS == 512 || MODE == FIN_GROUPS) && chunk <= 64 * LV) {
            // (up to 8 waves x 1024 half tiles = 1 M rows per shard: the benchmark; an owner's rescan is then
            // 16 register compares instead of 16 dependent LDS reads, 24 times per wave)
            unsigned vk[LV];                   // score keys; 0 = none / retired
#pragma unroll
            for (int j = 0; j < LV; ++j) {
                const int e = lo_e + lane + 64 * j;
                const float v = e < hi_e ? tv[e] : -INFINITY;
                vk[j] = v == -INFINITY ? 0u : f32_key(v);
            }
            auto rescan_regs = [&]() {
                bh = 0; bl = 0;
#pragma unroll
                for (int j = 0; j < LV; ++j) {             // ascending e: the first maximum has the lowest index
                    const unsigned l = ~(unsigned)(lo_e + lane + 64 * j);
                    if (vk[j] > bh) { bh = vk[j]; bl = l; }
                }
            };
            rescan_regs();
            for (int it = 0; it < kt; ++it) {
                const unsigned mh = wave_max_u32(bh);
                const unsigned ml = wave_max_u32(bh == mh ? bl : 0u);
                if (lane == 0) ckey[w * kt + it] = mh == 0u ? 0ull : (((unsigned long long)mh << 32) | ml);
                if (mh != 0u && bh == mh && bl == ml) {  // the owner retires it and finds its next best
                    const int jr = (int)((~ml) - (unsigned)(lo_e + lane)) >> 6;
#pragma unroll
                    for (int j = 0; j < LV; ++j)
                        if (j == jr) vk[j] = 0u;
                    rescan_regs();
                }
            }
        } else {
        auto rescan = [&]() {
            bh = 0; bl = 0;
            for (int e = lo_e + lane; e < hi_e; e += 64) {
                const float v = tv[e];
                if (v == -INFINITY) continue;
                const unsigned h = f32_key(v), l = ~(unsigned)e;
                if (h > bh || (h == bh && l > bl)) { bh = h; bl = l; }
            }
        };
        rescan();
        for (int it = 0; it < kt; ++it) {
            const unsigned mh = wave_max_u32(bh);
            const unsigned ml = wave_max_u32(bh == mh ? bl : 0u);
            if (lane == 0) ckey[w * kt + it] = mh == 0u ? 0ull : (((unsigned long long)mh << 32) | ml);
            if (mh != 0u && bh == mh && bl == ml) {      // the owner retires it and finds its next best
                tv[~ml] = -INFINITY;
                rescan();
            }
        }
        }
    }
    __syncthreads();
    rank_select(ckey, FIN_WAVES * kt, kt, sel);          // sel[rank] = slot in ckey
    __syncthreads();
    if (tid < kt) { const int c = sel[tid]; sel[tid] = c < 0 ? -1 : (int)key_id(ckey[c]); }   // -> half-tile index
    __syncthreads();

    // ---- level 2: among their groups, the kg2 groups with the largest maximum
    const int m2 = kt * GPH;
    for (int e = tid; e < m2; e += FIN_THREADS) {
        const int ht = sel[e / GPH];
        const long long g = (long long)ht * GPH + (e % GPH);
        ckey[e] = (ht >= 0 && g < ng) ? pack_key(gmax[(long long)qi * ldg + g], (unsigned)g) : 0ull;
    }
    __syncthreads();
    kg2 = min(kg, m2);
    rank_select(ckey, m2, kg2, sel2);
    __syncthreads();
    if constexpr (MODE == FIN_GROUPS) {
        for (int e = tid; e < kg; e += FIN_THREADS) {
            const int c = e < kg2 ? sel2[e] : -1;
            x.grp_ids[(long long)qi * kg + e] = c < 0 ? -1 : (int)key_id(ckey[c]);
            x.grp_max[(long long)qi * kg + e] = c < 0 ? -INFINITY : gmax[(long long)qi * ldg + key_id(ckey[c])];
        }
        return;
    }
    if (tid < kg2) { const int c = sel2[tid]; sel2[tid] = c < 0 ? -1 : (int)key_id(ckey[c]); }   // -> group index
    __syncthreads();
    } else {
        // ---- start from a group list; optionally filter it against the other shards' maxima
        kg2 = kg;
        for (int e = tid; e < kg; e += FIN_THREADS) {
            int g = x.grp_ids[(long long)qi * kg + e];
            if (g >= 0 && x.parts > 0) {
                const float v = x.grp_max[(long long)qi * kg + e];
                int greater = 0;
                for (int pp = 0; pp < x.parts; ++pp) {
                    const float* av = x.all_max + ((long long)pp * x.nq + qi) * kg;
                    for (int j = 0; j < kg; ++j) greater += av[j] > v ? 1 : 0;
                }
                if (greater >= kg) g = -1;
            }
            if (x.gparts > 1 && e % x.gparts != (int)blockIdx.y) g = -1;
            sel2[e] = g;
        }
        __syncthreads();
    }

    // ---- re-score: wave w takes groups w, w+8, ...; GROUP exact fp32 dot products each
    const char* qrow = Q + (long long)qi * ldq_b;
    for (int s = w; s < kg2; s += FIN_WAVES) {
        const int g = sel2[s];
        if (g < 0) {
            if (lane < GROUP) cval[s * GROUP + lane] = -INFINITY;
            continue;
        }
        const long long row0 = (long long)g * GROUP;
        if (x.dense_S) {                                  // the group's 8 scores are already there
            if (lane < GROUP)
                cval[s * GROUP + lane] = (row0 + lane < n) ? x.dense_S[(long long)qi * x.ld_s + row0 + lane] : -INFINITY;
            continue;
        }
        const char* rows[GROUP];
#pragma unroll
        for (int r = 0; r < GROUP; ++r) {
            long long rr = row0 + r;
            if (rr > n - 1) rr = n - 1;
            rows[r] = DB + rr * lddb_b;
        }
        float acc[GROUP];
#pragma unroll
        for (int r = 0; r < GROUP; ++r) acc[r] = 0.f;
        // RS_UNROLL chunks of 512 elements per trip: all their row loads are issued before the
        // first use, so RS_UNROLL * GROUP + RS_UNROLL 16-byte loads are in flight per lane.
        for (int d0 = lane * 8; d0 < d; d0 += 512 * RS_UNROLL) {
            uint4 qv[RS_UNROLL], rv[RS_UNROLL][GROUP];
#pragma unroll
            for (int u = 0; u < RS_UNROLL; ++u) {
                const int dd = d0 + u * 512;
                const bool ok = dd < d;
                qv[u] = ok ? *(const uint4*)(qrow + (long long)dd * 2) : make_uint4(0, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < GROUP; ++r)
                    rv[u][r] = ok ? *(const uint4*)(rows[r] + (long long)dd * 2) : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < RS_UNROLL; ++u) {
                const unsigned qw[4] = {qv[u].x, qv[u].y, qv[u].z, qv[u].w};
                float qf[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    qf[2 * e] = Mfma16<Tag>::to_f32((unsigned short)(qw[e] & 0xffffu));
                    qf[2 * e + 1] = Mfma16<Tag>::to_f32((unsigned short)(qw[e] >> 16));
                }
#pragma unroll
                for (int r = 0; r < GROUP; ++r) {
                    const unsigned w4[4] = {rv[u][r].x, rv[u][r].y, rv[u][r].z, rv[u][r].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[r] = fmaf(qf[2 * e], Mfma16<Tag>::to_f32((unsigned short)(w4[e] & 0xffffu)), acc[r]);
                        acc[r] = fmaf(qf[2 * e + 1], Mfma16<Tag>::to_f32((unsigned short)(w4[e] >> 16)), acc[r]);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < GROUP; ++r) {
            float v = acc[r];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (lane == r) cval[s * GROUP + r] = (row0 + r < n) ? v : -INFINITY;
        }
    }
    const long long oq = (long long)blockIdx.y * x.nq + qi;      // blockIdx.y > 0 only with gparts > 1
    for (int e = tid; e < k; e += FIN_THREADS) {          // defaults for slots past the candidates
        out_s[oq * k + e] = -INFINITY;
        out_i[oq * k + e] = -1;
    }
    for (int e = tid; e < kg; e += FIN_THREADS) sel[e] = -1;
    __syncthreads();

    // ---- final top-k of the kg2*GROUP re-scored rows (id = database row inside the shard)
    const int m3 = kg2 * GROUP;
    for (int e = tid; e < m3; e += FIN_THREADS) {
        const float v = cval[e];
        const int g = sel2[e / GROUP];
        ckey[e] = (g < 0 || v == -INFINITY) ? 0ull : pack_key(v, (unsigned)((long long)g * GROUP + (e % GROUP)));
    }
    __syncthreads();
    rank_select(ckey, m3, k, sel);
    __syncthreads();
    for (int e = tid; e < k; e += FIN_THREADS) {
        const int c = sel[e];
        if (c >= 0) {
            out_s[oq * k + e] = cval[c];
            out_i[oq * k + e] = (long long)key_id(ckey[c]) + row_offset;
        }
    }
}

// Global top-k from [parts, q, k] per-shard results (the all-gather layout); idx < 0 = empty slot.
// Rank-by-counting on (score key, id): ids are unique, so the order is total.
__global__ __launch_bounds__(FIN_THREADS) void merge_topk_kernel(const float* __restrict__ pscores,
                                                                 long long s_stride,
                                                                 const long long* __restrict__ pidx,
                                                                 long long i_stride, int parts, long long nq, int k,
                                                                 float* __restrict__ out_s,
                                                                 long long* __restrict__ out_i) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    const int qi = blockIdx.x, tid = threadIdx.x;
    const int m = parts * k;
    long long* ci = (long long*)dsm;
    unsigned* sk = (unsigned*)(dsm + (size_t)m * 8);
    float* cs = (float*)(dsm + (size_t)m * 12);
    for (int e = tid; e < m; e += FIN_THREADS) {
        const long long o = (long long)qi * k + (e % k);
        const long long id = pidx[(long long)(e / k) * i_stride + o];
        const float s = pscores[(long long)(e / k) * s_stride + o];
        ci[e] = id; cs[e] = s;
        sk[e] = id < 0 ? 0u : f32_key(s);
    }
    for (int e = tid; e < k; e += FIN_THREADS) {
        out_s[(long long)qi * k + e] = -INFINITY;
        out_i[(long long)qi * k + e] = -1;
    }
    __syncthreads();
    for (int e = tid; e < m; e += FIN_THREADS) {
        const unsigned se = sk[e];
        const long long ie = ci[e];
        if (ie < 0) continue;
        int rank = 0;
        for (int j = 0; j < m; ++j) {
            const unsigned sj = sk[j];
            const long long ij = ci[j];
            rank += (ij >= 0 && (sj > se || (sj == se && ij < ie))) ? 1 : 0;
        }
        if (rank < k) {
            out_s[(long long)qi * k + rank] = cs[e];
            out_i[(long long)qi * k + rank] = ie;
        }
    }
}

// Row L2 normalisation into the stored descriptor format.
template <typename Src, typename Tag>
__global__ __launch_bounds__(256) void l2_normalize_kernel(const Src* __restrict__ src, long long lds, int d,
                                                           int center, unsigned short* __restrict__ dst,
                                                           long long ldd) {
    // Rows too long for registers (the 75 008-d flattened SDAV descriptors): one workgroup per row, three walks over it
    // (mean, centred sum of squares, output), the second and third out of L2.  16-byte loads, four of them in flight per
    // lane, packed 16-bit stores -- element-wise 8-byte loads and 2-byte stores ran 1063 x 75 008 doubles at 1.4 TB/s.
    __shared__ double red[8];
    constexpr int VW = 16 / (int)sizeof(Src);            // elements per 16-byte vector: 4 floats / 2 doubles
    typedef Src vec_t __attribute__((ext_vector_type(VW)));
    const long long row = blockIdx.x;
    const Src* x = src + row * lds;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool vec_ok = (((uintptr_t)x) & 15) == 0;       // rows on 16-byte boundaries (else: the scalar walk)
    const int dv = vec_ok ? d / VW * VW : 0;              // elements covered by whole vectors
    auto walk = [&](auto&& f) {                           // f(value, element index) over the row, a fixed order per thread
        int e = tid * VW;
        for (; e + 3 * 256 * VW < dv; e += 4 * 256 * VW) {
            vec_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *(const vec_t*)(x + e + u * 256 * VW);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < VW; ++i) f((double)v[u][i], e + u * 256 * VW + i);
        }
        for (; e < dv; e += 256 * VW) {
            const vec_t v = *(const vec_t*)(x + e);
#pragma unroll
            for (int i = 0; i < VW; ++i) f((double)v[i], e + i);
        }
        for (int t = dv + tid; t < d; t += 256) f((double)x[t], t);
    };
    double mean = 0.0;
    if (center) {
        double s = 0.0;
        walk([&](double v, int) { s += v; });
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) red[w] = s;
        __syncthreads();
        mean = (red[0] + red[1] + red[2] + red[3]) / (double)d;
        __syncthreads();
    }
    double ss = 0.0;
    walk([&](double v, int) { const double c = v - mean; ss += c * c; });
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if (lane == 0) red[4 + w] = ss;
    __syncthreads();
    const double nrm = sqrt(red[4] + red[5] + red[6] + red[7]);
    const double inv = nrm > 0.0 ? 1.0 / nrm : 1.0;
    auto to16 = [&](double v) -> unsigned {
        const float f = (float)((v - mean) * inv);
        if constexpr (__is_same(Tag, dlc_bf16_tag)) return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)f);
        else return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)f);
    };
    unsigned short* o = dst + row * ldd;                  // ldd is a multiple of 64: rows of dst are 128-byte aligned
    // pairs of outputs as one 32-bit store (ldd is even); elements past d are zero padding
    for (int e = tid * 2; e < (int)ldd; e += 512) {
        const unsigned lo = e < d ? to16((double)x[e]) : 0u;
        const unsigned hi = e + 1 < d ? to16((double)x[e + 1]) : 0u;
        *(unsigned*)(o + e) = lo | (hi << 16);
    }
}

// One-pass form for rows that fit in registers: TPR threads per row (64 = one wave per row, no
// barrier; 256 = one workgroup per row), NVEC 16-byte vectors per thread, so a row is read from HBM
// once and written once (the multi-pass kernel above reads it up to three times: mean, norm,
// scale).  Same per-element arithmetic: fp64 statistics, (x - mean) * (1 / norm) in fp64, rounded
// to fp32 and then to the stored type.  HBM-bound: n * (d * sizeof(Src) + ldd * 2) bytes.
template <typename Src, typename Tag, int TPR, int NVEC>
__global__ __launch_bounds__(256) void l2_normalize_regs_kernel(const Src* __restrict__ src, long long lds, long long n,
                                                                int d, int center, unsigned short* __restrict__ dst,
                                                                long long ldd) {
    constexpr int VW = 16 / (int)sizeof(Src);            // elements per 16-byte vector: 4 floats / 2 doubles
    typedef Src vec_t __attribute__((ext_vector_type(VW)));
    __shared__ double red[2][4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int t = TPR == 64 ? lane : tid;                // index inside the row's thread group
    const long long row = TPR == 64 ? (long long)blockIdx.x * 4 + (tid >> 6) : (long long)blockIdx.x;
    if (row >= n) return;                                // wave-uniform (TPR == 64) or block-uniform
    const Src* x = src + row * lds;
    vec_t v[NVEC];
#pragma unroll
    for (int j = 0; j < NVEC; ++j) {
        const int e0 = (j * TPR + t) * VW;
        if (e0 + VW <= d) {
            v[j] = *(const vec_t*)(x + e0);
        } else {
#pragma unroll
            for (int i = 0; i < VW; ++i) v[j][i] = e0 + i < d ? x[e0 + i] : (Src)0;
        }
    }
    auto row_sum = [&](double s, int slot) -> double {
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if constexpr (TPR == 64) return s;
        if (lane == 0) red[slot][tid >> 6] = s;
        __syncthreads();
        return red[slot][0] + red[slot][1] + red[slot][2] + red[slot][3];
    };
    double mean = 0.0;
    if (center) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < NVEC; ++j)
#pragma unroll
            for (int i = 0; i < VW; ++i) s += (double)v[j][i];          // elements past d are zero
        mean = row_sum(s, 0) / (double)d;
    }
    double ss = 0.0;
#pragma unroll
    for (int j = 0; j < NVEC; ++j)
#pragma unroll
        for (int i = 0; i < VW; ++i) {
            const double c = (double)v[j][i] - mean;
            ss += ((j * TPR + t) * VW + i < d) ? c * c : 0.0;
        }
    const double nrm = sqrt(row_sum(ss, 1));
    const double inv = nrm > 0.0 ? 1.0 / nrm : 1.0;
    unsigned short* o = dst + row * ldd;
#pragma unroll
    for (int j = 0; j < NVEC; ++j) {
        const int e0 = (j * TPR + t) * VW;
        if (e0 >= ldd) continue;                         // ldd is a multiple of 64: vectors never straddle it
        unsigned short bits[VW];
#pragma unroll
        for (int i = 0; i < VW; ++i) {
            bits[i] = 0;
            if (e0 + i < d) {
                const float f = (float)(((double)v[j][i] - mean) * inv);
                if constexpr (__is_same(Tag, dlc_bf16_tag)) bits[i] = __builtin_bit_cast(unsigned short, (__bf16)f);
                else bits[i] = __builtin_bit_cast(unsigned short, (_Float16)f);
            }
        }
        if constexpr (VW == 4) {
            *(uint2*)(o + e0) = make_uint2((unsigned)bits[0] | ((unsigned)bits[1] << 16),
                                           (unsigned)bits[2] | ((unsigned)bits[3] << 16));
        } else {
            *(unsigned*)(o + e0) = (unsigned)bits[0] | ((unsigned)bits[1] << 16);
        }
    }
}

// Split-K second pass.  Sums the chunk partials in chunk order (deterministic) and applies the
// epilogue of the one-pass kernel: GROUPS -> gmax / tmax, DENSE -> the score matrix.
// grid (q, ceil(groups / 256)), 256 threads: one thread per group of 8 rows.
// keep_sum: also store the summed scores over chunk 0's slot (thread-private elements, so in place):
// the small-database plan selects its top-k from them.
__global__ __launch_bounds__(256) void splitk_groups_kernel(float* __restrict__ P, long long ldp, int q, int nsplit,
                                                            float* __restrict__ gmax, long long ldg, long long ng,
                                                            float* __restrict__ tmax, long long ldt, long long nh,
                                                            int keep_sum) {
    const int qi = blockIdx.x;
    const long long g = (long long)blockIdx.y * 256 + threadIdx.x;
    float m = -INFINITY;
    if (g < ng) {
        f32x4_t a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        float* dst0 = P + (long long)qi * ldp + g * GROUP;
        const float* src = dst0;
        for (int c = 0; c < nsplit; ++c) {
            a += *(const f32x4_t*)(src);
            b += *(const f32x4_t*)(src + 4);
            src += (long long)q * ldp;
        }
        if (keep_sum && nsplit > 1) {
            *(f32x4_t*)dst0 = a;
            *(f32x4_t*)(dst0 + 4) = b;
        }
        m = fmaxf(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), fmaxf(fmaxf(b[0], b[1]), fmaxf(b[2], b[3])));
        gmax[(long long)qi * ldg + g] = m;
    }
    // the 16 groups of a half tile sit in 16 consecutive lanes
    m = fmaxf(m, __shfl_xor(m, 1));
    m = fmaxf(m, __shfl_xor(m, 2));
    m = fmaxf(m, __shfl_xor(m, 4));
    m = fmaxf(m, __shfl_xor(m, 8));
    const long long ht = g / GROUPS_PER_HALF;
    if ((threadIdx.x & 15) == 0 && ht < nh) tmax[(long long)qi * ldt + ht] = m;
}

// grid (q, ceil(n / 1024)), 256 threads: four consecutive rows per thread.
__global__ __launch_bounds__(256) void splitk_dense_kernel(const float* __restrict__ P, long long ldp, int q, int nsplit,
                                                           float* __restrict__ S, long long lds, long long n) {
    const int qi = blockIdx.x;
    const long long r0 = ((long long)blockIdx.y * 256 + threadIdx.x) * 4;
    if (r0 >= n) return;
    f32x4_t a = {0.f, 0.f, 0.f, 0.f};
    const float* src = P + (long long)qi * ldp + r0;
    for (int c = 0; c < nsplit; ++c) {
        a += *(const f32x4_t*)src;
        src += (long long)q * ldp;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (r0 + r < n) S[(long long)qi * lds + r0 + r] = a[r];
}

// Split-K plan.  A workgroup walks its K tiles one after the other (~1.6 us each) and a launch with
// fewer workgroups than CUs leaves the rest idle, so few tiles x long rows are cut along K; the
// price is the partial score tiles written and read back (nsplit * q * rows * 4 bytes each way)
// and a second launch.  The plan minimises a small cost model of the two (calibrated with
// scripts/exp_split.py); nsplit = 1 means one pass.
struct SplitPlan {
    int nsplit, kchunk;
};
SplitPlan split_plan(int64_t q, int64_t n, int64_t d) {
    const int64_t ntiles = dlc::cdiv(n, BM);
    const int64_t base = dlc::cdiv(ntiles, (int64_t)8) * 8 * dlc::cdiv(q, BNQ);   // workgroups of one pass
    const int64_t nk = d / BK;
    SplitPlan best{1, (int)nk};
#ifdef DLC_EXPERIMENT_NO_SPLIT   // perf experiment build only (scripts/)
    return best;
#endif
    if (base >= 256 || nk < 8) return best;
    const double t_k = 1.6, t_fix = 8.0, t_launch = 5.0, bytes_per_us = 3.0e6;
    const double part_bytes = (double)q * (double)ntiles * BM * 4.0;             // one chunk's partial scores
    // rounds of the chip are counted on the workgroups that DO work: the grid pads the tile count to a multiple of 8
    // (XCD mapping), the padding exits at once.  (Counting the padded grid made 1063 x 1063 x 75 008 pick 21 chunks =
    // 525 working workgroups = two rounds and 13 stragglers; 10 chunks = 250 is one round.)
    const int64_t work = ntiles * dlc::cdiv(q, BNQ);
    double best_t = (double)dlc::cdiv(work, (int64_t)256) * ((double)nk * t_k + t_fix);
    for (int64_t ns = 2; ns <= nk / 4 && ns <= 256; ++ns) {
        const int64_t kc = dlc::cdiv(nk, ns), ns_eff = dlc::cdiv(nk, kc);
        if (ns_eff != ns) continue;                                               // the same chunking as a smaller ns
        if ((double)ns_eff * part_bytes > (double)(1ll << 30)) break;
        const double t = (double)dlc::cdiv(work * ns_eff, (int64_t)256) * ((double)kc * t_k + t_fix) +
                         2.0 * (double)ns_eff * part_bytes / bytes_per_us + t_launch;
        if (t < best_t * 0.9) {                                                   // split only for a clear gain
            best_t = t;
            best = SplitPlan{(int)ns_eff, (int)kc};
        }
    }
    return best;
}

struct WsLayout {
    size_t gmax, tmax, part, rs_ids, rs_max, rs_scores, rs_idx, total;
    long long ldg, ldt, ldp;
    int kg;
    SplitPlan sp;
    int rparts;     // > 1: dlc_cosine_topk re-scores with this many workgroups per query (few queries, long rows)
    bool dense;     // small database: the score matrix itself is kept (chunk 0 of `part`) and the top-k is read off it
};

// Few queries take the bandwidth kernel (stored queries in LDS: q <= 4 and q * d * 2 bytes <= 64 KiB).
inline bool gemv_shape(int64_t q, int64_t nk) {
    const int qb = q <= 1 ? 1 : (q <= 2 ? 2 : 4);
    return q <= GEMV_MAX_Q && (long long)qb * nk * 128 <= GEMV_MAX_LDS;
}

// Small-database plan.  The standard plan never writes the score matrix: it keeps group maxima and
// re-scores the kg * 8 rows of the selected groups exactly -- a gather of kg * 8 * d * 2 bytes per
// query whatever the database size.  Against a small database (the reference's own scale: 1063
// key-frames x 75 000-d, where that gather is 192 of the 1063 rows, 29 MB per query, 30 GB per
// call) the whole score matrix is cheaper than the gather: q * n * 4 bytes.  The score pass then
// writes its tile(s) to the workspace, the reducing pass keeps the sum, and the selection reads the
// scores of its groups from it.  Scores are the MFMA-order fp32 sums (chunk-ordered when split).
constexpr int64_t DENSE_MAX_ROWS = 16384;
inline bool dense_plan(int64_t q, int64_t n, int64_t d) {
#ifdef DLC_EXPERIMENT_NO_DENSE   // perf experiment build only (scripts/)
    return false;
#endif
    return n <= DENSE_MAX_ROWS && !gemv_shape(q, d / BK);
}

// One workgroup per query gathers kg * 8 rows: with a handful of queries and long rows (1 query x
// 75 000-d: 29 MB through one CU, 300 us) the re-score is spread over one workgroup per selected group.
int rescore_parts(int64_t q, int64_t d, int kg, int k) {
    if (q > 32 || (int64_t)kg * GROUP * d * 2 < (2 << 20)) return 1;
    return std::min(kg, 3072 / k);            // the merge of the parts keeps parts * k 16-byte keys in 48 KiB of LDS
}

WsLayout ws_layout(int64_t q, int64_t n, int64_t d, int k) {
    WsLayout w;
    const int64_t ntiles = dlc::cdiv(n, BM);
    w.ldg = ntiles * (BM / GROUP);
    w.ldt = ntiles * 2;
    w.kg = k + SLACK;
    size_t o = 0;
    w.gmax = o; o += dlc::align_up((size_t)q * w.ldg * 4, 256);
    w.tmax = o; o += dlc::align_up((size_t)q * w.ldt * 4, 256);
    w.sp = split_plan(q, n, d);
    w.dense = dense_plan(q, n, d);
    w.ldp = ntiles * BM;
    w.part = o;
    if (w.sp.nsplit > 1 || w.dense) o += dlc::align_up((size_t)w.sp.nsplit * q * w.ldp * 4, 256);
    w.rparts = w.dense ? 1 : rescore_parts(q, d, w.kg, k);
    w.rs_ids = w.rs_max = w.rs_scores = w.rs_idx = o;
    if (w.rparts > 1) {
        w.rs_ids = o; o += dlc::align_up((size_t)q * w.kg * 4, 256);
        w.rs_max = o; o += dlc::align_up((size_t)q * w.kg * 4, 256);
        w.rs_scores = o; o += dlc::align_up((size_t)w.rparts * q * k * 4, 256);
        w.rs_idx = o; o += dlc::align_up((size_t)w.rparts * q * k * 8, 256);
    }
    w.total = o;
    return w;
}

int check_operands(dlc_ctx* ctx, int dtype, const void* Q, int64_t q, int64_t ldq, const void* DB, int64_t n,
                   int64_t lddb, int64_t d) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (dtype != DLC_BF16 && dtype != DLC_F16)
        return dlc::fail(ctx, DLC_ERR_UNSUPPORTED, "cosine match: dtype %d (need DLC_BF16 or DLC_F16)", dtype);
    if (!Q || !DB || q < 1 || n < 1 || d < 1) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "cosine match: null/empty operand");
    if (d % BK != 0) return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "cosine match: d=%lld must be a multiple of %d", (long long)d, BK);
    if (ldq < d || lddb < d || (ldq % 8) || (lddb % 8))
        return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "cosine match: row strides must be >= d and multiples of 8 elements");
    if (((uintptr_t)Q & 15) || ((uintptr_t)DB & 15))
        return dlc::fail(ctx, DLC_ERR_BAD_ARG, "cosine match: operands must be 16-byte aligned");
    // the score GEMM addresses a lane's 16 bytes as a 32-bit offset from its tile's first row:
    // 255 rows * stride + 128 bytes must stay below 2^32
    if (ldq * 2 * 255 + 128 > 0xffffffffll || lddb * 2 * 255 + 128 > 0xffffffffll)
        return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "cosine match: row stride %lld elements too large (255 rows must span < 4 GiB)",
                         (long long)(ldq > lddb ? ldq : lddb));
    if (q > 0x7fffff00ll) return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "cosine match: q too large");
    if (n > 0x7ffffff0ll) return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "cosine match: more than 2^31 rows in one shard");
    return DLC_OK;
}

template <typename Tag, int MODE, bool MASKQ>
int launch_gemm_masked(dlc_ctx* ctx, const GemmArgs& a, hipStream_t st);

template <typename Tag, int MODE>
int launch_gemm(dlc_ctx* ctx, const GemmArgs& a, hipStream_t st) {
    const int tail = a.q % BNQ;                            // queries in the last query block (0 = full)
    return (tail > 0 && tail <= 192) ? launch_gemm_masked<Tag, MODE, true>(ctx, a, st)
                                     : launch_gemm_masked<Tag, MODE, false>(ctx, a, st);
}

template <typename Tag> constexpr int tag_id() { return __is_same(Tag, dlc_bf16_tag) ? 0 : 1; }

// hipFuncSetAttribute acts on the CURRENT device (the context's, under its DeviceGuard), so the "already
// raised" flag lives in the context -- one context per device; a process-wide static would leave the
// second GPU of a process without the 128 KiB limit.
template <typename K>
int raise_lds_limit(dlc_ctx* ctx, K kern, int bit, int bytes) {
    const unsigned long long m = 1ull << bit;
    if (!(ctx->func_attr_set & m)) {
        DLC_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        ctx->func_attr_set |= m;
    }
    return DLC_OK;
}

template <typename Tag, int MODE, bool MASKQ>
int launch_gemm_masked(dlc_ctx* ctx, const GemmArgs& a, hipStream_t st) {
    auto kern = score_gemm_kernel<Tag, MODE, MASKQ>;
    int rc_attr = raise_lds_limit(ctx, kern, DLC_ATTR_GEMM_BASE + tag_id<Tag>() * 8 + MODE * 2 + (MASKQ ? 1 : 0), LDS_BYTES);
    if (rc_attr != DLC_OK) return rc_attr;
    GemmArgs b = a;
    b.ntiles = dlc::cdiv(a.n, BM);
    b.nqb = (int)dlc::cdiv(a.q, BNQ);
    if (MODE != GEMM_PARTIAL) { b.nsplit = 1; b.kchunk = a.nk; }
    const long long nwg = dlc::cdiv(b.ntiles, 8) * 8 * b.nqb * b.nsplit;
    if (nwg > 0x7fffffffll) return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "cosine match: %lld workgroups exceed the grid limit", nwg);
    dim3 grid((unsigned)nwg);
    // measured: 0-30 us of stagger pays from ~3 dispatch rounds on, 0-15 us below (scripts/exp_rows.py, exp_gemm.py)
    b.stagger_mult = (nwg >= 3 * 256) ? 4 : 2;
    b.stagger_phases = (nwg >= 3 * 256) ? DLC_STAGGER_PHASES : 16;
    hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), LDS_BYTES, st, b);
    DLC_LAUNCH_CHECK(ctx, "score_gemm_kernel");
    return DLC_OK;
}

inline bool use_gemv(const GemmArgs& a) { return gemv_shape(a.q, a.nk); }

template <typename Tag, int QB>
int launch_gemv_qb(dlc_ctx* ctx, const GemmArgs& a, hipStream_t st) {
    auto kern = score_gemv_kernel<Tag, QB>;
    const int lds = QB * a.nk * 128;
    int rc_attr = raise_lds_limit(ctx, kern, DLC_ATTR_GEMV_BASE + tag_id<Tag>() * 3 + (QB == 1 ? 0 : (QB == 2 ? 1 : 2)), GEMV_MAX_LDS);
    if (rc_attr != DLC_OK) return rc_attr;
    hipLaunchKernelGGL(kern, dim3((unsigned)a.nh), dim3(256), lds, st, a);
    DLC_LAUNCH_CHECK(ctx, "score_gemv_kernel");
    return DLC_OK;
}

template <typename Tag>
int launch_gemv(dlc_ctx* ctx, const GemmArgs& a, hipStream_t st) {
    if (a.q <= 1) return launch_gemv_qb<Tag, 1>(ctx, a, st);
    if (a.q <= 2) return launch_gemv_qb<Tag, 2>(ctx, a, st);
    return launch_gemv_qb<Tag, 4>(ctx, a, st);
}

// The score pass of a match: one kernel, or split-K partials + the reducing second pass.
// dense: write S instead of the group maxima.
// keep: the top-k call's small-database plan -- score tile(s) to the workspace even when K is not split,
// group maxima from the reducing pass, the summed scores kept in chunk 0.
template <typename Tag>
int launch_scores(dlc_ctx* ctx, const GemmArgs& a, bool dense, hipStream_t st, bool keep = false) {
    if (!dense && !keep && use_gemv(a)) return launch_gemv<Tag>(ctx, a, st);
    if (a.nsplit <= 1 && !keep)
        return dense ? launch_gemm<Tag, GEMM_DENSE>(ctx, a, st) : launch_gemm<Tag, GEMM_GROUPS>(ctx, a, st);
    int rc = launch_gemm<Tag, GEMM_PARTIAL>(ctx, a, st);
    if (rc != DLC_OK) return rc;
    if (dense) {
        dim3 grid((unsigned)a.q, (unsigned)dlc::cdiv(a.n, (int64_t)1024));
        hipLaunchKernelGGL(splitk_dense_kernel, grid, dim3(256), 0, st, a.P, a.ldp, a.q, a.nsplit, a.S, a.lds, a.n);
        DLC_LAUNCH_CHECK(ctx, "splitk_dense_kernel");
    } else {
        const long long groups = dlc::cdiv(a.n, BM) * (BM / GROUP);    // whole tiles: every lane of a half-tile reduction is live
        dim3 grid((unsigned)a.q, (unsigned)dlc::cdiv(groups, (long long)256));
        hipLaunchKernelGGL(splitk_groups_kernel, grid, dim3(256), 0, st, a.P, a.ldp, a.q, a.nsplit, a.gmax, a.ldg, a.ng,
                           a.tmax, a.ldt, a.nh, keep ? 1 : 0);
        DLC_LAUNCH_CHECK(ctx, "splitk_groups_kernel");
    }
    return DLC_OK;
}

}  // namespace

extern "C" size_t dlc_cosine_topk_workspace_bytes(int64_t q, int64_t n, int64_t d, int k) {
    if (q < 1 || n < 1 || d < BK || k < 1 || k > DLC_MAX_K) return 0;
    return ws_layout(q, n, d, k).total;
}

namespace {

struct MatchCall {
    GemmArgs a;
    WsLayout w;
};

int prepare_match(dlc_ctx* ctx, const char* what, int dtype, const void* Q, int64_t q, int64_t ldq, const void* DB,
                  int64_t n, int64_t lddb, int64_t d, int k, void* workspace, size_t workspace_bytes, MatchCall* mc) {
    int rc = check_operands(ctx, dtype, Q, q, ldq, DB, n, lddb, d);
    if (rc != DLC_OK) return rc;
    if (k < 1 || k > DLC_MAX_K) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "%s: k=%d outside 1..%d", what, k, DLC_MAX_K);
    mc->w = ws_layout(q, n, d, k);
    if (!workspace || workspace_bytes < mc->w.total)
        return dlc::fail(ctx, DLC_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", what, workspace_bytes, mc->w.total);
    if (((uintptr_t)workspace & 255)) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "%s: workspace must be 256-byte aligned", what);
    char* ws = (char*)workspace;
    GemmArgs& a = mc->a;
    a.Q = (const char*)Q; a.DB = (const char*)DB;
    a.ldq_b = ldq * 2; a.lddb_b = lddb * 2;
    a.q = (int)q; a.n = n; a.nk = (int)(d / BK);
    a.gmax = (float*)(ws + mc->w.gmax); a.ldg = mc->w.ldg;
    a.tmax = (float*)(ws + mc->w.tmax); a.ldt = mc->w.ldt;
    a.ng = dlc::cdiv(n, GROUP); a.nh = dlc::cdiv(n, HALF);
    a.S = nullptr; a.lds = 0;
    a.nsplit = mc->w.sp.nsplit; a.kchunk = mc->w.sp.kchunk;
    a.P = (float*)(ws + mc->w.part); a.ldp = mc->w.ldp;
    return DLC_OK;
}

int run_score(dlc_ctx* ctx, int dtype, MatchCall& mc, hipStream_t st) {
#ifdef DLC_EXPERIMENT_ALIAS_ROWS   // perf experiment build only (scripts/): every database row aliases row 0
    mc.a.lddb_b = 0;
#endif
    const int slot = (int)(ctx->prof_calls % DLC_PROFILE_RING);
    if (ctx->profiling) DLC_HIP_CHECK(ctx, hipEventRecord(ctx->ev_start[slot], st));
    int rc = (dtype == DLC_BF16) ? launch_scores<dlc_bf16_tag>(ctx, mc.a, false, st, mc.w.dense)
                                 : launch_scores<dlc_f16_tag>(ctx, mc.a, false, st, mc.w.dense);
    if (rc != DLC_OK) return rc;
    if (ctx->profiling) {
        DLC_HIP_CHECK(ctx, hipEventRecord(ctx->ev_stop[slot], st));
        ctx->prof_calls++;
    }
    return DLC_OK;
}

template <typename Tag, int THREADS, int RS_UNROLL, int MODE>
int launch_finish(dlc_ctx* ctx, const MatchCall& mc, int k, int64_t n, int64_t d, int64_t q, int64_t row_offset,
                  float* out_scores, int64_t* out_idx, bool small_lds, const FinishExtra& x, hipStream_t st) {
    const GemmArgs& a = mc.a;
    size_t dsm = fin_lds_fixed(mc.w.kg);
    const int tv_in_lds = MODE != FIN_RESCORE && !small_lds && (size_t)a.nh * 4 <= 96 * 1024;
    if (tv_in_lds) dsm += (size_t)a.nh * 4;
    dsm = dlc::align_up(dsm, 16);
    auto fk = finish_topk_kernel<Tag, THREADS, RS_UNROLL, MODE>;
    if (dsm > 48 * 1024)
        DLC_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)fk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dsm));
    hipLaunchKernelGGL(fk, dim3((unsigned)q, (unsigned)(MODE == FIN_RESCORE && x.gparts > 1 ? x.gparts : 1)), dim3(THREADS), dsm, st, a.tmax, a.ldt, (int)a.nh, tv_in_lds, a.gmax, a.ldg,
                       a.ng, mc.w.kg, a.Q, a.ldq_b, a.DB, a.lddb_b, (long long)n, (int)d, k, (long long)row_offset,
                       out_scores, (long long*)out_idx, x);
    DLC_LAUNCH_CHECK(ctx, "finish_topk_kernel");
    return DLC_OK;
}

// FinishExtra of a fused selection: under the small-database plan the scores are read from the workspace
inline FinishExtra fused_extra(const MatchCall& mc) {
    FinishExtra x{};
    if (mc.w.dense) { x.dense_S = mc.a.P; x.ld_s = mc.a.ldp; }
    return x;
}

template <int MODE>
int run_select(dlc_ctx* ctx, int dtype, const MatchCall& mc, int k, int64_t n, int64_t d, int64_t q, int64_t row_offset,
               float* out_scores, int64_t* out_idx, int flags, const FinishExtra& x, hipStream_t st) {
    const bool coop = (flags & DLC_SELECT_COOP) != 0;
    if (dtype == DLC_BF16)
        return coop ? launch_finish<dlc_bf16_tag, 256, 1, MODE>(ctx, mc, k, n, d, q, row_offset, out_scores, out_idx, true, x, st)
                    : launch_finish<dlc_bf16_tag, 512, 4, MODE>(ctx, mc, k, n, d, q, row_offset, out_scores, out_idx, false, x, st);
    return coop ? launch_finish<dlc_f16_tag, 256, 1, MODE>(ctx, mc, k, n, d, q, row_offset, out_scores, out_idx, true, x, st)
                : launch_finish<dlc_f16_tag, 512, 4, MODE>(ctx, mc, k, n, d, q, row_offset, out_scores, out_idx, false, x, st);
}

}  // namespace

extern "C" int dlc_cosine_topk(dlc_ctx* ctx, int dtype, const void* Q, int64_t q, int64_t ldq, const void* DB,
                               int64_t n, int64_t lddb, int64_t d, int k, int64_t row_offset, float* out_scores,
                               int64_t* out_idx, void* workspace, size_t workspace_bytes, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!out_scores || !out_idx) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "cosine_topk: null output");
    MatchCall mc;
    int rc = prepare_match(ctx, "cosine_topk", dtype, Q, q, ldq, DB, n, lddb, d, k, workspace, workspace_bytes, &mc);
    if (rc != DLC_OK) return rc;
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    rc = run_score(ctx, dtype, mc, (hipStream_t)stream);
    if (rc != DLC_OK) return rc;
    if (mc.w.rparts <= 1)
        return run_select<FIN_FUSED>(ctx, dtype, mc, k, n, d, q, row_offset, out_scores, out_idx, 0, fused_extra(mc), (hipStream_t)stream);
    // few queries, long rows: group selection, re-score with one workgroup per selected group, merge
    char* ws = (char*)workspace;
    FinishExtra x{};
    x.grp_ids = (int*)(ws + mc.w.rs_ids); x.grp_max = (float*)(ws + mc.w.rs_max); x.nq = q;
    rc = run_select<FIN_GROUPS>(ctx, dtype, mc, k, n, d, q, 0, nullptr, nullptr, 0, x, (hipStream_t)stream);
    if (rc != DLC_OK) return rc;
    x.gparts = mc.w.rparts;
    float* ps = (float*)(ws + mc.w.rs_scores);
    int64_t* pi = (int64_t*)(ws + mc.w.rs_idx);
    rc = run_select<FIN_RESCORE>(ctx, dtype, mc, k, n, d, q, row_offset, ps, pi, DLC_SELECT_COOP, x, (hipStream_t)stream);
    if (rc != DLC_OK) return rc;
    return dlc_topk_merge_strided(ctx, ps, q * k, pi, q * k, mc.w.rparts, q, k, out_scores, out_idx, stream);
}

extern "C" int dlc_cosine_score_groups(dlc_ctx* ctx, int dtype, const void* Q, int64_t q, int64_t ldq, const void* DB,
                                       int64_t n, int64_t lddb, int64_t d, int k, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    MatchCall mc;
    int rc = prepare_match(ctx, "cosine_score_groups", dtype, Q, q, ldq, DB, n, lddb, d, k, workspace, workspace_bytes, &mc);
    if (rc != DLC_OK) return rc;
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    return run_score(ctx, dtype, mc, (hipStream_t)stream);
}

extern "C" int dlc_cosine_select_topk(dlc_ctx* ctx, int dtype, const void* Q, int64_t q, int64_t ldq, const void* DB,
                                      int64_t n, int64_t lddb, int64_t d, int k, int64_t row_offset, float* out_scores,
                                      int64_t* out_idx, void* workspace, size_t workspace_bytes, int flags,
                                      void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!out_scores || !out_idx) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "cosine_select_topk: null output");
    MatchCall mc;
    int rc = prepare_match(ctx, "cosine_select_topk", dtype, Q, q, ldq, DB, n, lddb, d, k, workspace, workspace_bytes, &mc);
    if (rc != DLC_OK) return rc;
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    return run_select<FIN_FUSED>(ctx, dtype, mc, k, n, d, q, row_offset, out_scores, out_idx, flags, fused_extra(mc), (hipStream_t)stream);
}

extern "C" int dlc_cosine_groups_per_query(int k) { return (k < 1 || k > DLC_MAX_K) ? 0 : k + SLACK; }

extern "C" int dlc_cosine_select_groups(dlc_ctx* ctx, int dtype, const void* Q, int64_t q, int64_t ldq, const void* DB,
                                        int64_t n, int64_t lddb, int64_t d, int k, void* workspace,
                                        size_t workspace_bytes, int32_t* group_ids, float* group_max, int flags,
                                        void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!group_ids || !group_max) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "cosine_select_groups: null output");
    MatchCall mc;
    int rc = prepare_match(ctx, "cosine_select_groups", dtype, Q, q, ldq, DB, n, lddb, d, k, workspace, workspace_bytes, &mc);
    if (rc != DLC_OK) return rc;
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    FinishExtra x{};
    x.grp_ids = group_ids; x.grp_max = group_max; x.nq = q;
    return run_select<FIN_GROUPS>(ctx, dtype, mc, k, n, d, q, 0, nullptr, nullptr, flags, x, (hipStream_t)stream);
}

extern "C" int dlc_cosine_rescore_topk(dlc_ctx* ctx, int dtype, const void* Q, int64_t q, int64_t ldq, const void* DB,
                                       int64_t n, int64_t lddb, int64_t d, int k, int64_t row_offset,
                                       const int32_t* group_ids, const float* group_max, const float* all_group_max,
                                       int parts, float* out_scores, int64_t* out_idx, int flags, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!out_scores || !out_idx || !group_ids || !group_max || parts < 0 || (parts > 0 && !all_group_max))
        return dlc::fail(ctx, DLC_ERR_BAD_ARG, "cosine_rescore_topk: bad argument");
    int rc = check_operands(ctx, dtype, Q, q, ldq, DB, n, lddb, d);
    if (rc != DLC_OK) return rc;
    if (k < 1 || k > DLC_MAX_K) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "cosine_rescore_topk: k=%d outside 1..%d", k, DLC_MAX_K);
    MatchCall mc;                                   // no workspace needed: only the operands and kg
    mc.w = ws_layout(q, n, d, k);
    mc.a = GemmArgs{};
    mc.a.Q = (const char*)Q; mc.a.DB = (const char*)DB;
    mc.a.ldq_b = ldq * 2; mc.a.lddb_b = lddb * 2;
    mc.a.q = (int)q; mc.a.n = n; mc.a.nk = (int)(d / BK);
    mc.a.ng = dlc::cdiv(n, GROUP); mc.a.nh = dlc::cdiv(n, HALF);
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    FinishExtra x{};
    x.grp_ids = const_cast<int32_t*>(group_ids); x.grp_max = const_cast<float*>(group_max);
    x.all_max = all_group_max; x.parts = parts; x.nq = q;
    return run_select<FIN_RESCORE>(ctx, dtype, mc, k, n, d, q, row_offset, out_scores, out_idx, flags, x, (hipStream_t)stream);
}

namespace {
// Row b of a best-first candidate list [rows, kk] keeps its first k entries whose id lies in [0, limit0 + b); the rest
// of its k slots hold (-inf, -1).  One wave per row; kk <= DLC_MAX_K.
__global__ __launch_bounds__(256) void keep_older_kernel(const float* __restrict__ scores, const long long* __restrict__ idx,
                                                         long long rows, int kk, long long limit0, int k,
                                                         float* __restrict__ out_s, long long* __restrict__ out_i) {
    const int lane = threadIdx.x & 63;
    const long long b = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= rows) return;
    int kept = 0;
    for (int c0 = 0; c0 < kk && kept < k; c0 += 64) {
        const int c = c0 + lane;
        const long long id = c < kk ? idx[b * kk + c] : -1;
        const bool ok = id >= 0 && id < limit0 + b;
        const unsigned long long m = __ballot(ok);
        const int pos = kept + __popcll(m & ((1ull << lane) - 1ull));
        if (ok && pos < k) {
            out_s[b * k + pos] = scores[b * kk + c];
            out_i[b * k + pos] = id;
        }
        kept += __popcll(m);
    }
    for (int e = (kept < k ? kept : k) + lane; e < k; e += 64) {
        out_s[b * k + e] = -INFINITY;
        out_i[b * k + e] = -1;
    }
}
}  // namespace

extern "C" int dlc_topk_keep_older(dlc_ctx* ctx, const float* scores, const int64_t* idx, int64_t rows, int kk,
                                   int64_t limit0, int k, float* out_scores, int64_t* out_idx, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!scores || !idx || !out_scores || !out_idx || rows < 1 || kk < 1 || kk > DLC_MAX_K || k < 1 || k > DLC_MAX_K)
        return dlc::fail(ctx, DLC_ERR_BAD_ARG, "topk_keep_older: bad argument (1 <= kk, k <= %d)", DLC_MAX_K);
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    hipLaunchKernelGGL(keep_older_kernel, dim3((unsigned)dlc::cdiv(rows, (int64_t)4)), dim3(256), 0, (hipStream_t)stream, scores,
                       (const long long*)idx, (long long)rows, kk, (long long)limit0, k, out_scores, (long long*)out_idx);
    DLC_LAUNCH_CHECK(ctx, "keep_older_kernel");
    return DLC_OK;
}

extern "C" int dlc_topk_merge_strided(dlc_ctx* ctx, const float* scores, int64_t score_part_stride, const int64_t* idx,
                                      int64_t idx_part_stride, int parts, int64_t q, int k, float* out_scores,
                                      int64_t* out_idx, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!scores || !idx || !out_scores || !out_idx || parts < 1 || q < 1 || k < 1 || k > DLC_MAX_K ||
        score_part_stride < q * k || idx_part_stride < q * k)
        return dlc::fail(ctx, DLC_ERR_BAD_ARG, "topk_merge: bad argument");
    const size_t m = (size_t)parts * k;
    const size_t dsm = m * 16;
    if (dsm > 48 * 1024) return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "topk_merge: parts*k=%zu too large", m);
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    hipLaunchKernelGGL(merge_topk_kernel, dim3((unsigned)q), dim3(FIN_THREADS), dsm, (hipStream_t)stream, scores,
                       (long long)score_part_stride, (const long long*)idx, (long long)idx_part_stride, parts,
                       (long long)q, k, out_scores, (long long*)out_idx);
    DLC_LAUNCH_CHECK(ctx, "merge_topk_kernel");
    return DLC_OK;
}

extern "C" int dlc_topk_merge(dlc_ctx* ctx, const float* scores, const int64_t* idx, int parts, int64_t q, int k,
                              float* out_scores, int64_t* out_idx, void* stream) {
    return dlc_topk_merge_strided(ctx, scores, q * k, idx, q * k, parts, q, k, out_scores, out_idx, stream);
}

extern "C" size_t dlc_cosine_scores_workspace_bytes(int64_t q, int64_t n, int64_t d) {
    if (q < 1 || n < 1 || d < BK) return 0;
    const SplitPlan sp = split_plan(q, n, d);
    return sp.nsplit > 1 ? dlc::align_up((size_t)sp.nsplit * q * dlc::cdiv(n, BM) * BM * 4, 256) : 0;
}

extern "C" int dlc_cosine_scores(dlc_ctx* ctx, int dtype, const void* Q, int64_t q, int64_t ldq, const void* DB,
                                 int64_t n, int64_t lddb, int64_t d, float* S, int64_t lds, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    int rc = check_operands(ctx, dtype, Q, q, ldq, DB, n, lddb, d);
    if (rc != DLC_OK) return rc;
    if (!S || lds < n) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "cosine_scores: bad output");
    const SplitPlan sp = split_plan(q, n, d);
    const size_t need = dlc_cosine_scores_workspace_bytes(q, n, d);
    if (need && (!workspace || workspace_bytes < need))
        return dlc::fail(ctx, DLC_ERR_WORKSPACE, "cosine_scores: workspace %zu < %zu bytes", workspace_bytes, need);
    if (need && ((uintptr_t)workspace & 255)) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "cosine_scores: workspace must be 256-byte aligned");
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    GemmArgs a{};
    a.Q = (const char*)Q; a.DB = (const char*)DB;
    a.ldq_b = ldq * 2; a.lddb_b = lddb * 2;
    a.q = (int)q; a.n = n; a.nk = (int)(d / BK);
    a.S = S; a.lds = lds;
    a.nsplit = sp.nsplit; a.kchunk = sp.kchunk;
    a.P = (float*)workspace; a.ldp = dlc::cdiv(n, BM) * BM;
    return (dtype == DLC_BF16) ? launch_scores<dlc_bf16_tag>(ctx, a, true, (hipStream_t)stream)
                               : launch_scores<dlc_f16_tag>(ctx, a, true, (hipStream_t)stream);
}

extern "C" int dlc_l2_normalize_rows(dlc_ctx* ctx, int src_dtype, const void* src, int64_t n, int64_t d, int64_t lds,
                                     int center, int dst_dtype, void* dst, int64_t ldd, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!src || !dst || n < 1 || d < 1 || lds < d || ldd < d)
        return dlc::fail(ctx, DLC_ERR_BAD_ARG, "l2_normalize_rows: bad argument");
    if (ldd % BK) return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "l2_normalize_rows: ldd=%lld must be a multiple of %d", (long long)ldd, BK);
    if (ldd > 0x7fffffff) return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "l2_normalize_rows: ldd too large");
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    hipStream_t st = (hipStream_t)stream;
    if (src_dtype != DLC_F32 && src_dtype != DLC_F64)
        return dlc::fail(ctx, DLC_ERR_UNSUPPORTED, "l2_normalize_rows: dtype pair %d -> %d", src_dtype, dst_dtype);
    if (dst_dtype != DLC_BF16 && dst_dtype != DLC_F16)
        return dlc::fail(ctx, DLC_ERR_UNSUPPORTED, "l2_normalize_rows: dtype pair %d -> %d", src_dtype, dst_dtype);
    unsigned short* o = (unsigned short*)dst;
    // one-pass register form: rows of <= 64 (a wave per row) or <= 256 threads x 16 vectors of 16 bytes
    // whose vectors can be loaded / stored aligned; anything else takes the multi-pass kernel
    const int vw = src_dtype == DLC_F32 ? 4 : 2;
    const bool aligned = ((uintptr_t)src & 15) == 0 && (lds % vw) == 0 && ((uintptr_t)dst & 7) == 0;
    const int64_t cap_wave = 64 * 16 * vw, cap_block = 256 * 16 * vw;
#define DLC_NORM_REGS(SRC, TAG, TPR)                                                                              \
    hipLaunchKernelGGL((l2_normalize_regs_kernel<SRC, TAG, TPR, 16>), dim3((unsigned)(TPR == 64 ? dlc::cdiv(n, 4) : n)), \
                       dim3(256), 0, st, (const SRC*)src, (long long)lds, (long long)n, (int)d, center, o, (long long)ldd)
#define DLC_NORM(SRC, TAG) \
    hipLaunchKernelGGL((l2_normalize_kernel<SRC, TAG>), dim3((unsigned)n), dim3(256), 0, st, (const SRC*)src, (long long)lds, (int)d, center, o, (long long)ldd)
#define DLC_NORM_PICK(SRC, TAG)                                                   \
    do {                                                                          \
        if (aligned && ldd <= cap_wave) DLC_NORM_REGS(SRC, TAG, 64);              \
        else if (aligned && ldd <= cap_block) DLC_NORM_REGS(SRC, TAG, 256);       \
        else DLC_NORM(SRC, TAG);                                                  \
    } while (0)
    if (src_dtype == DLC_F32 && dst_dtype == DLC_BF16) DLC_NORM_PICK(float, dlc_bf16_tag);
    else if (src_dtype == DLC_F32 && dst_dtype == DLC_F16) DLC_NORM_PICK(float, dlc_f16_tag);
    else if (src_dtype == DLC_F64 && dst_dtype == DLC_BF16) DLC_NORM_PICK(double, dlc_bf16_tag);
    else DLC_NORM_PICK(double, dlc_f16_tag);
#undef DLC_NORM_PICK
#undef DLC_NORM_REGS
#undef DLC_NORM
    DLC_LAUNCH_CHECK(ctx, "l2_normalize_kernel");
    return DLC_OK;
}
