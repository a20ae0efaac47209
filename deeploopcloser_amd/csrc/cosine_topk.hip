// Cosine-similarity + top-k match for gfx950 (MI355X).
//
// Pipeline of one dlc_cosine_topk() call (all on the caller's stream):
//   1. score_gemm_kernel   S~ = DB . Q^T on the bf16/f16 MFMA (fp32 accumulate).
//      The score tile never leaves the accumulators: the epilogue keeps, per
//      query, the maximum of every aligned block of 8 database rows
//      ("group", gmax) and of every 128 rows ("half tile", tmax).  The
//      database is streamed from HBM exactly once.  These fp32 scores only CHOOSE CANDIDATES.
//   2. finish_topk_kernel  one workgroup per query:
//      a. the kg = k + SLACK half tiles with the largest tmax, then the kg groups
//         with the largest gmax inside them.  Every member of the exact top-k
//         lies in one of those groups (the k-th largest group maximum is a
//         lower bound of the k-th largest score);
//      b. fp64 re-score of the 8 rows of each selected group (rescore8_f64: the fp64 sum of the exact products of the
//         stored elements in one fixed order -- THE score of a (query, row) pair in every plan, shard and batch; a
//         gather of kg*8 rows per query);
//      c. top-k of the kg*8 candidates on the fp64 key round(s * 2^40), descending, ties toward the lower database
//         index; and the certificate: the result is exact when the k-th fp64 score clears the best fp32 score left
//         behind by more than the score pass's error bound tau;
//   3. exhaustive_topk_kernel  the queries that failed the certificate (every other workgroup leaves at once).
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
constexpr int SLACK = 4;         // extra groups kept beyond k: how crowded the k-th score may be before the exhaustive pass has to run

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
// Selection.  Two kinds of keys:
//   * the fp32 (MFMA-order) scores of the score pass only CHOOSE CANDIDATES: packed 64-bit keys
//     (monotone score key << 32) | ~id32, so "larger key" == "higher score, then lower id"; key 0 = empty;
//   * the ORDER of every result is decided on fp64 re-scores of the candidates (rescore8_f64 below):
//     key = round(S64 * 2^40), larger key first, ties -> lower database index.  The same two rules hold in
//     every plan, in the exhaustive pass and in the merge of per-shard results, so the indices a call returns do
//     not depend on the plan, the shard count or the batch a query is part of.
// A selection is CERTIFIED when the k-th fp64 score exceeds, by more than the score pass's error bound tau,
// the largest fp32 score any row outside the candidate set can have; otherwise the exhaustive pass re-scores
// every group that could still hold a top-k row.
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned f32_key(float x) {   // monotone: larger float -> larger key
    unsigned u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_f32(unsigned k) {   // inverse of f32_key
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}
__device__ __forceinline__ unsigned long long pack_key(float score, unsigned id) {
    return ((unsigned long long)f32_key(score) << 32) | (unsigned long long)(~id);
}
__device__ __forceinline__ unsigned key_id(unsigned long long key) { return ~(unsigned)key; }

// Ordering key of an fp64 score: quantised to 2^-40 (oracle/cosine.py: order_key), so that two rows whose exact
// scores are equal but whose fp64 sums differ in the last bits (the same products in another order) still tie.
constexpr long long KEY64_EMPTY = (long long)0x8000000000000000ull;
__device__ __forceinline__ long long f64_key(double s) {
    const double x = s * 1099511627776.0;                 // 2^40
    if (!(x > -4.0e18)) return KEY64_EMPTY + 1;           // -inf, NaN, absurdly negative: last
    if (x > 4.0e18) return 0x7fffffffffffffffll;
    return __double2ll_rn(x);                             // round half to even, as np.round
}

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

constexpr int FIN_THREADS = 512;            // merge / exhaustive workgroup; finish workgroup of the stand-alone variant
// finish kernel LDS carve (bytes) for kg selected groups:
//   ckey u64 [kg*16] | ck64 i64 [kg*8] | cs64 f64 [kg*8] | crow i32 [kg*8] | sel i32 [kg+1] | sel2 i32 [kg+1] | misc 64 B
__host__ __device__ inline size_t fin_lds_fixed(int kg) {
    return (size_t)kg * (GROUPS_PER_HALF * 8 + GROUP * (8 + 8 + 4)) + (size_t)(kg + 1) * 8 + 64;
}

// Rank-by-counting on unique packed keys in LDS: out[rank] = element index, for rank < k -- and for rank == k when
// `plus_one` (the best element that was NOT selected: its score bounds everything left behind).
// Every thread walks the keys in the same order (LDS broadcast reads).
__device__ __forceinline__ void rank_select(const unsigned long long* keys, int m, int k, int* out, bool plus_one = false) {
    const int lim = plus_one ? k + 1 : k;
    for (int e = threadIdx.x; e < m; e += blockDim.x) {
        const unsigned long long ke = keys[e];
        if (ke == 0ull) continue;
        int rank = 0;
#pragma unroll 8
        for (int j = 0; j < m; ++j) rank += keys[j] > ke ? 1 : 0;
        if (rank < lim) out[rank] = e;
    }
}
// The same on (fp64 ordering key, id) pairs: larger key first, then the lower id; KEY64_EMPTY = no element.
template <typename Id>
__device__ __forceinline__ void rank_select64(const long long* keys, const Id* ids, int m, int k, int* out) {
    for (int e = threadIdx.x; e < m; e += blockDim.x) {
        const long long ke = keys[e];
        if (ke == KEY64_EMPTY) continue;
        const Id ie = ids[e];
        int rank = 0;
#pragma unroll 4
        for (int j = 0; j < m; ++j) {
            const long long kj = keys[j];
            rank += (kj > ke || (kj == ke && kj != KEY64_EMPTY && ids[j] < ie)) ? 1 : 0;
        }
        if (rank < k) out[rank] = e;
    }
}

// fp64 dot products of one stored query row with GROUP = 8 stored database rows: the ONE definition of a score's
// value in this library (include/dlc.h).  bf16 / fp16 -> fp64 conversions are exact, so every product is exact and
// only the additions round.  Lane l takes the 16-byte pieces l, l + 64, ... of the rows in ascending order, one fp64
// fma chain per row; the 64 chains are then combined by a butterfly (xor 32, 16, ..., 1; every lane ends with the
// sum).  The value depends on (query row, database row, d) only -- not on the plan, the shard, the kernel's thread
// count or RS_UNROLL (which only batches the loads: RS_UNROLL * (GROUP + 1) 16-byte loads in flight per lane).
template <typename Tag, int RS_UNROLL>
__device__ __forceinline__ void rescore8_f64(const char* qrow, const char* const (&rows)[GROUP], int d, int lane,
                                             double (&acc)[GROUP]) {
#pragma unroll
    for (int r = 0; r < GROUP; ++r) acc[r] = 0.0;
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
            double qd[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                qd[2 * e] = (double)Mfma16<Tag>::to_f32((unsigned short)(qw[e] & 0xffffu));
                qd[2 * e + 1] = (double)Mfma16<Tag>::to_f32((unsigned short)(qw[e] >> 16));
            }
#pragma unroll
            for (int r = 0; r < GROUP; ++r) {
                const unsigned w4[4] = {rv[u][r].x, rv[u][r].y, rv[u][r].z, rv[u][r].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[r] = fma(qd[2 * e], (double)Mfma16<Tag>::to_f32((unsigned short)(w4[e] & 0xffffu)), acc[r]);
                    acc[r] = fma(qd[2 * e + 1], (double)Mfma16<Tag>::to_f32((unsigned short)(w4[e] >> 16)), acc[r]);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < GROUP; ++r) {
        double v = acc[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        acc[r] = v;
    }
}

// R rows of rescore8_f64 (R = 1, 2, 3), U pieces of every row and of the query in flight per lane: the same pieces per lane
// in the same order, the same butterfly -- the same values.
template <typename Tag, int R, int U>
__device__ __forceinline__ void rescore_rows_f64(const char* qrow, const char* const (&rows)[R], int d, int lane,
                                                 double (&acc)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0;
    for (int d0 = lane * 8; d0 < d; d0 += 512 * U) {
        uint4 qv[U], rv[U][R];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int dd = d0 + u * 512;
            const bool ok = dd < d;
            qv[u] = ok ? *(const uint4*)(qrow + (long long)dd * 2) : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < R; ++r) rv[u][r] = ok ? *(const uint4*)(rows[r] + (long long)dd * 2) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned qw[4] = {qv[u].x, qv[u].y, qv[u].z, qv[u].w};
            double qd[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                qd[2 * e] = (double)Mfma16<Tag>::to_f32((unsigned short)(qw[e] & 0xffffu));
                qd[2 * e + 1] = (double)Mfma16<Tag>::to_f32((unsigned short)(qw[e] >> 16));
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const unsigned w4[4] = {rv[u][r].x, rv[u][r].y, rv[u][r].z, rv[u][r].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[r] = fma(qd[2 * e], (double)Mfma16<Tag>::to_f32((unsigned short)(w4[e] & 0xffffu)), acc[r]);
                    acc[r] = fma(qd[2 * e + 1], (double)Mfma16<Tag>::to_f32((unsigned short)(w4[e] >> 16)), acc[r]);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        double v = acc[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        acc[r] = v;
    }
}
template <typename Tag>
__device__ __forceinline__ double rescore1_f64(const char* qrow, const char* row, int d, int lane) {
    const char* const rows[1] = {row};
    double acc[1];
    rescore_rows_f64<Tag, 1, 4>(qrow, rows, d, lane, acc);
    return acc[0];
}

// One workgroup per query: half-tile selection -> group selection -> fp64 re-score of the candidates' rows ->
// final top-k -> certification.
// THREADS / RS_UNROLL pick the footprint: (512, 4) is the fastest stand-alone form; (256, 1) stays under 96 VGPRs
// and a few KiB of LDS so that its workgroups can share a CU with a resident score-GEMM workgroup (128 KiB LDS,
// 2 x 198 VGPRs per SIMD) when the two run on different streams.
// MODE: FIN_FUSED  all of it;
//       FIN_GROUPS stops after the group selection and writes the kg selected groups of every query (grp_ids, -1 =
//                  none; grp_max [q, kg + 1]: their maxima in rank order, and in column kg the largest maximum among
//                  the shard's groups that are NOT listed, -inf if there is none);
//       FIN_RESCORE starts from such a list.  With parts > 0 it first drops every own group that cannot be among
//                  the kg best groups of the WHOLE database: all_max holds the lists of all `parts` shards
//                  ([parts, q, kg + 1], an all-gather of grp_max), and a group with kg or more strictly larger
//                  maxima anywhere is out.  Each shard then re-scores ~kg/parts groups per query instead of kg, and
//                  writes to bound_out[q] the largest fp32 score a row outside the surviving groups of ALL shards
//                  can have (the same value on every shard): the merge certifies against it.
// tau of query qi: the plan's bound for operands of norm <= 1.005, times the query's scale where the caller supplied one
// (dlc_cosine_tau_scale).  tau = +inf is legal everywhere below: nothing is pruned, nothing certifies, the exhaustive
// pass takes every group.
__device__ __forceinline__ double scaled_tau(double tau, const float* __restrict__ tau_scale, int qi) {
    return tau_scale ? tau * (double)tau_scale[qi] : tau;
}

enum { FIN_FUSED = 0, FIN_GROUPS = 1, FIN_RESCORE = 2 };
struct FinishArgs {
    float* tmax; long long ldt; int nh; int tv_in_lds;
    const float* gmax; long long ldg; long long ng;
    int kg;                       // groups kept per query (k + SLACK)
    const char* Q; long long ldq_b; const char* DB; long long lddb_b;
    long long n; int d; int k; long long row_offset;
    float* out_s;                 // [q, k] fp32 (the fp64 score rounded once); may be null in FIN_RESCORE
    double* out_s64;              // [q, k] fp64, never null inside the kernels (the workspace lends one)
    long long* out_i;             // [q, k]
    int* status;                  // FIN_FUSED: [q] 0 = certified, 1 = the exhaustive pass has to run
    double tau;                   // error bound of the score pass's fp32 scores against the fp64 re-score (rows of norm <= 1.005)
    const float* tau_scale;       // [q] or null: query qi certifies with tau * tau_scale[qi] (dlc_cosine_tau_scale: operands
                                  // of any norm; inf = certify nothing, the exhaustive pass decides)
    int* grp_ids;                 // [q, kg]
    float* grp_max;               // [q, kg + 1]
    const float* all_max;         // [parts, q, kg + 1] or null
    int parts;
    long long nq;
    int gparts;                   // FIN_RESCORE: > 1 = grid.y workgroups per query, workgroup y re-scores the listed
                                  // groups e with e % gparts == y and writes its top-k at [y][q][k]
    float* bound_out;             // FIN_RESCORE: [q] or null
    const float* dense_S;         // small-database plan: the fp32 score matrix [q, ld_s] is in the workspace, so the
    long long ld_s;               // candidates are ROWS: the k + rslack best fp32 scores of the selected groups
    int rslack;
    int limited;                  // 1: query qi may only see rows < limit0 + qi (the streaming detector's batches); the group and
    long long limit0;             // half-tile maxima of the score pass stay upper bounds of what it may see
};
// rows / groups / half tiles query qi may see
__device__ __forceinline__ long long visible_rows(long long n, int limited, long long limit0, int qi) {
    if (!limited) return n;
    const long long lim = limit0 + qi;
    return lim < 0 ? 0 : (lim < n ? lim : n);
}

// (the body of the two finish_topk_kernel forms below)
template <typename Tag, int THREADS, int RS_UNROLL, int MODE>
__device__ __forceinline__ void finish_topk_body(const FinishArgs& a) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    constexpr int FIN_WAVES = THREADS / 64;
    constexpr int FIN_THREADS = THREADS;      // shadows the namespace constant inside this kernel
    constexpr int GPH = GROUPS_PER_HALF;
    const int kg = a.kg, k = a.k;
    const double tau = scaled_tau(a.tau, a.tau_scale, (int)blockIdx.x);
    const long long n = visible_rows(a.n, a.limited, a.limit0, (int)blockIdx.x);
    const long long ng = a.limited ? (n + GROUP - 1) / GROUP : a.ng;
    if (n <= 0) {                                               // (only with a limit) nothing this query may see: an empty list
        if constexpr (MODE == FIN_GROUPS) {
            for (int e = threadIdx.x; e <= kg; e += THREADS) {
                if (e < kg) a.grp_ids[(long long)blockIdx.x * kg + e] = -1;
                a.grp_max[(long long)blockIdx.x * (kg + 1) + e] = -INFINITY;
            }
        } else {
            const long long oq = (long long)blockIdx.y * a.nq + blockIdx.x;
            for (int e = threadIdx.x; e < k; e += THREADS) {
                if (a.out_s) a.out_s[oq * k + e] = -INFINITY;
                a.out_s64[oq * k + e] = -INFINITY;
                a.out_i[oq * k + e] = -1;
            }
            if (MODE == FIN_FUSED && threadIdx.x == 0) a.status[blockIdx.x] = 0;
            if (MODE == FIN_RESCORE && a.bound_out && blockIdx.y == 0 && threadIdx.x == 0) a.bound_out[blockIdx.x] = -INFINITY;
        }
        return;
    }
    unsigned long long* ckey = (unsigned long long*)dsm;                       // [kg * GPH] packed fp32 keys
    long long* ck64 = (long long*)(dsm + (size_t)kg * GPH * 8);                // [kg * GROUP] fp64 ordering keys
    double* cs64 = (double*)(dsm + (size_t)kg * (GPH * 8 + GROUP * 8));        // [kg * GROUP] fp64 scores
    int* crow = (int*)(dsm + (size_t)kg * (GPH * 8 + GROUP * 16));             // [kg * GROUP] candidate rows (shard-local)
    int* sel = (int*)(dsm + (size_t)kg * (GPH * 8 + GROUP * 20));              // [kg + 1]
    int* sel2 = sel + kg + 1;                                                  // [kg + 1]
    unsigned* misc = (unsigned*)(sel2 + kg + 1);                               // [16]
    const int qi = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // [nh] half-tile maxima of this query: staged in LDS, or consumed in place in the workspace
    // row (which the next GEMM rewrites anyway) when LDS is to be kept small or the shard is huge
    float* tv = a.tv_in_lds ? (float*)(dsm + fin_lds_fixed(kg)) : a.tmax + (long long)qi * a.ldt;
    if (a.tv_in_lds)
        for (int e = tid; e < a.nh; e += FIN_THREADS) tv[e] = a.tmax[(long long)qi * a.ldt + e];
    for (int e = tid; e <= kg; e += FIN_THREADS) { sel[e] = -1; sel2[e] = -1; }
    if (tid < 16) misc[tid] = 0u;
    __syncthreads();
    int kg2;
    unsigned bkey = 0u;      // fp32 key of the largest group maximum left behind by the selection; 0 = nothing left behind
    if constexpr (MODE != FIN_RESCORE) {
    // ---- level 1: the kt half tiles with the largest maximum (ties -> lower tile).
    // Each wave extracts the kt best of its slice by repeated wave arg-max (DPP, no barrier);
    // the FIN_WAVES * kt survivors are ranked together.  What a wave has left after its kt extractions
    // (misc[w]) and the survivor of rank kt bound every half tile that is not selected.
    const int nh = a.limited ? (int)((n + HALF - 1) / HALF) : a.nh;
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
        const unsigned rest = wave_max_u32(bh);          // the best half tile this wave did not hand in
        if (lane == 0) misc[w] = rest;
    }
    __syncthreads();
    rank_select(ckey, FIN_WAVES * kt, kt, sel, true);    // sel[rank] = slot in ckey
    __syncthreads();
#pragma unroll
    for (int ww = 0; ww < FIN_WAVES; ++ww) bkey = max(bkey, misc[ww]);
    if (sel[kt] >= 0) bkey = max(bkey, (unsigned)(ckey[sel[kt]] >> 32));
    if (tid < kt) { const int c = sel[tid]; sel[tid] = c < 0 ? -1 : (int)key_id(ckey[c]); }   // -> half-tile index
    __syncthreads();

    // ---- level 2: among their groups, the kg2 groups with the largest maximum (+ the best one left behind)
    const int m2 = kt * GPH;
    for (int e = tid; e < m2; e += FIN_THREADS) {
        const int ht = sel[e / GPH];
        const long long g = (long long)ht * GPH + (e % GPH);
        float gm = (ht >= 0 && g < ng) ? a.gmax[(long long)qi * a.ldg + g] : 0.0f;
        if (a.limited && a.dense_S && ht >= 0 && g == ng - 1 && (n & (GROUP - 1))) {
            // the one group a limit cuts through: its maximum over the rows the query may see, from the score matrix of the
            // small-database plan (the score pass's maximum counts up to 7 rows it may not see; as the k-th largest group
            // maximum such a number prunes groups that hold top-k rows -- safe, the certificate then fails, but that was
            // an exhaustive pass for one or two queries of every batch of 32)
            gm = -INFINITY;
            for (long long row = g * GROUP; row < n; ++row) gm = fmaxf(gm, a.dense_S[(long long)qi * a.ld_s + row]);
        }
        ckey[e] = (ht >= 0 && g < ng) ? pack_key(gm, (unsigned)g) : 0ull;
    }
    __syncthreads();
    kg2 = min(kg, m2);
    rank_select(ckey, m2, kg2, sel2, true);
    __syncthreads();
    if (sel2[kg2] >= 0) bkey = max(bkey, (unsigned)(ckey[sel2[kg2]] >> 32));
    if constexpr (MODE == FIN_GROUPS) {
        const long long W = kg + 1;
        for (int e = tid; e < kg; e += FIN_THREADS) {
            const int c = e < kg2 ? sel2[e] : -1;
            a.grp_ids[(long long)qi * kg + e] = c < 0 ? -1 : (int)key_id(ckey[c]);
            a.grp_max[(long long)qi * W + e] = c < 0 ? -INFINITY : a.gmax[(long long)qi * a.ldg + key_id(ckey[c])];
        }
        if (tid == 0) a.grp_max[(long long)qi * W + kg] = bkey ? key_f32(bkey) : -INFINITY;
        return;
    }
    // Slack groups that cannot hold a top-k row are not gathered: with u' = the k-th largest group maximum (k groups hold a
    // row with an fp32 score >= u', so the k-th best fp64 score is >= u' - tau), a group whose maximum is below u' - 2 tau
    // has no row with an fp64 score above u' - tau.  Ranks k .. kg-1 are usually such groups (they are there for the crowded
    // case): dropping them saves their 8 rows' gather each.  They join what is left behind -- and cannot fail the
    // certificate (s_k >= u' - tau > their maximum + tau).
    if (tid < kg2) {
        const int c = sel2[tid];
        int g = c < 0 ? -1 : (int)key_id(ckey[c]);
        // (with a limit and no score matrix the group the limit cuts through has an inflated maximum, u' may be that
        // number, and "k groups hold a row the query may see with a score >= u'" no longer holds: nothing is pruned then)
        if (tid >= k && g >= 0 && sel2[k - 1] >= 0 && !(a.limited && !a.dense_S)) {
            const double uk = (double)key_f32((unsigned)(ckey[sel2[k - 1]] >> 32));
            if ((double)key_f32((unsigned)(ckey[c] >> 32)) < uk - 2.0 * tau) g = -2;          // pruned (after the reads below)
        }
        misc[9] = 0u;                                       // (benign: every writer stores the same value)
        crow[tid] = g;                                      // staged: sel2 / ckey are still being read by the other threads
    }
    __syncthreads();
    if (tid < kg2) {
        const int g = crow[tid];
        if (g == -2) atomicMax(&misc[9], (unsigned)(ckey[sel2[tid]] >> 32));
    }
    __syncthreads();
    if (tid < kg2) sel2[tid] = crow[tid] < 0 ? -1 : crow[tid];                                 // -> group index
    bkey = max(bkey, misc[9]);
    __syncthreads();
    } else {
        // ---- start from a group list; optionally filter it against the other shards' maxima
        kg2 = kg;
        const long long W = kg + 1;
        if (a.parts > 0) {
            // what ALL shards leave behind: every shard's own rest bound (column kg) and every listed group that the
            // filter drops, here or elsewhere -- the same number on every shard
            unsigned bk = 0u;
            for (int e = tid; e < a.parts * (int)W; e += FIN_THREADS) {
                const int pp = e / (int)W, j = e % (int)W;
                const float v = a.all_max[((long long)pp * a.nq + qi) * W + j];
                if (v == -INFINITY) continue;
                if (j == kg) { bk = max(bk, f32_key(v)); continue; }
                int greater = 0;
                for (int p2 = 0; p2 < a.parts; ++p2) {
                    const float* av = a.all_max + ((long long)p2 * a.nq + qi) * W;
                    for (int j2 = 0; j2 < kg; ++j2) greater += av[j2] > v ? 1 : 0;
                }
                if (greater >= kg) bk = max(bk, f32_key(v));
            }
            if (bk) atomicMax(&misc[8], bk);
        }
        for (int e = tid; e < kg; e += FIN_THREADS) {
            int g = a.grp_ids[(long long)qi * kg + e];
            if (g >= 0 && a.parts > 0) {
                const float v = a.grp_max[(long long)qi * W + e];
                int greater = 0;
                for (int pp = 0; pp < a.parts; ++pp) {
                    const float* av = a.all_max + ((long long)pp * a.nq + qi) * W;
                    for (int j = 0; j < kg; ++j) greater += av[j] > v ? 1 : 0;
                }
                if (greater >= kg) g = -1;
            }
            if (a.gparts > 1 && e % a.gparts != (int)blockIdx.y) g = -1;
            sel2[e] = g;
        }
        __syncthreads();
        if (a.parts > 0) bkey = misc[8];
        else {
            const float v = a.grp_max[(long long)qi * W + kg];
            bkey = v == -INFINITY ? 0u : f32_key(v);
        }
        if (a.bound_out && blockIdx.y == 0 && tid == 0) a.bound_out[qi] = bkey ? key_f32(bkey) : -INFINITY;
    }

    // ---- candidate rows
    int m3;
    if (a.dense_S) {
        // small-database plan: the selected groups' fp32 scores are in the workspace -- keep their k + rslack best
        // ROWS (and remember the best row left behind)
        const int m = kg2 * GROUP;
        for (int e = tid; e < m; e += FIN_THREADS) {
            const int g = sel2[e / GROUP];
            const long long row = (long long)g * GROUP + (e % GROUP);
            ckey[e] = (g >= 0 && row < n) ? pack_key(a.dense_S[(long long)qi * a.ld_s + row], (unsigned)row) : 0ull;
        }
        m3 = min(k + a.rslack, m);
        for (int e = tid; e <= m3 && e < kg * GROUP; e += FIN_THREADS) crow[e] = -1;
        __syncthreads();
        if (m3 < m) rank_select(ckey, m, m3, crow, true);
        else rank_select(ckey, m, m3, crow, false);
        __syncthreads();
        if (m3 < m && crow[m3] >= 0) bkey = max(bkey, (unsigned)(ckey[crow[m3]] >> 32));
        __syncthreads();
        for (int e = tid; e < m3; e += FIN_THREADS) { const int c = crow[e]; crow[e] = c < 0 ? -1 : (int)key_id(ckey[c]); }
    } else {
        m3 = kg2 * GROUP;
        for (int e = tid; e < m3; e += FIN_THREADS) {
            const int g = sel2[e / GROUP];
            const long long row = (long long)g * GROUP + (e % GROUP);
            crow[e] = (g >= 0 && row < n) ? (int)row : -1;
        }
    }
    __syncthreads();

    // ---- fp64 re-score: wave w takes candidates 8w .. 8w + 7, then 8 (w + FIN_WAVES) ... (a group's 8 rows are
    // neighbours in memory).  The small-database plan's candidates are k + rslack single ROWS: they are dealt round the
    // waves instead, one row at a time (wave w: candidates w, w + FIN_WAVES, ...) -- nine rows of 8 KB were one wave's
    // 8 rows and a second wave's one.  Same value either way (rescore1_f64).
    const char* qrow = a.Q + (long long)qi * a.ldq_b;
    if (MODE == FIN_FUSED && a.dense_S != nullptr) {                   // (only the fused form runs the small-database plan)
        for (int ci = w; ci < m3; ci += FIN_WAVES) {
            const int id = __builtin_amdgcn_readfirstlane(crow[ci]);
            const double sc = id < 0 ? 0.0 : rescore1_f64<Tag>(qrow, a.DB + (long long)id * a.lddb_b, a.d, lane);
            if (lane == 0) {
                cs64[ci] = id < 0 ? -INFINITY : sc;
                ck64[ci] = id < 0 ? KEY64_EMPTY : f64_key(sc);
            }
        }
    } else
    for (int s = w; s * GROUP < m3; s += FIN_WAVES) {
        int ids[GROUP];
        bool any = false;
        const char* rows[GROUP];
#pragma unroll
        for (int r = 0; r < GROUP; ++r) {
            // (every lane reads the same LDS word: through readfirstlane the ids, and the row pointers behind them, live in
            // scalar registers -- sixteen vector registers fewer, which the 96-register cooperative forms did not have: the
            // re-score form of the sharded protocol spilled two)
            ids[r] = __builtin_amdgcn_readfirstlane(s * GROUP + r < m3 ? crow[s * GROUP + r] : -1);
            any |= ids[r] >= 0;
            rows[r] = a.DB + (long long)(ids[r] < 0 ? 0 : ids[r]) * a.lddb_b;
        }
        double acc[GROUP] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (any) rescore8_f64<Tag, RS_UNROLL>(qrow, rows, a.d, lane, acc);
#pragma unroll
        for (int r = 0; r < GROUP; ++r)
            if (lane == r && s * GROUP + r < m3) {
                cs64[s * GROUP + r] = ids[r] < 0 ? -INFINITY : acc[r];
                ck64[s * GROUP + r] = ids[r] < 0 ? KEY64_EMPTY : f64_key(acc[r]);
            }
    }
    const long long oq = (long long)blockIdx.y * a.nq + qi;      // blockIdx.y > 0 only with gparts > 1
    for (int e = tid; e < k; e += FIN_THREADS) {          // defaults for slots past the candidates
        if (a.out_s) a.out_s[oq * k + e] = -INFINITY;
        a.out_s64[oq * k + e] = -INFINITY;
        a.out_i[oq * k + e] = -1;
    }
    for (int e = tid; e <= kg; e += FIN_THREADS) sel[e] = -1;
    __syncthreads();

    // ---- final top-k of the re-scored rows: fp64 key descending, ties -> lower row
    rank_select64(ck64, crow, m3, k, sel);
    __syncthreads();
    for (int e = tid; e < k; e += FIN_THREADS) {
        const int c = sel[e];
        if (c >= 0) {
            if (a.out_s) a.out_s[oq * k + e] = (float)cs64[c];
            a.out_s64[oq * k + e] = cs64[c];
            a.out_i[oq * k + e] = (long long)crow[c] + a.row_offset;
        }
    }
    if constexpr (MODE == FIN_FUSED) {
        // certified: nothing was left behind, or the k-th fp64 score clears everything left behind by more than tau
        if (tid == 0) {
            const int c = sel[k - 1];
            const bool cert = bkey == 0u || (c >= 0 && cs64[c] > (double)key_f32(bkey) + tau);
            a.status[qi] = cert ? 0 : 1;
        }
    }
}

// The stand-alone form: 512 threads, four rows' worth of loads in flight per lane.
template <typename Tag, int THREADS, int RS_UNROLL, int MODE>
__global__ __launch_bounds__(THREADS) void finish_topk_kernel(FinishArgs a) {
    finish_topk_body<Tag, THREADS, RS_UNROLL, MODE>(a);
}
// The 256-thread forms are meant to sit beside a resident score-GEMM workgroup: its two waves per SIMD hold 2 x 200 of the
// SIMD's 512 registers, which leaves 112 for this kernel's one wave.  The cap is amdgpu_waves_per_eu(5) = 96 registers;
// the three forms take 95 / 52 / 94 and no scratch (r03's re-score form missed 96 by two and spilled; its row pointers now
// live in scalar registers).  `make` fails if any kernel of the library has a private segment: csrc/check_scratch.py.
template <typename Tag, int RS_UNROLL, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void finish_topk_coop_kernel(FinishArgs a) {
    finish_topk_body<Tag, 256, RS_UNROLL, MODE>(a);
}

// Exhaustive pass of the queries a selection could not certify (status[q] == 1; every other workgroup leaves at once).
// L = the k-th fp64 score found so far (lower[q * lower_stride]; -inf: fewer than k rows found) is a lower bound of
// the true k-th score, so every top-k row has an fp32 score >= L - tau and lies in a group whose maximum is >= L - tau:
// all such groups are re-scored in fp64, 64 groups at a time, against a running top-k (L only rises on the way).
// The result REPLACES the query's top-k; status becomes 2.  In the degenerate case (all scores within tau of each
// other) this is an fp64 brute force over the shard for that query -- slow, finite and exact.
struct ExhaustiveArgs {
    const float* gmax; long long ldg; long long ng;
    const char* Q; long long ldq_b; const char* DB; long long lddb_b;
    long long n; int d; int k; long long row_offset;
    const double* lower; long long lower_stride;
    float* out_s; double* out_s64; long long* out_i;     // [q, k]; out_s / out_s64 may be null
    int* status;
    double tau;
    const float* tau_scale;                                  // as in FinishArgs
    int limited; long long limit0;                           // as in FinishArgs
};
constexpr int EXH_BATCH = 64;                               // groups per re-score batch
constexpr int EXH_POOL = EXH_BATCH * GROUP + DLC_MAX_K;     // running top-k in front of the batch's rows

// (the body: also the tail of small_topk_kernel, whose workgroup carries on with it when its own selection did not certify
// -- no second launch behind every small-database match to find out that nothing is left to do)
template <typename Tag, int RS_UNROLL = 4>
__device__ __forceinline__ void exhaustive_topk_body(const ExhaustiveArgs& a, double lower) {
    const int qi = blockIdx.x;
    __shared__ long long pk[EXH_POOL];
    __shared__ double ps[EXH_POOL];
    __shared__ int pid[EXH_POOL];
    __shared__ int qlist[FIN_THREADS];
    __shared__ int wcnt[FIN_THREADS / 64];
    __shared__ int sel[DLC_MAX_K];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, k = a.k;
    const long long n_vis = visible_rows(a.n, a.limited, a.limit0, qi);
    const long long ng_vis = a.limited ? (n_vis + GROUP - 1) / GROUP : a.ng;
    const double tau = scaled_tau(a.tau, a.tau_scale, qi);
    double theta = lower - tau;                                          // -inf - tau = -inf: everything qualifies
    int cnt = 0;                                                         // rows in the running top-k (pool[0 .. cnt))
    const char* qrow = a.Q + (long long)qi * a.ldq_b;
    for (long long c0 = 0; c0 < ng_vis; c0 += FIN_THREADS) {
        const long long g = c0 + tid;
        const bool qual = g < ng_vis && (double)a.gmax[(long long)qi * a.ldg + g] >= theta;
        const unsigned long long bal = __ballot(qual);
        if (lane == 0) wcnt[w] = __popcll(bal);
        __syncthreads();
        int base = 0, nqual = 0;
#pragma unroll
        for (int ww = 0; ww < FIN_THREADS / 64; ++ww) {
            base += ww < w ? wcnt[ww] : 0;
            nqual += wcnt[ww];
        }
        if (qual) qlist[base + __popcll(bal & ((1ull << lane) - 1ull))] = (int)g;
        __syncthreads();
        for (int b0 = 0; b0 < nqual; b0 += EXH_BATCH) {
            const int nb = min(EXH_BATCH, nqual - b0);
            for (int s = w; s < nb; s += FIN_THREADS / 64) {
                const long long row0 = (long long)qlist[b0 + s] * GROUP;
                const char* rows[GROUP];
#pragma unroll
                for (int r = 0; r < GROUP; ++r) rows[r] = a.DB + (row0 + r < a.n ? row0 + r : a.n - 1) * a.lddb_b;
                double acc[GROUP];
                rescore8_f64<Tag, RS_UNROLL>(qrow, rows, a.d, lane, acc);
#pragma unroll
                for (int r = 0; r < GROUP; ++r)
                    if (lane == r) {
                        const bool ok = row0 + r < n_vis;
                        ps[cnt + s * GROUP + r] = ok ? acc[r] : -INFINITY;
                        pk[cnt + s * GROUP + r] = ok ? f64_key(acc[r]) : KEY64_EMPTY;
                        pid[cnt + s * GROUP + r] = (int)(row0 + r);
                    }
            }
            for (int e = tid; e < k; e += FIN_THREADS) sel[e] = -1;
            __syncthreads();
            const int m = cnt + nb * GROUP;
            rank_select64(pk, pid, m, k, sel);
            __syncthreads();
            // compact the winners to the front of the pool (through registers: k <= 128 < FIN_THREADS)
            long long tk = KEY64_EMPTY; double ts = -INFINITY; int ti = -1;
            if (tid < k && sel[tid] >= 0) { const int c = sel[tid]; tk = pk[c]; ts = ps[c]; ti = pid[c]; }
            const unsigned long long have = __ballot(tid < k && sel[tid] >= 0);
            if (lane == 0) wcnt[w] = __popcll(have);
            __syncthreads();
            if (tid < k) { pk[tid] = tk; ps[tid] = ts; pid[tid] = ti; }
            cnt = 0;
#pragma unroll
            for (int ww = 0; ww < FIN_THREADS / 64; ++ww) cnt += wcnt[ww];
            __syncthreads();
            if (cnt == k && ps[k - 1] - tau > theta) theta = ps[k - 1] - tau;
        }
        __syncthreads();
    }
    for (int e = tid; e < k; e += FIN_THREADS) {
        const bool ok = e < cnt;
        if (a.out_s) a.out_s[(long long)qi * k + e] = ok ? (float)ps[e] : -INFINITY;
        if (a.out_s64) a.out_s64[(long long)qi * k + e] = ok ? ps[e] : -INFINITY;
        a.out_i[(long long)qi * k + e] = ok ? (long long)pid[e] + a.row_offset : -1;
    }
    if (tid == 0) a.status[qi] = 2;
}

template <typename Tag>
__global__ __launch_bounds__(FIN_THREADS) void exhaustive_topk_kernel(ExhaustiveArgs a) {
    const int qi = blockIdx.x;
    if (a.status[qi] != 1) return;
    exhaustive_topk_body<Tag>(a, a.lower[(long long)qi * a.lower_stride]);
}

// The small-database plan in ONE launch per match (databases of <= 16384 rows, k + rslack < SMALL_KT_MAX): a workgroup
// per query sums the split-K partial scores of its row itself (thread t: the groups t, t + 512, ... of 8 rows; fp64 sum
// of the chunks rounded once, exactly splitk_groups_kernel's arithmetic, so the same tau holds), keeps the 32 scores in
// registers, extracts the k + rslack + 1 best ROWS (each wave the best of its slice by repeated DPP arg-max, the
// 8 x (k + rslack + 1) survivors ranked together), re-scores k + rslack of them in fp64 (dealt round the waves) and
// certifies against the one left behind -- and, where that fails, carries on with the exhaustive pass (the group maxima are
// written for it only).  With the
// hierarchical kernel this plan was three launches (partials, splitk_groups_kernel, finish_topk_kernel: half-tile
// selection, group selection, row selection) -- 34 us of a 32-frame batch's 45 in the streaming detector.
struct SmallArgs {
    const float* P; long long ldp; long long qstride; int nsplit;
    float* gmax; long long ldg;
    const char* Q; long long ldq_b; const char* DB; long long lddb_b;
    long long n; int d; int k; int m3max; long long row_offset;
    float* out_s; double* out_s64; long long* out_i; int* status; double tau;
    const float* tau_scale;                                   // as in FinishArgs
    int limited; long long limit0;
};
constexpr int SMALL_KT_MAX = 40;
constexpr int SMALL_GPT = 4;                                  // groups per thread: 512 x 4 x 8 = 16384 rows

// LONG: the instantiation for rows of 32 KB and more (configs[1]'s 75 008-d place descriptors), where the fp64 re-score is
// the kernel (1.9 G conversions + fma for 1063 queries: 160 M vector instructions, 0.30 ms of issue alone on 256 CUs) and
// two waves per SIMD waited on memory for 42 % of their cycles: fewer loads in flight per lane everywhere so that the
// kernel fits 128 registers and TWO workgroups share a CU.
template <typename Tag, bool LONG>
__device__ __forceinline__ void small_topk_body(const SmallArgs& a) {
    __shared__ unsigned long long ckey[(FIN_THREADS / 64) * SMALL_KT_MAX];
    __shared__ long long ck64[SMALL_KT_MAX];
    __shared__ double cs64[SMALL_KT_MAX];
    __shared__ int crow[SMALL_KT_MAX];
    __shared__ int sel[SMALL_KT_MAX + 1];
    const int qi = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, k = a.k;
    const long long n = visible_rows(a.n, a.limited, a.limit0, qi);
    for (int e = tid; e < k; e += FIN_THREADS) {             // defaults for slots past the candidates
        if (a.out_s) a.out_s[(long long)qi * k + e] = -INFINITY;
        a.out_s64[(long long)qi * k + e] = -INFINITY;
        a.out_i[(long long)qi * k + e] = -1;
    }
    for (int e = tid; e <= SMALL_KT_MAX; e += FIN_THREADS) sel[e] = -1;
    if (n <= 0) {                                              // (only with a limit) nothing this query may see
        if (tid == 0) a.status[qi] = 0;
        return;
    }
    // ---- scores of this thread's groups: fp64 sum of the chunk partials, rounded once; keys of the rows it may see
    const long long ng_all = (a.n + GROUP - 1) / GROUP;
    unsigned vk[SMALL_GPT][GROUP];
#pragma unroll
    for (int j = 0; j < SMALL_GPT; ++j) {
        const long long g = tid + (long long)FIN_THREADS * j;
#pragma unroll
        for (int r = 0; r < GROUP; ++r) vk[j][r] = 0u;
        if (g < ng_all) {
            f64x4_t a64 = {0., 0., 0., 0.}, b64 = {0., 0., 0., 0.};
            const float* src = a.P + (long long)qi * a.ldp + g * GROUP;
            constexpr int PU = LONG ? 4 : 16;
            for (int c0 = 0; c0 < a.nsplit; c0 += PU) {      // PU chunks' loads in flight, summed in chunk order
                f32x4_t pa[PU], pb[PU];
#pragma unroll
                for (int u = 0; u < PU; ++u) {
                    const float* sp = src + (long long)(c0 + u < a.nsplit ? c0 + u : c0) * a.qstride;
                    pa[u] = *(const f32x4_t*)(sp);
                    pb[u] = *(const f32x4_t*)(sp + 4);
                }
#pragma unroll
                for (int u = 0; u < PU; ++u)
                    if (c0 + u < a.nsplit) {
                        a64 += __builtin_convertvector(pa[u], f64x4_t);
                        b64 += __builtin_convertvector(pb[u], f64x4_t);
                    }
            }
            const f32x4_t lo = __builtin_convertvector(a64, f32x4_t), hi = __builtin_convertvector(b64, f32x4_t);
            a.gmax[(long long)qi * a.ldg + g] =
                fmaxf(fmaxf(fmaxf(lo[0], lo[1]), fmaxf(lo[2], lo[3])), fmaxf(fmaxf(hi[0], hi[1]), fmaxf(hi[2], hi[3])));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                vk[j][r] = g * GROUP + r < n ? f32_key(lo[r]) : 0u;
                vk[j][r + 4] = g * GROUP + r + 4 < n ? f32_key(hi[r]) : 0u;
            }
        }
    }
    // ---- the kt = k + rslack + 1 best rows: every wave hands in the kt best of its lanes' rows
    const int m3 = (int)(n < (long long)a.m3max ? n : (long long)a.m3max);
    const int kt = (int)(n < (long long)(m3 + 1) ? n : (long long)(m3 + 1));
    unsigned bh = 0u, bl = 0u;                                 // this lane's best key (score, ~row); 0, 0 = none
    unsigned gh[SMALL_GPT], gl[SMALL_GPT];                     // ... and the best of each of its groups
    auto rescan_group = [&](int j) {                           // (j is a constant after unrolling)
        gh[j] = 0u; gl[j] = 0u;
#pragma unroll
        for (int r = 0; r < GROUP; ++r)                        // rows ascending: the first maximum is the lowest row
            if (vk[j][r] > gh[j]) { gh[j] = vk[j][r]; gl[j] = ~(unsigned)((tid + FIN_THREADS * j) * GROUP + r); }
    };
    auto combine = [&]() {
        bh = 0u; bl = 0u;
#pragma unroll
        for (int j = 0; j < SMALL_GPT; ++j)
            if (gh[j] > bh) { bh = gh[j]; bl = gl[j]; }
    };
#pragma unroll
    for (int j = 0; j < SMALL_GPT; ++j) rescan_group(j);
    combine();
    for (int it = 0; it < kt; ++it) {
        const unsigned mh = wave_max_u32(bh);
        const unsigned ml = wave_max_u32(bh == mh ? bl : 0u);
        if (lane == 0) ckey[w * kt + it] = mh == 0u ? 0ull : (((unsigned long long)mh << 32) | ml);
        if (mh != 0u && bh == mh && bl == ml) {                // the owner retires it and finds its next best
            const unsigned row = ~ml;
#pragma unroll
            for (int j = 0; j < SMALL_GPT; ++j)
                if ((row >> 3) == (unsigned)(tid + FIN_THREADS * j)) {
#pragma unroll
                    for (int r = 0; r < GROUP; ++r)
                        if ((row & 7u) == (unsigned)r) vk[j][r] = 0u;
                    rescan_group(j);
                }
            combine();
        }
    }
    __syncthreads();
    rank_select(ckey, (FIN_THREADS / 64) * kt, kt, sel);
    __syncthreads();
    // the best row left behind bounds every row that is not re-scored (0 = every row the query may see is a candidate)
    const unsigned bkey = (kt > m3 && sel[m3] >= 0) ? (unsigned)(ckey[sel[m3]] >> 32) : 0u;
    if (tid < m3) crow[tid] = sel[tid] < 0 ? -1 : (int)key_id(ckey[sel[tid]]);
    __syncthreads();
    // ---- fp64 re-score, final top-k (fp64 key descending, ties -> lower row), certificate
    // (candidates dealt round the waves, a wave's two at a time with eight pieces of each in flight: these are dependent
    // round trips to memory, and with k + rslack = 9 one wave has two rows)
    // Long rows (>= 32 KB) are bandwidth, not latency: there a wave takes eight candidates at a time and reads the query
    // once for them (1063 queries x 24 rows of 150 KB: dealt in pairs every wave read the query again, +18 % time).
    const char* qrow = a.Q + (long long)qi * a.ldq_b;
    constexpr int NW = FIN_THREADS / 64;
    if constexpr (LONG) {
        // THREE candidates per wave and pass: k + rslack = 24 candidates are one pass of all eight waves, the query read once
        // per wave.  (Eight per wave -- r04 -- left five of the eight waves idle and each of the three busy ones alternating
        // between 36 loads in flight and the 2 300 conversions + fma of a batch with nothing in flight: 0.57 ms for configs[1]'s
        // 1063 queries at an L2 hit rate of 0.44 and 3.8 TB/s over the fabric -- latency, not bytes.  Pairs -- the form
        // below -- take two passes for 24.)  A candidate's value is rescore8_f64's whatever the dealing: the same pieces
        // per lane in the same order, the same butterfly.
        for (int s3 = w; s3 * 3 < m3; s3 += NW) {
            int ids[3];
            const char* rows[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                ids[r] = __builtin_amdgcn_readfirstlane(s3 * 3 + r < m3 ? crow[s3 * 3 + r] : -1);
                rows[r] = a.DB + (long long)(ids[r] < 0 ? 0 : ids[r]) * a.lddb_b;
            }
            double acc[3];
            rescore_rows_f64<Tag, 3, 4>(qrow, rows, a.d, lane, acc);
#pragma unroll
            for (int r = 0; r < 3; ++r)
                if (lane == r && s3 * 3 + r < m3) {
                    cs64[s3 * 3 + r] = ids[r] < 0 ? -INFINITY : acc[r];
                    ck64[s3 * 3 + r] = ids[r] < 0 ? KEY64_EMPTY : f64_key(acc[r]);
                }
        }
    } else {
    int ci = w;
    for (; ci + NW < m3; ci += 2 * NW) {
        const int id0 = __builtin_amdgcn_readfirstlane(crow[ci]), id1 = __builtin_amdgcn_readfirstlane(crow[ci + NW]);
        const char* const rows[2] = {a.DB + (long long)(id0 < 0 ? 0 : id0) * a.lddb_b,
                                     a.DB + (long long)(id1 < 0 ? 0 : id1) * a.lddb_b};
        double sc[2];
        rescore_rows_f64<Tag, 2, 8>(qrow, rows, a.d, lane, sc);
        if (lane == 0) {
            cs64[ci] = id0 < 0 ? -INFINITY : sc[0];
            ck64[ci] = id0 < 0 ? KEY64_EMPTY : f64_key(sc[0]);
            cs64[ci + NW] = id1 < 0 ? -INFINITY : sc[1];
            ck64[ci + NW] = id1 < 0 ? KEY64_EMPTY : f64_key(sc[1]);
        }
    }
    if (ci < m3) {
        const int id = __builtin_amdgcn_readfirstlane(crow[ci]);
        const char* const rows[1] = {a.DB + (long long)(id < 0 ? 0 : id) * a.lddb_b};
        double sc[1];
        rescore_rows_f64<Tag, 1, 8>(qrow, rows, a.d, lane, sc);
        if (lane == 0) {
            cs64[ci] = id < 0 ? -INFINITY : sc[0];
            ck64[ci] = id < 0 ? KEY64_EMPTY : f64_key(sc[0]);
        }
    }
    }
    for (int e = tid; e <= SMALL_KT_MAX; e += FIN_THREADS) sel[e] = -1;
    __syncthreads();
    rank_select64(ck64, crow, m3, k, sel);
    __syncthreads();
    for (int e = tid; e < k; e += FIN_THREADS) {
        const int c = sel[e];
        if (c >= 0) {
            if (a.out_s) a.out_s[(long long)qi * k + e] = (float)cs64[c];
            a.out_s64[(long long)qi * k + e] = cs64[c];
            a.out_i[(long long)qi * k + e] = (long long)crow[c] + a.row_offset;
        }
    }
    // certified: nothing was left behind, or the k-th fp64 score clears everything left behind by more than tau (uniform:
    // every thread reads the same LDS words).  Otherwise this workgroup runs the exhaustive pass itself.
    const int ck = sel[k - 1];
    const double kth = ck >= 0 ? cs64[ck] : -INFINITY;
    const bool cert = bkey == 0u || (ck >= 0 && kth > (double)key_f32(bkey) + scaled_tau(a.tau, a.tau_scale, qi));
    if (cert) {
        if (tid == 0) a.status[qi] = 0;
        return;
    }
    __syncthreads();                                           // the group maxima and the lists above are written; LDS is re-used
    ExhaustiveArgs e;
    e.gmax = a.gmax; e.ldg = a.ldg; e.ng = ng_all;
    e.Q = a.Q; e.ldq_b = a.ldq_b; e.DB = a.DB; e.lddb_b = a.lddb_b;
    e.n = a.n; e.d = a.d; e.k = k; e.row_offset = a.row_offset;
    e.lower = nullptr; e.lower_stride = 0;
    e.out_s = a.out_s; e.out_s64 = a.out_s64; e.out_i = a.out_i;
    e.status = a.status; e.tau = a.tau; e.tau_scale = a.tau_scale;
    e.limited = a.limited; e.limit0 = a.limit0;
    exhaustive_topk_body<Tag, LONG ? 1 : 4>(e, kth);
}

template <typename Tag>
__global__ __launch_bounds__(FIN_THREADS) void small_topk_kernel(SmallArgs a) {
    small_topk_body<Tag, false>(a);
}
template <typename Tag>
__global__ __launch_bounds__(FIN_THREADS) __attribute__((amdgpu_waves_per_eu(4))) void small_topk_long_kernel(SmallArgs a) {
    small_topk_body<Tag, true>(a);
}

// Global top-k from [parts, q, k] per-shard results (the all-gather layout); idx < 0 = empty slot.
// Rank-by-counting on (fp64 ordering key, id): ids are unique, so the order is total -- and it is the order every
// shard used for its own list.  With `bound` (the largest fp32 score a row outside all shards' candidates can have,
// dlc_cosine_rescore_topk) the merge also certifies: status[q] = 0 when the k-th score clears bound[q] by more
// than tau, 1 otherwise.
__global__ __launch_bounds__(FIN_THREADS) void merge_topk_kernel(const double* __restrict__ pscores,
                                                                 long long s_stride,
                                                                 const long long* __restrict__ pidx,
                                                                 long long i_stride, int parts, long long nq, int k,
                                                                 const float* __restrict__ bound, double tau,
                                                                 const float* __restrict__ tau_scale,
                                                                 float* __restrict__ out_s,
                                                                 double* __restrict__ out_s64,
                                                                 long long* __restrict__ out_i,
                                                                 int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    __shared__ double kth;
    __shared__ int have_kth;
    const int qi = blockIdx.x, tid = threadIdx.x;
    const int m = parts * k;
    long long* ci = (long long*)dsm;
    long long* ck = (long long*)(dsm + (size_t)m * 8);
    double* cs = (double*)(dsm + (size_t)m * 16);
    if (tid == 0) { kth = -INFINITY; have_kth = 0; }
    for (int e = tid; e < m; e += FIN_THREADS) {
        const long long o = (long long)qi * k + (e % k);
        const long long id = pidx[(long long)(e / k) * i_stride + o];
        const double s = pscores[(long long)(e / k) * s_stride + o];
        ci[e] = id; cs[e] = s;
        ck[e] = id < 0 ? KEY64_EMPTY : f64_key(s);
    }
    for (int e = tid; e < k; e += FIN_THREADS) {
        if (out_s) out_s[(long long)qi * k + e] = -INFINITY;
        if (out_s64) out_s64[(long long)qi * k + e] = -INFINITY;
        out_i[(long long)qi * k + e] = -1;
    }
    __syncthreads();
    for (int e = tid; e < m; e += FIN_THREADS) {
        const long long ke = ck[e];
        const long long ie = ci[e];
        if (ke == KEY64_EMPTY) continue;
        int rank = 0;
        for (int j = 0; j < m; ++j) {
            const long long kj = ck[j];
            rank += (kj > ke || (kj == ke && kj != KEY64_EMPTY && ci[j] < ie)) ? 1 : 0;
        }
        if (rank < k) {
            if (out_s) out_s[(long long)qi * k + rank] = (float)cs[e];
            if (out_s64) out_s64[(long long)qi * k + rank] = cs[e];
            out_i[(long long)qi * k + rank] = ie;
            if (rank == k - 1) { kth = cs[e]; have_kth = 1; }
        }
    }
    if (status) {
        __syncthreads();
        if (tid == 0) {
            const float b = bound ? bound[qi] : -INFINITY;
            status[qi] = (b == -INFINITY || (have_kth && kth > (double)b + scaled_tau(tau, tau_scale, qi))) ? 0 : 1;
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

// Split-K second pass.  Sums the chunk partials in chunk order (deterministic; in fp64, rounded once) and applies the
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
        // chunk partials summed in fp64 and rounded ONCE: the error of a split score against the fp64 re-score is then
        // that of one chunk's MFMA chain (the tau of the selection's certificate), not of the whole row's
        f64x4_t a64 = {0., 0., 0., 0.}, b64 = {0., 0., 0., 0.};
        float* dst0 = P + (long long)qi * ldp + g * GROUP;
        const float* src = dst0;
        for (int c = 0; c < nsplit; ++c) {
            a64 += __builtin_convertvector(*(const f32x4_t*)(src), f64x4_t);
            b64 += __builtin_convertvector(*(const f32x4_t*)(src + 4), f64x4_t);
            src += (long long)q * ldp;
        }
        const f32x4_t a = __builtin_convertvector(a64, f32x4_t), b = __builtin_convertvector(b64, f32x4_t);
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
    f64x4_t a64 = {0., 0., 0., 0.};                       // as splitk_groups_kernel: fp64 sum of the chunks, one rounding
    const float* src = P + (long long)qi * ldp + r0;
    for (int c = 0; c < nsplit; ++c) {
        a64 += __builtin_convertvector(*(const f32x4_t*)src, f64x4_t);
        src += (long long)q * ldp;
    }
    const f32x4_t a = __builtin_convertvector(a64, f32x4_t);
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
    size_t gmax, tmax, part, status, s64, rs_ids, rs_max, rs_scores, rs_idx, rs_bound, total;
    long long ldg, ldt, ldp;
    int kg;
    SplitPlan sp;
    int rparts;     // > 1: dlc_cosine_topk re-scores with this many workgroups per query (few queries, long rows)
    bool dense;     // small database: the score matrix itself is kept (chunk 0 of `part`) and the candidates are ROWS
    double tau;     // error bound of the score pass against the fp64 re-score (score_error_bound below)
};

// Few queries take the bandwidth kernel (stored queries in LDS: q <= 4 and q * d * 2 bytes <= 64 KiB).
inline bool gemv_shape(int64_t q, int64_t nk) {
    const int qb = q <= 1 ? 1 : (q <= 2 ? 2 : 4);
    return q <= GEMV_MAX_Q && (long long)qb * nk * 128 <= GEMV_MAX_LDS;
}

// Small-database plan.  The standard plan never writes the score matrix: it keeps group maxima and
// re-scores the kg * 8 rows of the selected groups in fp64 -- a gather of kg * 8 * d * 2 bytes per
// query whatever the database size.  Against a small database (the reference's own scale: 1063
// key-frames x 75 000-d, where that gather is 192 of the 1063 rows, 29 MB per query, 30 GB per
// call) the whole fp32 score matrix is cheap (q * n * 4 bytes) and lets the selection pick its
// candidates by ROW: the k + DENSE_ROW_SLACK best fp32 scores of the selected groups are re-scored
// (24 rows per query at k = 20 instead of 192).
constexpr int64_t DENSE_MAX_ROWS = 16384;
constexpr int DENSE_ROW_SLACK = 4;
// (a handful of queries otherwise take the bandwidth kernel and the hierarchical selection: right for a database that
// has to be streamed from HBM, three dependent launches of 13 + 29 + 5 us for a frame against 1000 resident key-frames --
// up to DENSE_GEMV_BYTES of database they take this plan too: 11 + 9 us)
constexpr int64_t DENSE_GEMV_BYTES = 32ll << 20;
inline bool dense_plan(int64_t q, int64_t n, int64_t d) {
    return n <= DENSE_MAX_ROWS && (!gemv_shape(q, d / BK) || n * d * 2 <= DENSE_GEMV_BYTES);
}

// One workgroup per query gathers kg * 8 rows: with a handful of queries and long rows (1 query x
// 75 000-d: 29 MB through one CU, 300 us) the re-score is spread over one workgroup per selected group.
int rescore_parts(int64_t q, int64_t d, int kg, int k) {
    if (q > 32 || (int64_t)kg * GROUP * d * 2 < (2 << 20)) return 1;
    return std::min(kg, 2048 / k);            // the merge of the parts keeps parts * k 24-byte entries in 48 KiB of LDS
}

// |fp32 score of the score pass - fp64 re-score| <= tau for stored rows of norm <= 1.005 (what dlc_l2_normalize_rows
// writes: unit rows rounded to bf16 / fp16).  One MFMA step (32 exact products added to the accumulator) or one v_dot2
// step (2 products) is taken to err by at most SCORE_STEP_EPS * (|accumulator| + sum |products|) -- twice the
// half-ulp of a single correctly rounded fp32 result, measured on the device by tests/test_gpu_parity.py::
// test_score_error_bound_holds; partial sums are bounded by sum |q_i x_i| <= |q| |x| <= 1.01 (Cauchy-Schwarz), a
// chain of s steps therefore errs by at most (s + 1) * SCORE_STEP_EPS * 1.01, split-K chunks are summed in fp64 and
// rounded once (+1 step), and 2^-38 covers the quantisation of the ordering key.
constexpr double SCORE_STEP_EPS = 1.1920928955078125e-07;     // 2^-23
double score_error_bound(int64_t q, int64_t nk, const SplitPlan& sp, bool dense) {
    double steps;
    if (!dense && gemv_shape(q, nk)) steps = (double)nk * 0.5 + 8.0;   // a lane's chain of nk/2 v_dot2 steps + 6 butterfly adds
    else steps = 2.0 * (double)sp.kchunk + 2.0;                        // two MFMA k-slices per K tile of the chunk
    return SCORE_STEP_EPS * 1.01 * steps + 3.7e-12;
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
    w.tau = score_error_bound(q, d / BK, w.sp, w.dense);
    w.ldp = ntiles * BM;
    w.part = o;
    if (w.sp.nsplit > 1 || w.dense) o += dlc::align_up((size_t)w.sp.nsplit * q * w.ldp * 4, 256);
    w.status = o; o += dlc::align_up((size_t)q * 4, 256);
    w.s64 = o; o += dlc::align_up((size_t)q * k * 8, 256);        // fp64 scores when the caller does not ask for them
    w.rparts = w.dense ? 1 : rescore_parts(q, d, w.kg, k);
    w.rs_ids = w.rs_max = w.rs_scores = w.rs_idx = w.rs_bound = o;
    if (w.rparts > 1) {
        w.rs_ids = o; o += dlc::align_up((size_t)q * w.kg * 4, 256);
        w.rs_max = o; o += dlc::align_up((size_t)q * (w.kg + 1) * 4, 256);
        w.rs_scores = o; o += dlc::align_up((size_t)w.rparts * q * k * 8, 256);
        w.rs_idx = o; o += dlc::align_up((size_t)w.rparts * q * k * 8, 256);
        w.rs_bound = o; o += dlc::align_up((size_t)q * 4, 256);
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
int launch_scores(dlc_ctx* ctx, const GemmArgs& a, bool dense, hipStream_t st, bool keep = false, bool partials_only = false) {
    if (!dense && !keep && use_gemv(a)) return launch_gemv<Tag>(ctx, a, st);
    if (a.nsplit <= 1 && !keep)
        return dense ? launch_gemm<Tag, GEMM_DENSE>(ctx, a, st) : launch_gemm<Tag, GEMM_GROUPS>(ctx, a, st);
    int rc = launch_gemm<Tag, GEMM_PARTIAL>(ctx, a, st);
    if (rc != DLC_OK) return rc;
    if (partials_only) return DLC_OK;                        // small_topk_kernel sums the chunks itself
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

// The small-database plan as one selection launch (small_topk_kernel) instead of the reducing pass + the hierarchical kernel.
inline bool small_direct(const MatchCall& mc, int k) {
    return mc.w.dense && mc.a.n <= (int64_t)FIN_THREADS * SMALL_GPT * GROUP && k + DENSE_ROW_SLACK + 1 <= SMALL_KT_MAX;
}

int run_score(dlc_ctx* ctx, int dtype, MatchCall& mc, hipStream_t st, bool partials_only = false) {
    const int slot = (int)(ctx->prof_calls % DLC_PROFILE_RING);
    if (ctx->profiling) DLC_HIP_CHECK(ctx, hipEventRecord(ctx->ev_start[slot], st));
    int rc = (dtype == DLC_BF16) ? launch_scores<dlc_bf16_tag>(ctx, mc.a, false, st, mc.w.dense, partials_only)
                                 : launch_scores<dlc_f16_tag>(ctx, mc.a, false, st, mc.w.dense, partials_only);
    if (rc != DLC_OK) return rc;
    if (ctx->profiling) {
        DLC_HIP_CHECK(ctx, hipEventRecord(ctx->ev_stop[slot], st));
        ctx->prof_calls++;
    }
    return DLC_OK;
}

// Arguments every selection kernel of a prepared match shares.
FinishArgs finish_args(const MatchCall& mc, int k, int64_t n, int64_t d, int64_t q, int64_t row_offset) {
    const GemmArgs& g = mc.a;
    FinishArgs f{};
    f.tmax = g.tmax; f.ldt = g.ldt; f.nh = (int)g.nh;
    f.gmax = g.gmax; f.ldg = g.ldg; f.ng = g.ng;
    f.kg = mc.w.kg;
    f.Q = g.Q; f.ldq_b = g.ldq_b; f.DB = g.DB; f.lddb_b = g.lddb_b;
    f.n = n; f.d = (int)d; f.k = k; f.row_offset = row_offset;
    f.tau = mc.w.tau;
    f.nq = q;
    return f;
}

template <typename Tag, int THREADS, int RS_UNROLL, int MODE>
int launch_finish(dlc_ctx* ctx, FinishArgs f, int64_t q, bool small_lds, hipStream_t st) {
    size_t dsm = fin_lds_fixed(f.kg);
    f.tv_in_lds = MODE != FIN_RESCORE && !small_lds && (size_t)f.nh * 4 <= 96 * 1024;
    if (f.tv_in_lds) dsm += (size_t)f.nh * 4;
    dsm = dlc::align_up(dsm, 16);
    void (*fk)(FinishArgs);
    if constexpr (THREADS == 256) fk = finish_topk_coop_kernel<Tag, RS_UNROLL, MODE>;
    else fk = finish_topk_kernel<Tag, THREADS, RS_UNROLL, MODE>;
    if (dsm > 48 * 1024)
        DLC_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)fk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dsm));
    hipLaunchKernelGGL(fk, dim3((unsigned)q, (unsigned)(MODE == FIN_RESCORE && f.gparts > 1 ? f.gparts : 1)), dim3(THREADS),
                       dsm, st, f);
    DLC_LAUNCH_CHECK(ctx, "finish_topk_kernel");
    return DLC_OK;
}

template <int MODE>
int run_select(dlc_ctx* ctx, int dtype, const FinishArgs& f, int64_t q, int flags, hipStream_t st) {
    const bool coop = (flags & DLC_SELECT_COOP) != 0;
    if (dtype == DLC_BF16)
        return coop ? launch_finish<dlc_bf16_tag, 256, 1, MODE>(ctx, f, q, true, st)
                    : launch_finish<dlc_bf16_tag, 512, 4, MODE>(ctx, f, q, false, st);
    return coop ? launch_finish<dlc_f16_tag, 256, 1, MODE>(ctx, f, q, true, st)
                : launch_finish<dlc_f16_tag, 512, 4, MODE>(ctx, f, q, false, st);
}

// The exhaustive pass behind a certifying kernel: every workgroup whose query is certified leaves at once.
int run_exhaustive(dlc_ctx* ctx, int dtype, const FinishArgs& f, int64_t q, const double* lower, int64_t lower_stride,
                   float* out_s, double* out_s64, int64_t* out_i, int* status, hipStream_t st) {
    ExhaustiveArgs e{};
    e.gmax = f.gmax; e.ldg = f.ldg; e.ng = f.ng;
    e.Q = f.Q; e.ldq_b = f.ldq_b; e.DB = f.DB; e.lddb_b = f.lddb_b;
    e.n = f.n; e.d = f.d; e.k = f.k; e.row_offset = f.row_offset;
    e.lower = lower; e.lower_stride = lower_stride;
    e.out_s = out_s; e.out_s64 = out_s64; e.out_i = (long long*)out_i;
    e.status = status; e.tau = f.tau; e.tau_scale = f.tau_scale;
    e.limited = f.limited; e.limit0 = f.limit0;
    if (dtype == DLC_BF16)
        hipLaunchKernelGGL(exhaustive_topk_kernel<dlc_bf16_tag>, dim3((unsigned)q), dim3(FIN_THREADS), 0, st, e);
    else
        hipLaunchKernelGGL(exhaustive_topk_kernel<dlc_f16_tag>, dim3((unsigned)q), dim3(FIN_THREADS), 0, st, e);
    DLC_LAUNCH_CHECK(ctx, "exhaustive_topk_kernel");
    return DLC_OK;
}

int launch_merge(dlc_ctx* ctx, const double* scores, int64_t score_part_stride, const int64_t* idx, int64_t idx_part_stride,
                 int parts, int64_t q, int k, const float* bound, double tau, const float* tau_scale, float* out_scores,
                 double* out_scores_f64, int64_t* out_idx, int* status, hipStream_t st) {
    const size_t m = (size_t)parts * k;
    const size_t dsm = m * 24;
    if (dsm > 48 * 1024) return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "topk_merge: parts*k=%zu too large (max 2048)", m);
    hipLaunchKernelGGL(merge_topk_kernel, dim3((unsigned)q), dim3(FIN_THREADS), dsm, st, scores, (long long)score_part_stride,
                       (const long long*)idx, (long long)idx_part_stride, parts, (long long)q, k, bound, tau, tau_scale,
                       out_scores, out_scores_f64, (long long*)out_idx, status);
    DLC_LAUNCH_CHECK(ctx, "merge_topk_kernel");
    return DLC_OK;
}

// Stage 2 of a match from a filled workspace: selection, fp64 re-score, final top-k, certification, exhaustive pass.
int run_finish(dlc_ctx* ctx, int dtype, const MatchCall& mc, int k, int64_t n, int64_t d, int64_t q, int64_t row_offset,
               float* out_scores, double* out_scores_f64, int64_t* out_idx, int32_t* out_status, void* workspace, int flags,
               const float* tau_scale, hipStream_t st, int limited = 0, int64_t limit0 = 0, bool direct = false) {
    char* ws = (char*)workspace;
    FinishArgs f = finish_args(mc, k, n, d, q, row_offset);
    f.limited = limited; f.limit0 = limit0; f.tau_scale = tau_scale;
    int* status = out_status ? out_status : (int*)(ws + mc.w.status);
    double* s64 = out_scores_f64 ? out_scores_f64 : (double*)(ws + mc.w.s64);
    int rc;
    if (direct) {
        SmallArgs sa{};
        sa.P = mc.a.P; sa.ldp = mc.a.ldp; sa.qstride = (long long)q * mc.a.ldp; sa.nsplit = mc.a.nsplit;
        sa.gmax = mc.a.gmax; sa.ldg = mc.a.ldg;
        sa.Q = f.Q; sa.ldq_b = f.ldq_b; sa.DB = f.DB; sa.lddb_b = f.lddb_b;
        sa.n = n; sa.d = (int)d; sa.k = k; sa.m3max = k + DENSE_ROW_SLACK; sa.row_offset = row_offset;
        sa.out_s = out_scores; sa.out_s64 = s64; sa.out_i = (long long*)out_idx; sa.status = status; sa.tau = mc.w.tau;
        sa.tau_scale = tau_scale; sa.limited = limited; sa.limit0 = limit0;
        const bool long_rows = d >= 16384;                      // (rows of 32 KB and more: the re-score IS the kernel there)
        if (dtype == DLC_BF16) {
            if (long_rows) hipLaunchKernelGGL(small_topk_long_kernel<dlc_bf16_tag>, dim3((unsigned)q), dim3(FIN_THREADS), 0, st, sa);
            else hipLaunchKernelGGL(small_topk_kernel<dlc_bf16_tag>, dim3((unsigned)q), dim3(FIN_THREADS), 0, st, sa);
        } else {
            if (long_rows) hipLaunchKernelGGL(small_topk_long_kernel<dlc_f16_tag>, dim3((unsigned)q), dim3(FIN_THREADS), 0, st, sa);
            else hipLaunchKernelGGL(small_topk_kernel<dlc_f16_tag>, dim3((unsigned)q), dim3(FIN_THREADS), 0, st, sa);
        }
        DLC_LAUNCH_CHECK(ctx, "small_topk_kernel");
    } else if (mc.w.rparts <= 1) {
        f.out_s = out_scores; f.out_s64 = s64; f.out_i = (long long*)out_idx; f.status = status;
        if (mc.w.dense) { f.dense_S = mc.a.P; f.ld_s = mc.a.ldp; f.rslack = DENSE_ROW_SLACK; }
        rc = run_select<FIN_FUSED>(ctx, dtype, f, q, flags, st);
        if (rc != DLC_OK) return rc;
    } else {
        // few queries, long rows: group selection, re-score with one workgroup per selected group, certifying merge
        f.grp_ids = (int*)(ws + mc.w.rs_ids); f.grp_max = (float*)(ws + mc.w.rs_max);
        rc = run_select<FIN_GROUPS>(ctx, dtype, f, q, 0, st);
        if (rc != DLC_OK) return rc;
        double* ps = (double*)(ws + mc.w.rs_scores);
        int64_t* pi = (int64_t*)(ws + mc.w.rs_idx);
        float* pb = (float*)(ws + mc.w.rs_bound);
        f.gparts = mc.w.rparts; f.out_s = nullptr; f.out_s64 = ps; f.out_i = (long long*)pi; f.bound_out = pb;
        rc = run_select<FIN_RESCORE>(ctx, dtype, f, q, DLC_SELECT_COOP, st);
        if (rc != DLC_OK) return rc;
        rc = launch_merge(ctx, ps, q * k, pi, q * k, mc.w.rparts, q, k, pb, mc.w.tau, tau_scale, out_scores, s64, out_idx, status, st);
        if (rc != DLC_OK) return rc;
    }
    if (direct) return DLC_OK;                                  // small_topk_kernel runs the exhaustive pass of its own queries
    return run_exhaustive(ctx, dtype, f, q, s64 + (k - 1), k, out_scores, s64, out_idx, status, st);
}

}  // namespace

extern "C" double dlc_cosine_score_error_bound(int64_t q, int64_t n, int64_t d, int k) {
    if (q < 1 || n < 1 || d < BK || k < 1 || k > DLC_MAX_K) return 0.0;
    return ws_layout(q, n, d, k).tau;
}

extern "C" int dlc_cosine_topk(dlc_ctx* ctx, int dtype, const void* Q, int64_t q, int64_t ldq, const void* DB,
                               int64_t n, int64_t lddb, int64_t d, int k, int64_t row_offset, float* out_scores,
                               double* out_scores_f64, int64_t* out_idx, int32_t* out_status, const float* tau_scale,
                               void* workspace, size_t workspace_bytes, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!out_scores || !out_idx) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "cosine_topk: null output");
    MatchCall mc;
    int rc = prepare_match(ctx, "cosine_topk", dtype, Q, q, ldq, DB, n, lddb, d, k, workspace, workspace_bytes, &mc);
    if (rc != DLC_OK) return rc;
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    const bool direct = small_direct(mc, k);
    rc = run_score(ctx, dtype, mc, (hipStream_t)stream, direct);
    if (rc != DLC_OK) return rc;
    return run_finish(ctx, dtype, mc, k, n, d, q, row_offset, out_scores, out_scores_f64, out_idx, out_status, workspace, 0,
                      tau_scale, (hipStream_t)stream, 0, 0, direct);
}

extern "C" int dlc_cosine_topk_older(dlc_ctx* ctx, int dtype, const void* Q, int64_t q, int64_t ldq, const void* DB,
                                     int64_t n, int64_t lddb, int64_t d, int k, int64_t row_offset, int64_t limit0,
                                     float* out_scores, double* out_scores_f64, int64_t* out_idx, int32_t* out_status,
                                     const float* tau_scale, void* workspace, size_t workspace_bytes, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!out_scores || !out_idx) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "cosine_topk_older: null output");
    MatchCall mc;
    int rc = prepare_match(ctx, "cosine_topk_older", dtype, Q, q, ldq, DB, n, lddb, d, k, workspace, workspace_bytes, &mc);
    if (rc != DLC_OK) return rc;
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    const bool direct = small_direct(mc, k);
    rc = run_score(ctx, dtype, mc, (hipStream_t)stream, direct);
    if (rc != DLC_OK) return rc;
    return run_finish(ctx, dtype, mc, k, n, d, q, row_offset, out_scores, out_scores_f64, out_idx, out_status, workspace, 0,
                      tau_scale, (hipStream_t)stream, 1, limit0, direct);
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
                                      double* out_scores_f64, int64_t* out_idx, int32_t* out_status,
                                      const float* tau_scale, void* workspace, size_t workspace_bytes, int flags,
                                      void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!out_scores || !out_idx) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "cosine_select_topk: null output");
    MatchCall mc;
    int rc = prepare_match(ctx, "cosine_select_topk", dtype, Q, q, ldq, DB, n, lddb, d, k, workspace, workspace_bytes, &mc);
    if (rc != DLC_OK) return rc;
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    return run_finish(ctx, dtype, mc, k, n, d, q, row_offset, out_scores, out_scores_f64, out_idx, out_status, workspace, flags,
                      tau_scale, (hipStream_t)stream);
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
    FinishArgs f = finish_args(mc, k, n, d, q, 0);
    f.grp_ids = group_ids; f.grp_max = group_max;
    return run_select<FIN_GROUPS>(ctx, dtype, f, q, flags, (hipStream_t)stream);
}

namespace {
// A MatchCall for the calls that need the operands and the plan's constants but no workspace.
int operands_only(dlc_ctx* ctx, const char* what, int dtype, const void* Q, int64_t q, int64_t ldq, const void* DB, int64_t n,
                  int64_t lddb, int64_t d, int k, MatchCall* mc) {
    int rc = check_operands(ctx, dtype, Q, q, ldq, DB, n, lddb, d);
    if (rc != DLC_OK) return rc;
    if (k < 1 || k > DLC_MAX_K) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "%s: k=%d outside 1..%d", what, k, DLC_MAX_K);
    mc->w = ws_layout(q, n, d, k);
    mc->a = GemmArgs{};
    mc->a.Q = (const char*)Q; mc->a.DB = (const char*)DB;
    mc->a.ldq_b = ldq * 2; mc->a.lddb_b = lddb * 2;
    mc->a.q = (int)q; mc->a.n = n; mc->a.nk = (int)(d / BK);
    mc->a.ng = dlc::cdiv(n, GROUP); mc->a.nh = dlc::cdiv(n, HALF);
    return DLC_OK;
}
}  // namespace

extern "C" int dlc_cosine_rescore_topk(dlc_ctx* ctx, int dtype, const void* Q, int64_t q, int64_t ldq, const void* DB,
                                       int64_t n, int64_t lddb, int64_t d, int k, int64_t row_offset,
                                       const int32_t* group_ids, const float* group_max, const float* all_group_max,
                                       int parts, double* out_scores_f64, int64_t* out_idx, float* out_bound,
                                       const float* tau_scale, int flags, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!out_scores_f64 || !out_idx || !group_ids || !group_max || parts < 0 || (parts > 0 && !all_group_max))
        return dlc::fail(ctx, DLC_ERR_BAD_ARG, "cosine_rescore_topk: bad argument");
    MatchCall mc;
    int rc = operands_only(ctx, "cosine_rescore_topk", dtype, Q, q, ldq, DB, n, lddb, d, k, &mc);
    if (rc != DLC_OK) return rc;
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    FinishArgs f = finish_args(mc, k, n, d, q, row_offset);
    f.grp_ids = const_cast<int32_t*>(group_ids); f.grp_max = const_cast<float*>(group_max);
    f.all_max = all_group_max; f.parts = parts;
    f.out_s = nullptr; f.out_s64 = out_scores_f64; f.out_i = (long long*)out_idx; f.bound_out = out_bound;
    f.tau_scale = tau_scale;
    return run_select<FIN_RESCORE>(ctx, dtype, f, q, flags, (hipStream_t)stream);
}

extern "C" int dlc_cosine_exhaustive_topk(dlc_ctx* ctx, int dtype, const void* Q, int64_t q, int64_t ldq, const void* DB,
                                          int64_t n, int64_t lddb, int64_t d, int k, int64_t row_offset,
                                          const double* lower, int64_t lower_stride, double tau,
                                          const float* tau_scale, int32_t* status, float* out_scores, double* out_scores_f64, int64_t* out_idx, void* workspace,
                                          size_t workspace_bytes, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!lower || lower_stride < 0 || !status || !out_idx || !(tau >= 0.0))
        return dlc::fail(ctx, DLC_ERR_BAD_ARG, "cosine_exhaustive_topk: bad argument");
    MatchCall mc;
    int rc = prepare_match(ctx, "cosine_exhaustive_topk", dtype, Q, q, ldq, DB, n, lddb, d, k, workspace, workspace_bytes, &mc);
    if (rc != DLC_OK) return rc;
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    FinishArgs f = finish_args(mc, k, n, d, q, row_offset);
    f.tau = tau; f.tau_scale = tau_scale;
    return run_exhaustive(ctx, dtype, f, q, lower, lower_stride, out_scores, out_scores_f64, out_idx, status, (hipStream_t)stream);
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

extern "C" int dlc_topk_merge_strided(dlc_ctx* ctx, const double* scores_f64, int64_t score_part_stride, const int64_t* idx,
                                      int64_t idx_part_stride, int parts, int64_t q, int k, const float* bound, double tau,
                                      const float* tau_scale, float* out_scores, double* out_scores_f64, int64_t* out_idx,
                                      int32_t* out_status, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!scores_f64 || !idx || !out_idx || parts < 1 || q < 1 || k < 1 || k > DLC_MAX_K ||
        score_part_stride < q * k || idx_part_stride < q * k || !(tau >= 0.0))
        return dlc::fail(ctx, DLC_ERR_BAD_ARG, "topk_merge: bad argument");
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    return launch_merge(ctx, scores_f64, score_part_stride, idx, idx_part_stride, parts, q, k, bound, tau, tau_scale,
                        out_scores, out_scores_f64, out_idx, out_status, (hipStream_t)stream);
}

extern "C" int dlc_topk_merge(dlc_ctx* ctx, const double* scores_f64, const int64_t* idx, int parts, int64_t q, int k,
                              float* out_scores, double* out_scores_f64, int64_t* out_idx, void* stream) {
    return dlc_topk_merge_strided(ctx, scores_f64, q * k, idx, q * k, parts, q, k, nullptr, 0.0, nullptr, out_scores,
                                  out_scores_f64, out_idx, nullptr, stream);
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
    // one-pass register form: rows of <= 64 threads x 4 (a wave per row) or <= 256 threads x 16 vectors of 16 bytes
    // whose vectors can be loaded / stored aligned; anything else takes the multi-pass kernel
    const int vw = src_dtype == DLC_F32 ? 4 : 2;
    const bool aligned = ((uintptr_t)src & 15) == 0 && (lds % vw) == 0 && ((uintptr_t)dst & 7) == 0;
    // (a wave per row up to 4 vectors per lane: with 16 -- rows of 4096 floats -- a lane's 64 fp64 accumulations and 64 fp64
    // scalings made a batch of 32 rows an 8 us launch; the choice depends on the row length alone, so that a row's stored
    // bits do not depend on how many rows are normalised with it)
    const int64_t cap_wave = 64 * 4 * vw, cap_block = 256 * 16 * vw;
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

// ---- operands of any norm: the certificate's tau follows the data --------------------------------------------------------
// tau of the plans above is stated for |q| |x| <= 1.01 (rows dlc_l2_normalize_rows wrote).  The score pass's error is
// linear in the product of the norms (every MFMA / v_dot2 step errs by at most eps * (|accumulator| + sum |products|) and
// partial sums are bounded by |q| |x|), so for other operands query qi certifies with tau * tau_scale[qi],
// tau_scale[qi] = max(1, |q_qi| * R / 1.01), R >= every database row's norm (of EVERY shard, when sharded).
namespace {
// fp32 norm of a stored row, rounded UP (one wave; the sum of squares of a lane's chain + a butterfly errs by < 1e-4
// relative for rows up to 2^20 elements; 1.001 covers it and the square root); non-finite -> +inf
template <typename Tag>
__device__ __forceinline__ float row_norm_up(const char* __restrict__ row, int d, int lane) {
    float acc = 0.f;
    for (int c = lane * 8; c < d; c += 512) {
        const u32x4_t v = *(const u32x4_t*)(row + (long long)c * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float lo, hi;
            if constexpr (__is_same(Tag, dlc_bf16_tag)) {
                lo = __uint_as_float(v[j] << 16);
                hi = __uint_as_float(v[j] & 0xffff0000u);
            } else {
                lo = dlc_f16_bits_to_f32((unsigned short)(v[j] & 0xffffu));
                hi = dlc_f16_bits_to_f32((unsigned short)(v[j] >> 16));
            }
            acc = fmaf(lo, lo, acc);
            acc = fmaf(hi, hi, acc);
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    const float nrm = sqrtf(acc) * 1.001f;
    return (nrm <= 3.0e38f) ? nrm : INFINITY;
}

template <typename Tag>
__global__ __launch_bounds__(256) void max_row_norm_kernel(const char* __restrict__ rows, long long ld_b, long long n, int d,
                                                           float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    float best = 0.f;
    for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += (long long)gridDim.x * 4)
        best = fmaxf(best, row_norm_up<Tag>(rows + r * ld_b, d, lane));
    // non-negative floats (and +inf) order like their bit patterns
    if (lane == 0) atomicMax((unsigned*)out, __float_as_uint(best));
}

template <typename Tag>
__global__ __launch_bounds__(256) void tau_scale_kernel(const char* __restrict__ Q, long long ldq_b, long long q, int d,
                                                        const float* __restrict__ db_max_norm, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long qi = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= q) return;
    const float nq = row_norm_up<Tag>(Q + qi * ldq_b, d, lane);
    const float R = db_max_norm ? *db_max_norm : 1.005f;
    // 0 * inf (an all-zero query against a poisoned database) must not come out as "certified with tau": NaN -> +inf
    const float prod = nq * R * (1.0f / 1.01f) * 1.000001f;
    const float s = (prod <= 3.0e38f) ? fmaxf(prod, 1.0f) : INFINITY;
    if (lane == 0) out[qi] = s;
}

int check_rows(dlc_ctx* ctx, const char* what, int dtype, const void* rows, int64_t n, int64_t ld, int64_t d) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (dtype != DLC_BF16 && dtype != DLC_F16)
        return dlc::fail(ctx, DLC_ERR_UNSUPPORTED, "%s: dtype %d (need DLC_BF16 or DLC_F16)", what, dtype);
    if (!rows || n < 1 || d < 1) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "%s: null/empty operand", what);
    if (d % 8 != 0 || ld < d || (ld % 8) || ((uintptr_t)rows & 15))
        return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "%s: rows must be 16-byte aligned, d and the stride multiples of 8 elements", what);
    if (d > 0x7ffffff0ll) return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "%s: d too large", what);
    return DLC_OK;
}
}  // namespace

extern "C" int dlc_max_row_norm(dlc_ctx* ctx, int dtype, const void* rows, int64_t n, int64_t ld, int64_t d,
                                float* max_norm, void* stream) {
    int rc = check_rows(ctx, "max_row_norm", dtype, rows, n, ld, d);
    if (rc != DLC_OK) return rc;
    if (!max_norm) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "max_row_norm: null output");
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    const unsigned grid = (unsigned)std::min<int64_t>(dlc::cdiv(n, 4), 256 * 16);
    if (dtype == DLC_BF16)
        hipLaunchKernelGGL(max_row_norm_kernel<dlc_bf16_tag>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const char*)rows,
                           (long long)ld * 2, (long long)n, (int)d, max_norm);
    else
        hipLaunchKernelGGL(max_row_norm_kernel<dlc_f16_tag>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const char*)rows,
                           (long long)ld * 2, (long long)n, (int)d, max_norm);
    DLC_LAUNCH_CHECK(ctx, "max_row_norm_kernel");
    return DLC_OK;
}

extern "C" int dlc_cosine_tau_scale(dlc_ctx* ctx, int dtype, const void* Q, int64_t q, int64_t ldq, int64_t d,
                                    const float* db_max_norm, float* tau_scale, void* stream) {
    int rc = check_rows(ctx, "cosine_tau_scale", dtype, Q, q, ldq, d);
    if (rc != DLC_OK) return rc;
    if (!tau_scale) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "cosine_tau_scale: null output");
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    const unsigned grid = (unsigned)dlc::cdiv(q, 4);
    if (dtype == DLC_BF16)
        hipLaunchKernelGGL(tau_scale_kernel<dlc_bf16_tag>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const char*)Q,
                           (long long)ldq * 2, (long long)q, (int)d, db_max_norm, tau_scale);
    else
        hipLaunchKernelGGL(tau_scale_kernel<dlc_f16_tag>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const char*)Q,
                           (long long)ldq * 2, (long long)q, (int)d, db_max_norm, tau_scale);
    DLC_LAUNCH_CHECK(ctx, "tau_scale_kernel");
    return DLC_OK;
}

extern "C" double dlc_cosine_score_error_bound_any_plan(int64_t d) {
    if (d < BK) return 0.0;
    const double nk = (double)(d / BK);
    // the unsplit MFMA pass (2 k-slices per K tile of 64) or the bandwidth kernel's lane chain: split-K chunks are shorter
    const double steps = std::max(2.0 * nk + 2.0, nk * 0.5 + 8.0);
    return SCORE_STEP_EPS * 1.01 * steps + 3.7e-12;
}
