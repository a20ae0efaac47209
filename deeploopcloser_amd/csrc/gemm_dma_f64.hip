// fp64 GEMM with fused bias + activation whose operands reach LDS by DMA (global_load_lds_dwordx4) through a
// 3-deep ring, for the large launches of the SDAV layers (SDAV.py:129-157), the CnnVtl convolutions as implicit
// GEMMs (cnn_vtl.py:47-93) and the Gram blocks of the SDAV similarity (SimilarityCalculator.py:30-37).
//
// Why a second kernel.  The register-staged kernel of gemm_dense.hip keeps the fp64 matrix pipe 72-80 % busy
// (profiles/r02a_gemm_f64_pmc.json); timing builds without its operand staging run the same MFMAs + LDS reads
// 13 % faster, without its barrier another 6 %.  What costs is not LDS or HBM bandwidth but (a)
// 16 global loads, 16 LDS stores and their address arithmetic per thread and K tile in the waves' instruction
// streams, and (b) one K tile of prefetch distance, so that every workgroup's barrier waits for the slowest of its
// 2048 loads.  Here a K tile of both operands is 6 DMA instructions per wave: no registers, no LDS stores.
//
// Tile 256 x 128 x 16, 512 threads = 8 waves (4 x 2), each wave 64 x 64 = 4 x 4 v_mfma_f64_16x16x4_f64 tiles.
// (A 128 x 128 form of the same kernel -- MI = 2: each wave 32 x 64, stages of 32 KiB -- takes launches of at most half a
// round of 256-row tiles.)  One workgroup per CU (two waves per SIMD), LDS ring of 3 stages x 48 KiB:
//   A stage  256 rows x 128 B (16 doubles of K per row), 16-byte piece q of row r at slot q ^ ((r >> 1) & 7)
//   B stage  [N,K] operand: 128 rows x 128 B, same swizzle;  [K,N] operand: 16 k-rows x 1 KiB, the tile's eight
//            16-column groups in kn_unit() order, neighbouring 128-byte units swapped in odd k-rows
// so that each half-wave of a ds_read_b64 fragment read (16 rows x one 16-byte slot: even rows sit in banks 0-31,
// odd rows in 32-63, and 8 rows of one parity take 8 different slots) touches every bank once.  (Keyed on r & 7
// instead, rows r and r + 8 shared their banks: SQ_LDS_BANK_CONFLICT was half of SQ_LDS_IDX_ACTIVE.)  The swizzle is applied to the DMA's per-lane SOURCE address; the LDS destination stays lane-linear.
// Rows past M / columns past N are clamped to the last valid one (their results are never stored); a K tail reads a
// page of zeros instead, convolution padding is served as zeros by out-of-range buffer offsets (dma_a4b).
// Addresses cost the vector unit nothing per K tile: DMA sources are a scalar base + fixed 32-bit lane offsets, LDS
// read pointers step once per tile -- every vector / LDS instruction beside an fp64 MFMA costs the SIMD ~7 cycles.
//
// One barrier per K tile t: before it every wave waits for its own DMA pieces of tile t, behind it tile t is
// readable and tile t+1's DMA is issued into the stage of tile t-2.  Waves 0-3 then run k-slices 0-3 of tile t;
// their SIMD partners 4-7 run half a tile behind (k-slices 2-3 of tile t-1, then 0-1 of tile t) -- see the main loop.
#include <algorithm>
#include "gemm_internal.h"

namespace dlc_gemm {
namespace {

constexpr int TM3 = 256, TN3 = 128, TK3 = 16, NT3 = 512;
constexpr int B_STAGE = TN3 * TK3 * 8;          // 16 KiB
constexpr int NSTAGE = 3;

struct DmaArgs {
    const char* A; long long lda_b;             // plain: row stride in BYTES; conv: unused
    const char* B; long long ldb_b;
    const double* bias;
    double* C; long long ldc;
    long long M, N, K;                          // K: the reduction length as the A operand has it (even)
    long long Kb;                               // ... and as B has it (<= K): B's k-rows Kb .. K-1 do not exist and read as zeros
    int act;                                    // DLC_ACT_* of include/dlc.h, or ACT_AXPY: C += alpha * (A . B) (no bias)
    double alpha;
    ConvGeom cv;
    long long m_base;                           // CONV: output pixel index of row 0 (a launch over the tail rows of a convolution)
    int cv_all_valid;                           // no tap of any output pixel falls outside the input (VALID, no padding)
    const char* zero;                           // >= 128 bytes of zeros
    long long kchunk;                           // > 0: split-K -- workgroup id = tile * nchunks + c sums k in [c * kchunk, (c + 1) * kchunk)
    double* P;                                  // ... into its chunk's partial tile P[c][M][N] (no bias, no activation)
    int nchunks;
    long long tiles_m, tiles_n, nbr, nblocks;
    int br, bc, lg_blk;                         // blocks of br x bc = 1 << lg_blk tiles
    int tri_p;
    long long tri_row0, tri_col0;
    // triangular launches enumerate only the blocks that hold wanted entries, column by column (see block_of)
    long long tri_nbc;                          // block columns
    int tri_blk_cols, tri_rem0, tri_step, tri_lg_rows;
};

// Triangular launches (the Gram blocks of the similarity: only entries whose row frame < column frame are read).
// Workgroup ids go round-robin to the 8 XCDs, so XCD x runs blocks x, x + 8, ... of whatever order the blocks are
// numbered in.  Numbered row-major over the full rectangle, with a row count that is a multiple of 8 (32 at 1063
// frames), XCD x owns block ROWS x, x + 8, ..: XCD 0 got 80 blocks' worth of the triangle and XCD 7 got 52 -- the
// launch took 49.4 ms where half of the full product's 80.5 ms is 40.7 (scripts/exp_gram_shapes.py).  So: number only
// the blocks that hold at least one wanted entry, column by column; every XCD then gets the same count +- 1.
// In block column c the wanted block rows are 0 .. cnt(c) - 1:  block (r, c) is wanted iff its last column's frame
// is past its first row's frame, (col0 + (c + 1) * BC - 1) / p > (row0 + r * BR) / p, i.e. r * BR < F * p - row0 with
// F * p = the column's last index rounded down to a multiple of p.  BR is a power of two; the remainder is stepped.
struct TriWalk {
    long long t;                                // last column index (col0 based) of the current block column
    int rem;                                    // t % p
    __host__ __device__ long long count(const DmaArgs& a) const {
        const long long thr = t - rem - a.tri_row0;
        if (thr <= 0) return 0;
        const long long c = (thr + (1ll << a.tri_lg_rows) - 1) >> a.tri_lg_rows;
        return c < a.nbr ? c : a.nbr;
    }
    __host__ __device__ void step(const DmaArgs& a) {
        t += a.tri_blk_cols;
        rem += a.tri_step;
        if (rem >= a.tri_p) rem -= a.tri_p;
    }
};

typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) const char* lcptr_t;

// Between the k-slices of a K tile: for the plain operands, a scheduling fence -- the source's order (reads of slice
// k+1, then the 16 MFMAs of slice k) is the schedule; left alone hipcc pairs B reads of neighbouring slices and moves
// them up, which measured 1.8 % (SDAV layers) and 2.7 % (Gram) slower.  The convolution form, whose DMA addresses are
// VALU work the compiler spreads between the MFMAs, is 1 % faster without the fence (re-measured after that VALU work
// had moved to the scalar unit: still 1.2 %).
// Measured and not kept (scripts/exp_dgemm.py, one device): s_setprio by progress through the tile (the wave that is
// behind gets the matrix pipe), by half tile, or static for the late waves: each 2 % SLOWER than no priorities -- the
// 1.6 : 1 split of a K tile's cycles between the two waves of a SIMD (cycle stamps) is not what costs
// the time; volatile fragment reads (to keep hipcc from pairing A reads into ds_read2st64_b64, which has half the
// rate of ds_read_b64 and sees 32 banks): 37 % slower, every read followed by a full wait.
#define DLC_SLICE_FENCE() do { if constexpr (!CONV) __builtin_amdgcn_sched_barrier(0); } while (0)


// LDS-DMA wave-instructions of one K tile: four 1 KiB pieces of the A stage (dma_a4), two of the B stage (dma_b2).
// Inline asm so that hipcc does not count them in vmcnt (it would wait for vmcnt(0) in front of every LDS read); M0
// carries the wave-uniform LDS destination and is saved / restored because the compiler owns it.  s_nop 4 covers
// SGPR operands freshly written by v_readfirstlane.
__device__ __forceinline__ void dma_a4(const char* a0, const char* a1, const char* a2, const char* a3, unsigned lds_a) {
    unsigned keep;
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %5\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_add_u32 m0, %5, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, off\n\t"
        "s_add_u32 m0, %5, 0x800\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, off\n\t"
        "s_add_u32 m0, %5, 0xc00\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %4, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "s"(lds_a)
        : "memory", "scc");
}
__device__ __forceinline__ void dma_b2(const char* b0, const char* b1, unsigned lds_b) {
    unsigned keep;
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_add_u32 m0, %3, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(b0), "v"(b1), "s"(lds_b)
        : "memory", "scc");
}

// The same with a wave-uniform 64-bit base in SGPRs and 32-bit per-lane offsets that never change: a K tile's
// addresses then cost scalar adds only (each vector instruction beside the fp64 MFMAs costs the SIMD ~7 cycles:
// 20 more of them per K tile measured 1.4 % -- DLC_EXP_DMA_SPLIT_A_READS).
__device__ __forceinline__ void dma_a4s(unsigned o0, unsigned o1, unsigned o2, unsigned o3, const char* base, unsigned lds_a) {
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
        : "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(base), "s"(lds_a)
        : "memory", "scc");
}
// The convolution's A operand: buffer addressing -- a wave-uniform descriptor (base stepping with the kernel tap and
// channel block) + per-lane 32-bit offsets; a lane whose tap falls into the padding carries an offset past
// num_records and the hardware writes ZEROS to its LDS slot (scripts/micro/buffer_lds_oob.hip: out-of-range dwords of
// a `buffer_load ... lds` land as 0; an soffset counts in the range check, so the base is stepped instead).
// "s" operands must BE in SGPRs: hipcc does not move a value it keeps in VGPRs there by itself (a diagnostic build
// failed to assemble that way), so the wave-uniform bases go through readfirstlane -- a no-op on a value that is
// already scalar.
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}
typedef __attribute__((ext_vector_type(4))) unsigned rsrc_t;
constexpr unsigned DMA_OOB = 0xfffffff0u;       // >= num_records of every descriptor built here
constexpr unsigned DMA_NUM_RECORDS = 0x80000000u;
__device__ __forceinline__ rsrc_t make_rsrc(const char* base) {
    const unsigned long long a = (unsigned long long)base;
    rsrc_t r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);       // stride 0: raw buffer
    r[2] = DMA_NUM_RECORDS;
    r[3] = 0x00020000u;                         // gfx9 raw-buffer word (32-bit data format)
    return r;
}
__device__ __forceinline__ void dma_a4b(unsigned o0, unsigned o1, unsigned o2, unsigned o3, rsrc_t rsrc, unsigned lds_a) {
    unsigned keep;
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %6\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %5, 0 offen lds\n\t"
        "s_add_u32 m0, %6, 0x400\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %2, %5, 0 offen lds\n\t"
        "s_add_u32 m0, %6, 0x800\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %3, %5, 0 offen lds\n\t"
        "s_add_u32 m0, %6, 0xc00\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %4, %5, 0 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(rsrc), "s"(lds_a)
        : "memory", "scc");
}
// ... and of one (the 64-row tile's A part: 8 rows per wave)
__device__ __forceinline__ void dma_1(const char* a0, unsigned lds_a) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(a0), "s"(lds_a) : "memory", "scc");
}
__device__ __forceinline__ void dma_1s(unsigned o0, const char* base, unsigned lds_a) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(o0), "s"(base), "s"(lds_a) : "memory", "scc");
}
__device__ __forceinline__ void dma_1b(unsigned o0, rsrc_t rsrc, unsigned lds_a) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(o0), "s"(rsrc), "s"(lds_a) : "memory", "scc");
}
__device__ __forceinline__ void dma_a2b(unsigned o0, unsigned o1, rsrc_t rsrc, unsigned lds_a) {
    unsigned keep;
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %4\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %3, 0 offen lds\n\t"
        "s_add_u32 m0, %4, 0x400\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %2, %3, 0 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(o0), "v"(o1), "s"(rsrc), "s"(lds_a)
        : "memory", "scc");
}
__device__ __forceinline__ void dma_b2s(unsigned o0, unsigned o1, const char* base, unsigned lds_b) {
    unsigned keep;
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %4\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %3\n\t"
        "s_add_u32 m0, %4, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %3\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(o0), "v"(o1), "s"(base), "s"(lds_b)
        : "memory", "scc");
}

constexpr int ACT_AXPY = 16;                      // internal: the product is added into C (the SGD step of a weight gradient)
__device__ __forceinline__ double act_f64(double z, int act) {
    if (act == DLC_ACT_SIGMOID) return 1.0 / (1.0 + exp(-z));
    if (act == DLC_ACT_RELU) return z > 0.0 ? z : 0.0;
    return z;
}

// [K,N] B stage: where the tile's 16-column groups sit in a 1 KiB k-row.  Group (wc, j) -- wave column wc, MFMA
// column tile j -- is unit ((j >> 1) << 2) | (wc << 1) | (j & 1), so that a wave's groups j and j + 2 are 512 bytes
// apart: one ds_read2st64_b64 fetches both (the fragment reads of a k-slice are then 2 + 2 instructions, not 2 + 4).
__device__ __forceinline__ int kn_unit(int wc, int j) { return ((j >> 1) << 2) | (wc << 1) | (j & 1); }
template <int NJ>
__device__ __forceinline__ int kn_group(int unit) {       // inverse: the column group (0 .. 2 * NJ - 1) a unit holds
    const int j = ((unit >> 2) << 1) | (unit & 1), wc = (unit >> 1) & 1;
    const int g = wc * NJ + j;
    return g < 2 * NJ ? (j < NJ ? g : 2 * NJ - 1) : 2 * NJ - 1;  // NJ = 3: units of j = 3 are never read
}

// BLAYOUT: DLC_B_KN / DLC_B_NK.  CONV: A is the NHWC input of a convolution with C % 16 == 0 (a K tile is 16
// consecutive channels of one kernel tap), B its HWIO kernel as [K,N].
// NJ: MFMA column tiles per wave (4: the 128-column tile; 3: a 96-column tile for N <= 96 such as conv1's 96 filters,
// which would waste a quarter of a 128-column tile's MFMAs).  The B stage keeps its 128-column geometry.
// MI: MFMA row tiles per wave -- 4: the 256-row tile; 2: a 128-row tile (8 waves of 32 x 64) for launches of at most
// half a round of 256-row tiles, which then spread over twice the CUs at half the work each.  Same k order, same bits.
template <int BLAYOUT, bool CONV, int NJ, int MI>
__global__ __launch_bounds__(NT3, 2) void gemm_dma_f64_kernel(DmaArgs p) {
    constexpr int TM = 64 * MI;                   // rows per tile: four wave rows of MI MFMA tiles
    constexpr int A_STAGE = TM * TK3 * 8;         // 32 / 16 KiB (shadows the namespace constants of the 256-row tile)
    constexpr int STAGE = A_STAGE + B_STAGE;
    constexpr int TNJ = 2 * NJ * 16;              // columns per tile: two wave columns of NJ MFMA tiles
    extern __shared__ __attribute__((aligned(16))) char smem3[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 1, wc = w & 1;
    // ---- workgroup id -> tile: blocks of br x bc = 32 tiles, one block per XCD at a time (gemm_dense.hip, tile order)
    long long tile_m, tile_n;
    int chunk = 0;
    if (!CONV && p.kchunk > 0) {
        // split-K launches are small (fewer tiles than CUs): no blocks, no holes -- the id counts (tile, chunk) pairs, chunk
        // fastest, so that consecutive ids (dealt round-robin to XCDs and shader engines) are all real work.  (With the
        // chunk in grid.y every slice repeated the block order's holes: 500 workgroups of 1280 ran on the SAME hundred
        // CUs' worth of positions, and five chunks took as long as one pass.)
        const long long wg = blockIdx.x;
        chunk = (int)(wg % p.nchunks);
        const long long t = wg / p.nchunks;
        tile_m = t % p.tiles_m;
        tile_n = t / p.tiles_m;
        if (tile_n >= p.tiles_n) return;
    } else {
        const long long wg = blockIdx.x;
        const long long l = wg >> 3;
        long long gb = (l >> p.lg_blk) * 8 + (wg & 7);
        if (gb >= p.nblocks) return;
        const int i = (int)(l & ((1 << p.lg_blk) - 1));
        long long brow, bcol;
        if (p.tri_p > 0) {                                               // the gb-th wanted block, columns first
            TriWalk tw{p.tri_col0 + p.tri_blk_cols - 1, p.tri_rem0};
            bcol = 0;
            for (;;) {
                const long long cnt = tw.count(p);
                if (gb < cnt) break;
                gb -= cnt;
                tw.step(p);
                if (++bcol >= p.tri_nbc) return;                          // cannot happen: nblocks is the sum of the counts
            }
            brow = gb;
        } else {
            brow = gb % p.nbr;
            bcol = gb / p.nbr;
        }
        // Inside an XCD the dispatcher deals workgroups round-robin to 4 shader engines of 8 CUs, and a workgroup waits
        // for ITS engine even while others have free CUs (scripts/exp_placement.py).  The tiles of a block that exist
        // must therefore be spread over i % 4: columns first in general (a right-edge block keeps i < br * cols), rows
        // first in the last block row when it is cut short (it keeps i < bc * rows) -- numbered columns first, the three
        // tiles of a 1-row block sat at i = 0, 8, 16, all on one engine: conv3 of 128 frames ran two rounds for 195 tiles.
        if (brow == p.nbr - 1 && (p.tiles_m & (p.br - 1)) != 0) {
            tile_m = brow * p.br + (i / p.bc);
            tile_n = bcol * p.bc + (i % p.bc);
        } else {
            tile_m = brow * p.br + (i % p.br);
            tile_n = bcol * p.bc + (i / p.br);
        }
        if (tile_m >= p.tiles_m || tile_n >= p.tiles_n) return;
    }
    const long long m0 = tile_m * TM, n0 = tile_n * TNJ;
    if (p.tri_p > 0 && (p.tri_col0 + n0 + TNJ - 1) / p.tri_p <= (p.tri_row0 + m0) / p.tri_p) return;
    if constexpr (!CONV) {
        // split-K (few tiles, long K: the training step's 300-row products, an encode of a few frames): this workgroup's
        // K range is chunk blockIdx.y; from here on the kernel is the one-pass kernel on the shifted operands, writing the
        // bare sums to the chunk's partial tile.  kchunk is a multiple of the K tile, so only the last chunk has a tail.
        if (p.kchunk > 0) {
            const long long k0 = (long long)chunk * p.kchunk;
            p.A += k0 * 8;
            p.B += BLAYOUT == DLC_B_NK ? k0 * 8 : k0 * p.ldb_b;
            const long long kc = p.K - k0 < p.kchunk ? p.K - k0 : p.kchunk;
            long long kb = p.Kb - k0;
            kb = kb < 0 ? 0 : (kb > kc ? kc : kb);
            p.K = kc; p.Kb = kb;
            p.C = p.P + (long long)chunk * p.M * p.N;
            p.ldc = p.N; p.act = DLC_ACT_NONE; p.bias = nullptr;
        }
    }
    const int nkt = (int)((p.K + TK3 - 1) / TK3);
    const unsigned lds_base = (unsigned)(unsigned long long)(lptr_t)smem3;

    // ---- DMA sources.  A (and an [N,K] B): instruction j of this wave covers rows 8 * (4w + j) .. + 7 of the stage,
    // lane -> (row = lane >> 3, slot = lane & 7), source piece = slot ^ ((row >> 1) & 7).
    const int slot = lane & 7;
    // Plain operands: a wave-uniform 64-bit base that steps from K tile to K tile (scalar adds) + per-lane 32-bit byte
    // offsets that never change; 64-bit per-lane addresses are formed for a K-tail tile only (at most the last one).
    unsigned a_off[MI];
    int a_piece[MI];                               // source piece (k offset 2 * piece doubles inside the K tile)
    // CONV: a row is an output pixel.  a_off = byte offset of its tap (0, 0), channel 2 * piece, from the image of the
    // tile's first row shifted up-left by the padding (so that it is never negative); the tap and the channel block
    // move the descriptor's base, the same for every row.  iy0 / ix0: the pixel's input position at tap (0, 0).
    int cv_iy0[MI], cv_ix0[MI];
    unsigned a_eff[MI];                            // a_off, or DMA_OOB while the current tap is padding for this row
    long long cv_img0 = 0;
    if constexpr (CONV) cv_img0 = (p.m_base + m0) / ((long long)p.cv.OH * p.cv.OW);
#pragma unroll
    for (int j = 0; j < MI; ++j) {
        const int r = (w * MI + j) * 8 + (lane >> 3);
        a_piece[j] = slot ^ ((r >> 1) & 7);
        long long gm = m0 + r;
        if (gm > p.M - 1) gm = p.M - 1;
        if constexpr (CONV) {
            const long long px = p.m_base + gm;                          // output pixel index over all images
            const long long img = px / ((long long)p.cv.OH * p.cv.OW);
            const int rem = (int)(px - img * p.cv.OH * p.cv.OW);
            const int oy = rem / p.cv.OW, ox = rem - oy * p.cv.OW;
            cv_iy0[j] = oy * p.cv.stride - p.cv.pad_t;
            cv_ix0[j] = ox * p.cv.stride - p.cv.pad_l;
            a_off[j] = (unsigned)((((img - cv_img0) * p.cv.H + oy * p.cv.stride) * p.cv.W + ox * p.cv.stride) * p.cv.C + a_piece[j] * 2) * 8u;
            a_eff[j] = a_off[j];
        } else {
            a_off[j] = (unsigned)((gm - m0) * p.lda_b) + a_piece[j] * 16;     // < 2^32: checked by the launcher
        }
    }
    // CONV: element offset of tap (0, 0), channel 0 of the tile's first image, shifted by the padding
    const long long cv_e0 = CONV ? (cv_img0 * p.cv.H - p.cv.pad_t) * (long long)p.cv.W * p.cv.C - (long long)p.cv.pad_l * p.cv.C : 0;
    const char* a_base = p.A + m0 * p.lda_b;      // (CONV: unused)
    unsigned b_off[2];
    int b_piece[2];
    const char* b_base;
    long long b_step;                             // bytes from one K tile to the next
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if constexpr (BLAYOUT == DLC_B_NK) {
            const int r = (w * 2 + j) * 8 + (lane >> 3);
            b_piece[j] = slot ^ ((r >> 1) & 7);
            long long gn = n0 + r;
            if (gn > p.N - 1) gn = p.N - 1;
            b_off[j] = (unsigned)((gn - n0) * p.ldb_b) + b_piece[j] * 16;
        } else {
            // [K,N]: instruction j covers k-row kr = 2w + j of the stage; lane -> 16-byte slot of the 1 KiB row.  The
            // row holds the tile's eight 16-column groups (128 B each) in the order kn_unit() gives, odd k-rows with
            // neighbouring units swapped (the bank swizzle): unit u of the row comes from column group kn_group(u).
            const int kr = w * 2 + j;
            int piece = kn_group<NJ>((lane >> 3) ^ (kr & 1)) * 8 + (lane & 7);
            const long long cols = p.N - n0 < TNJ ? p.N - n0 : TNJ;      // valid columns of this tile (even: N is)
            const int last = (int)(cols / 2) - 1;
            if (piece > last) piece = last;                              // columns past N: never stored
            b_piece[j] = kr;
            b_off[j] = (unsigned)(kr * p.ldb_b) + piece * 16;
        }
    }
    if constexpr (BLAYOUT == DLC_B_NK) { b_base = p.B + n0 * p.ldb_b; b_step = TK3 * 8; }
    else { b_base = p.B + n0 * 8; b_step = TK3 * p.ldb_b; }
    const char* zsrc = p.zero + slot * 16;
    const bool a_tail = (p.K & (TK3 - 1)) != 0;                          // the last tile's k past K reads zeros
    const bool b_tail = a_tail || p.Kb != p.K;                           // ... and B's k past Kb (Kb >= the last tile's first k)
    // wave-uniform kernel tap of the NEXT tile to issue (CONV): k0 = (ky * KW + kx) * C + c0
    int cv_c0 = 0, cv_kx = 0, cv_ky = 0;

    // The A part (4 instructions) and the B part (2) of K tile t's DMA.
    auto issue_a = [&](int t, int stage) {
        const int tt = t;
        const unsigned lds_a = lds_base + stage * STAGE + w * (MI * 1024);
        if constexpr (CONV) {
            // (Keeping the rows' validity as four SGPR lane masks and selecting the offsets with one v_cndmask per row and
            // tile removes the 8-12 register copies per tile this conditional update costs -- and measured 4 % slower.)
            if (cv_c0 == 0 && !p.cv_all_valid) {                         // a new tap: which rows does it send into the padding?
#pragma unroll
                for (int j = 0; j < MI; ++j) {
                    const bool ok = (unsigned)(cv_iy0[j] + cv_ky) < (unsigned)p.cv.H && (unsigned)(cv_ix0[j] + cv_kx) < (unsigned)p.cv.W;
                    a_eff[j] = ok ? a_off[j] : DMA_OOB;
                }
            }
            const long long e = cv_e0 + ((long long)cv_ky * p.cv.W + cv_kx) * p.cv.C + cv_c0;
            const rsrc_t rs = make_rsrc(p.A + e * 8);
            if (t < nkt - 1) {                                           // step the tap; frozen once the last tile is reached
                cv_c0 += TK3;
                if (cv_c0 >= p.cv.C) {
                    cv_c0 = 0;
                    if (++cv_kx == p.cv.KW) { cv_kx = 0; ++cv_ky; }
                }
            }
            if constexpr (MI == 4) dma_a4b(a_eff[0], a_eff[1], a_eff[2], a_eff[3], rs, lds_a);
            else if constexpr (MI == 2) dma_a2b(a_eff[0], a_eff[1], rs, lds_a);
            else dma_1b(a_eff[0], rs, lds_a);
        } else {
            const char* base = uniform_ptr(a_base + (long long)tt * (TK3 * 8));
            if (a_tail && tt == nkt - 1) {
                const int klim = (int)(p.K - (long long)tt * TK3);       // valid k of this tile
                const char* sa[MI];
#pragma unroll
                for (int j = 0; j < MI; ++j) sa[j] = a_piece[j] * 2 >= klim ? zsrc : base + a_off[j];
                if constexpr (MI == 4) dma_a4(sa[0], sa[1], sa[2], sa[3], lds_a);
                else if constexpr (MI == 2) dma_b2(sa[0], sa[1], lds_a);
                else dma_1(sa[0], lds_a);
            } else {
                if constexpr (MI == 4) dma_a4s(a_off[0], a_off[1], a_off[2], a_off[3], base, lds_a);
                else if constexpr (MI == 2) dma_b2s(a_off[0], a_off[1], base, lds_a);
                else dma_1s(a_off[0], base, lds_a);
            }
        }
    };
    auto issue_b = [&](int t, int stage) {
        const int tt = t;
        const unsigned lds_b = lds_base + stage * STAGE + A_STAGE + w * 2048;
        const char* base = uniform_ptr(b_base + (long long)tt * b_step);
        if (b_tail && tt == nkt - 1) {
            const long long left = p.Kb - (long long)tt * TK3;           // B's own reduction length
            const int klim = (int)(left < TK3 ? (left > 0 ? left : 0) : TK3);
            const char* sb[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if constexpr (BLAYOUT == DLC_B_NK) sb[j] = b_piece[j] * 2 >= klim ? zsrc : base + b_off[j];
                else sb[j] = b_piece[j] >= klim ? zsrc : base + b_off[j];
            }
            dma_b2(sb[0], sb[1], lds_b);
        } else {
            dma_b2s(b_off[0], b_off[1], base, lds_b);
        }
    };

    // ---- fragment read addresses: byte offsets into the ring, for the stage being read; they step with the ring once
    // per K tile (8 vector adds) instead of being rebuilt from stage + lane offsets in front of every k-slice's reads.
    // (Unrolling the K loop by the ring's three stages, so that a stage's offset becomes an immediate of the reads and
    // the 6 adds go away, measured no faster -- SDAV 27.78 vs 27.72 ms, CnnVtl 30.2 vs 29.7 -- at five times the loop code.)
    const int fr = lane & 15, fk = lane >> 4;
    const int x7 = (fr >> 1) & 7;           // the rows' swizzle key: stage row = 16 * something + fr
    lcptr_t fa_addr[4];                     // A, k-slice kk: row wr * 64 + fr; MFMA row tile i adds i * 2048 (an immediate)
    lcptr_t fb_addr[4];                     // [N,K] B, k-slice kk: row wc * NJ * 16 + fr (column tile j adds j * 2048);
                                            // [K,N] B, column tile j: k-row fk (k-slice kk adds kk * 4096)
    const lcptr_t ring = (lcptr_t)smem3;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int pk = (((kk * 2 + (fk >> 1)) ^ x7) << 4) + (fk & 1) * 8;
        fa_addr[kk] = ring + (wr * (16 * MI) + fr) * 128 + pk;
        if constexpr (BLAYOUT == DLC_B_NK) fb_addr[kk] = ring + A_STAGE + (wc * NJ * 16 + fr) * 128 + pk;
        else fb_addr[kk] = ring + A_STAGE + fk * 1024 + ((kn_unit(wc, kk & 1) ^ (fk & 1)) << 7) + fr * 8;       // kk = j: 0, 1
    }
    auto advance = [&](int stage_now) {                                  // the addresses move on to the next stage of the ring
        const int d = stage_now == NSTAGE - 1 ? -(NSTAGE - 1) * STAGE : STAGE;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            fa_addr[x] += d;
            asm volatile("" : "+v"(fa_addr[x]));
            if (BLAYOUT == DLC_B_NK || x < 2) {
                fb_addr[x] += d;
                asm volatile("" : "+v"(fb_addr[x]));
            }
        }
    };

    f64x4_t acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f64x4_t){0, 0, 0, 0};

    // fragments of one k-slice, double-buffered in registers: rd() reads slice kk of a stage into buffer b, mm() runs
    // the 16 MFMAs of a buffer.  A slice's reads are always issued before the MFMAs of the slice in front of it.
    double fa[2][MI], fb[2][NJ];
    auto rd = [&](int kk, int b) {
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[b][i] = *(const __attribute__((address_space(3))) double*)(fa_addr[kk] + i * 2048);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if constexpr (BLAYOUT == DLC_B_NK) fb[b][j] = *(const __attribute__((address_space(3))) double*)(fb_addr[kk] + j * 2048);
            else fb[b][j] = *(const __attribute__((address_space(3))) double*)(fb_addr[j & 1] + (j >> 1) * 512 + kk * 4096);
        }
    };
    auto mm = [&](int b) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[b][i], fb[b][j], acc[i][j], 0, 0, 0);
    };
    // wait for this wave's DMA pieces (all that are in flight belong to the tile about to become readable), then the
    // workgroup barrier: behind it that tile is visible to every wave and the stage of the tile two back is free
    auto arrive = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto issue = [&](int t, int stage) {
        issue_a(t, stage);
        issue_b(t, stage);
    };
    auto next_stage = [](int s_) { return s_ + 1 == NSTAGE ? 0 : s_ + 1; };

    // The two waves of a SIMD (w and w + 4) run HALF A K TILE APART.  Run in phase, all eight waves leave a barrier
    // together, issue their fragment reads together and wait for LDS together, with the matrix pipes empty; and they
    // do so again wherever the compiler batches reads.  Out of phase, the late wave of each SIMD crosses barrier t
    // with the fragments of k-slice 2 of tile t-1 already in registers and issues MFMAs at once, while the early
    // wave reads k-slice 0 of tile t; half a tile later the roles are swapped.  Tile t+1's DMA is issued behind
    // barrier t into the stage of tile t-2, whose last readers (the late waves, k-slices 2-3, during iteration t-1)
    // are through; it has a whole iteration to land.  Both groups execute nkt + 1 barriers.
    // (A second barrier in the middle of the iteration, behind which tile t+2 can go into tile t-1's stage -- a tile
    // and a half of prefetch distance -- measured the same for the SDAV layers and 3 % slower for the convolutions.
    // Timing builds that re-read ONE resident K tile with the same DMA instructions run as fast as builds without
    // any DMA: what the operand traffic costs, 8 %, is neither instruction issue nor prefetch distance.)
    issue(0, 0);
    int cur = 0;                                                         // the stage the read addresses point at
    if (w < 4) {
        for (int t = 0; t < nkt; ++t) {
            arrive();                                                    // barrier t
            const int nxt = next_stage(cur);
            rd(0, 0);
            if (t + 1 < nkt) issue(t + 1, nxt);
            rd(1, 1); DLC_SLICE_FENCE(); mm(0);
            rd(2, 0); DLC_SLICE_FENCE(); mm(1);
            rd(3, 1); DLC_SLICE_FENCE(); mm(0);
            advance(cur);
            DLC_SLICE_FENCE(); mm(1);
            cur = nxt;
        }
        arrive();                                                        // barrier nkt (the late waves' last half tile)
    } else {
        {                                                                // t = 0: the first half of tile 0
            arrive();
            rd(0, 0);
            if (1 < nkt) issue(1, 1);
            rd(1, 1); mm(0);
            rd(2, 0); mm(1);                                             // buffer 0 now holds k-slice 2 of tile 0
        }
        for (int t = 1; t < nkt; ++t) {
            arrive();                                                    // barrier t
            const int nxt = next_stage(cur);                             // tile t's stage; the addresses still point at tile t-1's
            rd(3, 1); DLC_SLICE_FENCE(); mm(0);
            if (t + 1 < nkt) issue(t + 1, next_stage(nxt));
            advance(cur);
            rd(0, 0); DLC_SLICE_FENCE(); mm(1);
            rd(1, 1); DLC_SLICE_FENCE(); mm(0);
            rd(2, 0); DLC_SLICE_FENCE(); mm(1);
            cur = nxt;
        }
        arrive();                                                        // barrier nkt
        rd(3, 1); mm(0);
        mm(1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // no DMA may outlive the workgroup's LDS

    // ---- epilogue: bias + activation, C/D layout of v_mfma_f64_16x16x4_f64: row = (lane >> 4) + 4 * reg, col = lane & 15
    // CONV with mm_keys: the minimum / maximum of every image's outputs is folded into ordered keys on the way out
    // (the CnnVtl descriptor's per-frame range, cnn_vtl.py:110-112: a separate pass over the five layers' outputs read
    // 4.9 GB again).  A wave's 64 rows (output pixels) touch at most two images -- the launcher checks OH * OW >= 64.
    if (!CONV && p.act == ACT_AXPY) {
        // C += alpha * acc: ALL of the lane's old values first, then the fmas and the stores (written as one expression per
        // element, every element was load -> wait -> fma -> store: sixteen memory round trips in a row)
        constexpr int JB = MI >= 4 ? 1 : (MI == 2 ? 2 : NJ);      // column groups per batch: 16 old values in registers at a time
#pragma unroll
        for (int j0 = 0; j0 < NJ; j0 += JB) {
            double old[MI][JB][4];
#pragma unroll
            for (int jj = 0; jj < JB; ++jj) {
                const long long gn = n0 + (wc * NJ + j0 + jj) * 16 + fr;
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const long long gm = m0 + wr * (16 * MI) + i * 16 + fk + 4 * r;
                        old[i][jj][r] = (j0 + jj < NJ && gn < p.N && gm < p.M) ? p.C[gm * p.ldc + gn] : 0.0;
                    }
            }
#pragma unroll
            for (int jj = 0; jj < JB; ++jj) {
                const long long gn = n0 + (wc * NJ + j0 + jj) * 16 + fr;
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const long long gm = m0 + wr * (16 * MI) + i * 16 + fk + 4 * r;
                        if (j0 + jj < NJ && gn < p.N && gm < p.M) p.C[gm * p.ldc + gn] = fma(p.alpha, acc[i][j0 + jj < NJ ? j0 + jj : 0][r], old[i][jj][r]);
                    }
            }
        }
        return;
    }
    double mn0 = INFINITY, mx0 = -INFINITY, mn1 = INFINITY, mx1 = -INFINITY;
    long long mm_img0 = 0, mm_bnd = 0;
    const bool fold = CONV && p.cv.mm_keys != nullptr;
    if (fold) {
        const long long per_img = (long long)p.cv.OH * p.cv.OW;
        mm_img0 = (p.m_base + m0 + wr * (16 * MI)) / per_img;
        mm_bnd = (mm_img0 + 1) * per_img - p.m_base;                     // first row of the next image
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const long long gn = n0 + (wc * NJ + j) * 16 + fr;
        if (gn >= p.N) continue;
        const double bv = p.bias ? p.bias[gn] : 0.0;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long gm = m0 + wr * (16 * MI) + i * 16 + fk + 4 * r;
                if (gm < p.M) {
                    const double v = act_f64(acc[i][j][r] + bv, p.act);
                    p.C[gm * p.ldc + gn] = v;
                    if (fold) {
                        if (gm < mm_bnd) { mn0 = fmin(mn0, v); mx0 = fmax(mx0, v); }
                        else { mn1 = fmin(mn1, v); mx1 = fmax(mx1, v); }
                    }
                }
            }
    }
    if (fold) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn0 = fmin(mn0, __shfl_xor(mn0, o)); mx0 = fmax(mx0, __shfl_xor(mx0, o));
            mn1 = fmin(mn1, __shfl_xor(mn1, o)); mx1 = fmax(mx1, __shfl_xor(mx1, o));
        }
        if (lane == 0) {
            if (mn0 <= mx0) {                                            // saw at least one element of that image
                atomicMin(&p.cv.mm_keys[2 * mm_img0], dlc_f64_key(mn0));
                atomicMax(&p.cv.mm_keys[2 * mm_img0 + 1], dlc_f64_key(mx0));
            }
            if (mn1 <= mx1) {
                atomicMin(&p.cv.mm_keys[2 * mm_img0 + 2], dlc_f64_key(mn1));
                atomicMax(&p.cv.mm_keys[2 * mm_img0 + 3], dlc_f64_key(mx1));
            }
        }
    }
}

// (A one-wave-per-SIMD form of this kernel -- 4 waves of 128 x 64, the barrier in the middle of a K tile, every wave
// leaving it with fragments in registers -- is in commit 0d57e84: bit-identical, and 49.5 vs 30.5 ms on SDAV.transform,
// 88.7 vs 36.0 ms on CnnVtl.transform.  The cycle stamps that motivated it (scripts/exp_dma_stamps.py: the older wave of
// a SIMD runs its 64 MFMAs in 5971 cycles and then sits 3755 at the barrier while its partner needs 9329; no wait for
// DMA data at all) are real, but a lone compiler-scheduled wave with 256 accumulator registers spills inside the loop
// and does not keep the matrix pipe fed; two waves covering each other do better.)

template <int BLAYOUT, bool CONV, int NJ, int MI>
int launch_one(dlc_ctx* ctx, const DmaArgs& a, long long nwg, hipStream_t st) {
    auto kern = gemm_dma_f64_kernel<BLAYOUT, CONV, NJ, MI>;
    constexpr int threads = NT3;
    constexpr int lds = NSTAGE * (64 * MI * TK3 * 8 + B_STAGE);          // 144 / 96 KiB
    const unsigned long long m = 1ull << (DLC_ATTR_DMA64_BASE + (CONV ? 2 : (BLAYOUT == DLC_B_KN ? 0 : 1)) + (NJ == 3 ? 3 : 0) +
                                          (MI == 2 ? 6 : (MI == 1 ? 12 : 0)));
    if (!(ctx->func_attr_set & m)) {
        DLC_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        ctx->func_attr_set |= m;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(threads), lds, st, a);
    return DLC_OK;
}

}  // namespace

// One launch.  m_base: CONV, the output pixel index of row 0 (A stays the whole input, C points at row 0's outputs);
// force_tm: 128 / 256 rows per tile; dry: only say whether the launch would be taken (DLC_OK / 1).
static int launch_dma_part(dlc_ctx* ctx, int blayout, int act, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda,
                           const double* B, int64_t ldb, const double* bias, double* C, int64_t ldc, hipStream_t st,
                           const ConvGeom* cv, const TriSkip* tri, int64_t Kb, int64_t m_base, int force_tm, bool dry, double alpha = 0.0,
                           int64_t kchunk = 0, double* partials = nullptr) {
    if (Kb <= 0 || Kb > K) Kb = K;
    // 16-byte pieces: operand rows must start on 16-byte boundaries and K, N be even (a piece = 2 doubles)
    // ([N,K] operands: B's rows are K long and C is stored element by element, so N may be odd there)
    if (!ctx->zero_page || (K & 1) || ((N & 1) && blayout != DLC_B_NK) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) || (ldb & 1)) return 1;
    if (Kb != K && blayout != DLC_B_KN) return 1;          // a shorter B is a [K,N] operand with fewer rows
    if (Kb < (dlc::cdiv(K, (int64_t)TK3) - 1) * TK3) return 1;   // ... whose missing rows all lie in the last K tile
    // per-lane source offsets inside a tile are 32-bit: 256 rows of A, 128 rows ([N,K]) or 16 k-rows ([K,N]) of B
    if (!cv && lda * 8 * TM3 > 0xffffffffll) return 1;
    if (ldb * 8 * (blayout == DLC_B_NK ? TN3 : TK3) + 4096 > 0xffffffffll) return 1;
    if (cv) {
        if (blayout != DLC_B_KN || cv->C % TK3 != 0) return 1;
    } else if (lda & 1) {
        return 1;
    }
    // worth a 256 x 128 tile per CU only when the launch fills the chip a few times over
    // N <= 96 (conv1's 96 filters): 96-column tiles, so that no quarter of the MFMAs works on padding
    const bool narrow = N <= 96;
    const int tn = narrow ? 96 : TN3;
    // tile height: the caller's choice (launch_dma_f64); triangular launches and the 96-column form keep the large tile
    int tm = (((force_tm == TM3 / 2 && !tri) || force_tm == TM3 / 4) && !narrow) ? force_tm : TM3;
    // (conv1's 96-column form on 64-row tiles, two workgroups per CU: CnnVtl.transform 29.1 against 28.75 ms -- not kept)
    const int64_t tiles_m = dlc::cdiv(M, (int64_t)tm), tiles_n = dlc::cdiv(N, (int64_t)tn);
    // From 16 tiles on, and with more than 3/4 of a tile's rows real (scripts/exp_dma_threshold.py: below that the
    // register-staged 128 x 128 kernel's twice as many workgroups win; above it this kernel wins at every size once the
    // tiles of edge blocks are dealt evenly to the shader engines -- it used to be taken from 512 tiles on only)
#ifndef DLC_DMA_MIN_TILES
#define DLC_DMA_MIN_TILES 16
#endif
    // (a single frame's 30 rows on the 128-row tile: SDAV.transform 2.0 -> 1.5 ms -- the register-staged kernel's
    // K loop is the slower one even at a quarter of the tile's rows; the 256-row tile wants 3/4 of its rows real)
    // (the 128-row form has as many workgroups as the register-staged kernel and the faster K loop: it is taken at any
    // tile count -- CnnVtl.transform of 1 / 8 frames 2.0 / 2.6 -> 1.5 ms)
    if ((tm == TM3 && (tiles_m * tiles_n < DLC_DMA_MIN_TILES || M < TM3 * 3 / 4)) || K < 4 * TK3) return 1;
    if (dry) return DLC_OK;
    DmaArgs a;
    a.A = (const char*)A; a.lda_b = lda * 8; a.B = (const char*)B; a.ldb_b = ldb * 8;
    a.bias = bias; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.Kb = Kb; a.act = act; a.alpha = alpha;
    a.cv = cv ? *cv : ConvGeom{};
    a.m_base = m_base;
    a.cv_all_valid = 0;
    if (cv && cv->mm_keys && (int64_t)cv->OH * cv->OW < 64) return 1;     // the epilogue folds at most two images per wave
    if (cv) {
        // per-lane offsets of the A operand are 32-bit and must stay below the descriptor's num_records: a tile's
        // 256 output pixels span at most cdiv(256, OH * OW) + 1 images
        const int64_t img_bytes = (int64_t)cv->H * cv->W * cv->C * 8;
        if ((dlc::cdiv((int64_t)TM3, (int64_t)cv->OH * cv->OW) + 1) * img_bytes >= 0x7ff00000ll) return 1;
        const int64_t KH = K / ((int64_t)cv->KW * cv->C);
        a.cv_all_valid = cv->pad_t == 0 && cv->pad_l == 0 && (int64_t)(cv->OH - 1) * cv->stride + KH <= cv->H &&
                         (int64_t)(cv->OW - 1) * cv->stride + cv->KW <= cv->W;
    }
    a.zero = (const char*)ctx->zero_page;
    a.kchunk = kchunk; a.P = partials;
    const int chunks = kchunk > 0 ? (int)dlc::cdiv(K, kchunk) : 1;
    a.nchunks = chunks;
    a.tiles_m = tiles_m; a.tiles_n = tiles_n;
    a.tri_p = tri ? tri->p : 0; a.tri_row0 = tri ? tri->row0 : 0; a.tri_col0 = tri ? tri->col0 : 0;
    // block of 32 tiles (one XCD's 32 CUs): 4 row tiles x 8 column tiles = 1024 x 1024 outputs, narrower where the
    // matrix has fewer tiles along a dimension (powers of two).  Blocks go round-robin to the XCDs, so a launch of a
    // few blocks can leave some XCDs with twice the tiles of others (conv3 of 128 frames: 195 tiles in 9 blocks -- XCD 0
    // ran two rounds of its 32 CUs while XCDs 1-7 ran one, 19 % on the whole CnnVtl.transform call).  For such launches
    // the block shrinks (rows first) until the busiest XCD needs no more rounds than the tiles / 256 CUs do; L2
    // sharing inside a block matters little when the whole launch is a round or two.
    int bc = 8;
    while (bc > 1 && bc / 2 >= tiles_n) bc /= 2;
    int br = 32 / bc;
    if (tiles_m < br) {
        br = 1;
        while (br < tiles_m) br *= 2;
        bc = 32 / br;
    }
    if (!tri) {
        auto busiest_rounds = [&](int r, int c) {           // rounds the fullest shader engine (8 CUs) of any XCD runs
            const int64_t nr = dlc::cdiv(tiles_m, (int64_t)r), nc = dlc::cdiv(tiles_n, (int64_t)c);
            int64_t load[8][4] = {};
            for (int64_t g = 0; g < nr * nc; ++g) {
                const int64_t rows = std::min<int64_t>(r, tiles_m - (g % nr) * r), cols = std::min<int64_t>(c, tiles_n - (g / nr) * c);
                const bool rows_first = (g % nr) == nr - 1 && (tiles_m & (r - 1)) != 0;
                for (int i = 0; i < r * c; ++i) {
                    const bool exists = rows_first ? (i / c < rows && i % c < cols) : (i % r < rows && i / r < cols);
                    if (exists) load[g & 7][((g >> 3) * (r * c) + i) & 3]++;     // this XCD's ((g >> 3) * block + i)-th workgroup
                }
            }
            int64_t m = 0;
            for (int x = 0; x < 8; ++x)
                for (int e = 0; e < 4; ++e) m = std::max(m, load[x][e]);
            return dlc::cdiv(m, (int64_t)8);
        };
        if (tiles_m * tiles_n <= 32 * 1024) {               // larger launches: dozens of rounds, the block stays
            const int64_t want = dlc::cdiv(tiles_m * tiles_n, (int64_t)256);
            while (br * bc > 1 && busiest_rounds(br, bc) > want) {
                if (br > 1) br /= 2; else bc /= 2;
            }
        }
    }
    int lg_blk = 0;
    while ((1 << lg_blk) < br * bc) ++lg_blk;
    a.lg_blk = lg_blk;
    a.br = br; a.bc = bc;
    a.nbr = dlc::cdiv(tiles_m, (int64_t)br);
    a.tri_nbc = dlc::cdiv(tiles_n, (int64_t)bc);
    a.nblocks = a.nbr * a.tri_nbc;
    a.tri_blk_cols = bc * tn; a.tri_rem0 = 0; a.tri_step = 0; a.tri_lg_rows = 0;
    if (a.tri_p > 0) {
        while ((1 << a.tri_lg_rows) < br * tm) ++a.tri_lg_rows;         // br is a power of two, and so is the tile height
        a.tri_step = a.tri_blk_cols % a.tri_p;
        TriWalk tw{a.tri_col0 + a.tri_blk_cols - 1, (int)((a.tri_col0 + a.tri_blk_cols - 1) % a.tri_p)};
        a.tri_rem0 = tw.rem;
        a.nblocks = 0;
        for (long long c = 0; c < a.tri_nbc; ++c, tw.step(a)) a.nblocks += tw.count(a);
        if (a.nblocks == 0) return DLC_OK;                               // nothing wanted (the caller never reads this block)
    }
    const long long nwg = kchunk > 0 ? tiles_m * tiles_n * chunks : (dlc::cdiv(a.nblocks, (int64_t)8) * 8) << a.lg_blk;
    if (nwg > 0x7fffffffll) return 1;
    const int prof_slot = (int)(ctx->prof_calls % DLC_PROFILE_RING);
    if (ctx->profiling) DLC_HIP_CHECK(ctx, hipEventRecord(ctx->ev_start[prof_slot], st));
    int rc;
    if (tm == TM3) {
        if (cv) rc = narrow ? launch_one<DLC_B_KN, true, 3, 4>(ctx, a, nwg, st) : launch_one<DLC_B_KN, true, 4, 4>(ctx, a, nwg, st);
        else if (blayout == DLC_B_KN) rc = narrow ? launch_one<DLC_B_KN, false, 3, 4>(ctx, a, nwg, st) : launch_one<DLC_B_KN, false, 4, 4>(ctx, a, nwg, st);
        else rc = narrow ? launch_one<DLC_B_NK, false, 3, 4>(ctx, a, nwg, st) : launch_one<DLC_B_NK, false, 4, 4>(ctx, a, nwg, st);
    } else if (tm == TM3 / 2) {
        if (cv) rc = launch_one<DLC_B_KN, true, 4, 2>(ctx, a, nwg, st);
        else if (blayout == DLC_B_KN) rc = launch_one<DLC_B_KN, false, 4, 2>(ctx, a, nwg, st);
        else rc = launch_one<DLC_B_NK, false, 4, 2>(ctx, a, nwg, st);
    } else {
        if (cv) rc = launch_one<DLC_B_KN, true, 4, 1>(ctx, a, nwg, st);
        else if (blayout == DLC_B_KN) rc = launch_one<DLC_B_KN, false, 4, 1>(ctx, a, nwg, st);
        else rc = launch_one<DLC_B_NK, false, 4, 1>(ctx, a, nwg, st);
    }
    if (rc != DLC_OK) return rc;
    DLC_LAUNCH_CHECK(ctx, "gemm_dma_f64_kernel");
    if (ctx->profiling) {
        DLC_HIP_CHECK(ctx, hipEventRecord(ctx->ev_stop[prof_slot], st));
        ctx->prof_calls++;
    }
    return DLC_OK;
}

// The public entry chooses between four forms by the rounds of the chip's 256 CUs each would take (a 128-row tile
// costs 0.51 of a 256-row tile at full occupancy -- SDAV.transform on 128-row tiles only: 28.3 against 27.8 ms; a
// 64-row tile 0.248, two of its workgroups sharing a CU -- 26.8 ms; 0.27 in the convolution form):
//   one launch of 256-row tiles;  one of 128-row tiles;  one of 64-row tiles (small launches: a workgroup's K loop is
//   what they wait for; and plain operands at any size);  or 256-row tiles for the whole rounds and a second launch of
//   128-row tiles for the rows behind them (conv3-5 of 1063 frames: 6.3 / 6.3 / 4.2 rounds).
// Rows are independent and every form sums k in the same order: the same bits whichever is taken.
int launch_dma_f64(dlc_ctx* ctx, int blayout, int act, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda,
                   const double* B, int64_t ldb, const double* bias, double* C, int64_t ldc, hipStream_t st,
                   const ConvGeom* cv, const TriSkip* tri, int64_t Kb, double alpha) {
    int tm = TM3;
    if (!tri && N > 96) {
        constexpr double HALF = 0.51;
        const int64_t tn_ = dlc::cdiv(N, (int64_t)TN3), t4 = dlc::cdiv(M, (int64_t)TM3) * tn_;
        const int64_t t2 = dlc::cdiv(M, (int64_t)(TM3 / 2)) * tn_;
        const double whole4 = (double)dlc::cdiv(t4, (int64_t)256);
        const double whole2 = HALF * (double)dlc::cdiv(t2, (int64_t)256) + 0.01;
        const double quarter = (cv ? 0.27 : 0.248) * (double)dlc::cdiv(dlc::cdiv(M, (int64_t)(TM3 / 4)) * tn_, (int64_t)256) + 0.02;
        if (t4 > 256 && t4 % 256 != 0) {
            const int64_t rm = (t4 / 256) * 256 / tn_;                    // row tiles of the main launch
            const int64_t m1 = rm * TM3, m2 = M - m1;
            if (rm > 0 && m2 > 0) {
                const int64_t tt = dlc::cdiv(m2, (int64_t)(TM3 / 2)) * tn_;
                const double split = (double)dlc::cdiv(rm * tn_, (int64_t)256) + HALF * (double)dlc::cdiv(tt, (int64_t)256) + 0.05;
                const double* a2 = cv ? A : A + m1 * lda;
                if (split < whole4 - 0.15 && split < whole2 - 0.03 && split < quarter - 0.03 &&
                    launch_dma_part(ctx, blayout, act, m1, N, K, A, lda, B, ldb, bias, C, ldc, st, cv, tri, Kb, 0, TM3, true, alpha) == DLC_OK &&
                    launch_dma_part(ctx, blayout, act, m2, N, K, a2, lda, B, ldb, bias, C + m1 * ldc, ldc, st, cv, tri, Kb, cv ? m1 : 0, TM3 / 2, true, alpha) == DLC_OK) {
                    const int rc = launch_dma_part(ctx, blayout, act, m1, N, K, A, lda, B, ldb, bias, C, ldc, st, cv, tri, Kb, 0, TM3, false, alpha);
                    if (rc != DLC_OK) return rc;
                    return launch_dma_part(ctx, blayout, act, m2, N, K, a2, lda, B, ldb, bias, C + m1 * ldc, ldc, st, cv, tri, Kb, cv ? m1 : 0, TM3 / 2, false, alpha);
                }
            }
        }
        if (whole2 < whole4 - 0.1) tm = TM3 / 2;
        // 64-row tiles (two workgroups fit a CU) while they leave the chip under one round: the K loop of a workgroup is
        // what a small launch waits for, and a quarter of the MFMAs per K tile shortens it
        // ... and, for plain operands, whenever their round count comes out lower: two 64-row workgroups share a CU
        // and cover each other's prologues, epilogues and barriers -- SDAV.transform of 1063 frames on 64-row tiles
        // only: 26.8 against 27.8 ms, i.e. 0.248 of a 256-row tile each; the convolution form pays its per-tile
        // tap bookkeeping four times over (CnnVtl.transform 29.2 against 28.8 ms): 0.27
        const int64_t t1 = dlc::cdiv(M, (int64_t)(TM3 / 4)) * tn_;
        const double whole1 = (cv ? 0.27 : 0.248) * (double)dlc::cdiv(t1, (int64_t)256) + 0.02;
        if (whole1 < whole4 - 0.1 && whole1 < whole2 - 0.05) tm = TM3 / 4;
    }
    // the Gram blocks of the similarity: 64-row tiles for the same reason (similarity of 1063 frames 39.8 -> 39.0 ms)
    if (tri && N > 96 && dlc::cdiv(M, (int64_t)TM3) * dlc::cdiv(N, (int64_t)TN3) > 512) tm = TM3 / 4;
    return launch_dma_part(ctx, blayout, act, M, N, K, A, lda, B, ldb, bias, C, ldc, st, cv, tri, Kb, 0, tm, false, alpha);
}

// Split-K on 64-row tiles (plain operands, no triangle): `chunks` chunks of kchunk (a multiple of the K tile) into the
// partial tiles P[chunks][M][N]; the caller sums them in chunk order and applies bias + activation (splitk_reduce_f64,
// gemm_dense.hip).  DLC_OK, or 1 when the shape / alignment is not one the kernel handles (dry: only say which).
int gemm_dma_f64_splitk(dlc_ctx* ctx, int blayout, int64_t M, int64_t N, int64_t K, int64_t Kb, const double* A, int64_t lda,
                        const double* B, int64_t ldb, double* partials, int64_t kchunk, hipStream_t st, bool dry) {
    if (kchunk <= 0 || kchunk % TK3 != 0 || kchunk < 4 * TK3 || !partials) return 1;
    if (N <= 96) return 1;                                   // (the 96-column form keeps the 256-row tile)
    return launch_dma_part(ctx, blayout, DLC_ACT_NONE, M, N, K, A, lda, B, ldb, nullptr, partials, N, st, nullptr, nullptr, Kb, 0,
                           TM3 / 4, dry, 0.0, kchunk, partials);
}

// C += alpha * (A . B) in the epilogue of the LDS-DMA kernel (the SGD step of a weight gradient: W -= lr * dW without dW
// ever reaching memory -- SDAV.py:223-226).  DLC_OK, or 1 when the shape / alignment is not one the kernel handles.
int gemm_axpy_dma_f64(dlc_ctx* ctx, int blayout, double alpha, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda,
                      const double* B, int64_t ldb, double* C, int64_t ldc, hipStream_t st) {
    return launch_dma_f64(ctx, blayout, ACT_AXPY, M, N, K, A, lda, B, ldb, nullptr, C, ldc, st, nullptr, nullptr, 0, alpha);
}


}  // namespace dlc_gemm
