// The SDAV similarity's patch matching as an exact-arithmetic FILTER (match_ref.hip, SimilarityCalculator.py:30-37).
//
// What the reference needs from the 31 890 x 31 890 patch products is one thing only: for patch a of frame i, WHICH
// patch b of frame j is nearest (np.argmin of the norms).  The distances themselves never reach the result.  So the
// Gram matrix does not have to be an fp64 product -- it has to decide the arg-min, and say when it cannot:
//
//   v_k = (x_k - c_k) * 0.996 / s in [-0.498, 0.498]   c_k: the midpoint of COLUMN k's extremes over the dataset, s: the
//         largest column range.  |x_a - x_b| does not change under a per-column offset, so arg-min |v_a - v_b| = arg-min
//         |x_a - x_b|, and the fixed-point bits go where the data varies.  (r03 spent them on (x - lo) / (hi - lo) of the
//         WHOLE dataset: low-contrast descriptors -- every column within 1e-3 of its own mean, the means spread over
//         [0.15, 0.88]: real frames through 1/sqrt(fan_in) weights -- had every one of their arg-mins inside the error
//         window.)
//   q = rint(v * 2^24) as three SIGNED 8-bit digits, q = s1 2^16 + s2 2^8 + s3, s_i in [-128, 127]
//   acc = C2 + floor((C3 + floor(C4 / 256)) / 256),  Cc = sum over i + j = c of s_i . s_j'   (v_mfma_i32_16x16x64_i8: exact)
//
// acc 2^-15 is 2 v_a . v_b to within
//   2^-24 (sum |v_a| + sum |v_b|) + H 2^-46  [rounding of v]  +  H (2^-24 + 2^-33)  [the dropped classes s2 s3', s3 s3']
//   + 1.004 2^-15  [the two floors],
// a rigorous bound with no rounding in it (integer accumulation).  The product kernel takes the arg-min of
// d2 = |v_b|^2 - acc 2^-15 (units of 2^-15; |v_b|^2 rounded to them: 2^-16 more) and accepts it when the runner-up is
// more than twice that bound away (dlc_sim_window, gemm_internal.h); otherwise it leaves the candidates inside the window
// to be evaluated directly in fp64 from the descriptors, near-ties in NumPy's own summation order (match_ref.hip).
// Signed digits and the midpoint offset carry 24 bits in the three bytes r03 carried 21 in: the window is 6.8e-4 of the
// unit square distance at H = 2500 where r03's was 7.4e-3.
// Six int8 products of K = H replace one fp64 product: a sixth of its time as measured (6 ms against 36.5 at 1063
// frames), and the result is the arg-min of the true distances either way.
// Test infrastructure never enters: the oracle (oracle/similarity.py) only checks the outcome in tests/.
//
// Layout: one panel, X = (s1 | s2 | s3) along K (Kp = H rounded up to 256, zero padded), tiled the way the MFMA reads
// it: [16-row group][k-step of 64 bytes: slice-major, 3 Kp / 64 of them][lane l: row l % 16, bytes (l / 16) * 16 .. + 15] --
// every row-side LDS-DMA piece is 1 KiB of consecutive bytes, eight whole cache lines.  Row patches and column patches are
// rows of the same panel: a column-side piece gathers 16 consecutive rows that may begin inside a group (GramI8Args).  (Row-major slices made each piece 16 half lines; every line crossed the L2 -> L1 path twice, once
// per k-step, and the kernel sat at 12 B / clock / CU.)
//
// r03: ONE sweep over K with three accumulator sets.  A k-step brings the 64 bytes of all three slices of the tile's rows
// and columns and feeds the six slice products of the three classes at once -- C2 += s1.s1', C3 += s2.s1' + s1.s2',
// C4 += s3.s1' + s2.s2' + s1.s3' -- and the epilogue forms acc = C2 + ((C3 + (C4 >> 8)) >> 8) (arithmetic shifts: floors).  The r02 kernel had one accumulator set and walked K three times (classes 4, 3, 2 with a shift in between):
// 240 k-steps of 32 MFMAs per wave instead of 40 of 96, every slice streamed again per class (18.6 GB over the fabric
// for 0.49 GB of panels, profiles/r02i), 12 LDS fragment reads per 32 MFMAs instead of 24 per 96.
#include "gemm_internal.h"

namespace dlc_gemm {
namespace {

typedef __attribute__((ext_vector_type(4))) int v4i;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int GI_T = 128;                 // tile: 128 row patches x 128 column patches, 4 waves of 64 x 64
constexpr int GI_KS = 64;                 // bytes of K (per slice) per LDS stage = one MFMA k-step
constexpr int GI_KPAD = 256;              // the slices' padded length is a multiple of this
constexpr int GI_HALF = GI_T * GI_KS * 3; // one operand's part of a stage: 8 groups x 3 slices x 1 KiB = 24 KiB
constexpr int GI_STAGE = 2 * GI_HALF;     // 48 KiB
constexpr int GI_NSTAGE = 3;              // 144 KiB: two stages in flight behind the one being read
constexpr int GI_BR = 4, GI_BC = 8;       // an XCD's 32 resident workgroups take one block of 4 x 8 tiles

struct GramI8Args {
    const char* X;                        // THE panel: patch rows in order, groups of 16 (+ at least one group of zero rows
                                          // from row `zrow` on).  Row operands are its groups as they lie; COLUMN operands are
                                          // gathered from it, lane by lane, in UNITS of 64 columns = fpu whole frames + zero
                                          // rows, so that every frame's P columns lie inside one wave's 64-column block
    long long zrow;                       // first row of an all-zero group of X
    const int* nbp;                       // |v_b|^2 of the column units' rows, units of 2^-15 (unit layout)
    const unsigned long long* keys;       // [3]: largest row sum of |v| (the error bound), [2]: non-finite / exit flag
    unsigned char* abi;                   // out [nfp, rp]: the nearest patch b of column frame j to row patch a
    unsigned* acand;                      // out [nfp, rp]: 0 = decided; else the patches inside the error window (bit b)
    long long nfp, rp;                    // column frames of abi / acand (padded: whole tiles) and their pitch (row patches)
    long long nrows, nframes;
    long long gpitch;                     // bytes of one 16-row group: 3 Kp / 64 k-steps of 1 KiB
    int kp, H, P, fpu;
    int tiles_m, tiles_n, nsm, nsn, nsup;
    const int2* blk;                                // [nsup] (block row, block column) of the wanted blocks, row by row
    // a STRIP of the triangle (the streaming form's batches, gram_argmin_i8_strip): only the column tiles tn_lo .. tn_hi are
    // wanted, and abi / acand hold the column frames from fj_base on (row fj - fj_base).  The whole triangle: 0, INT_MAX, 0.
    int tn_lo, tn_hi;
    long long fj_base;
    int strip_cols, sj_lo;                          // > 0: a strip -- no table, block `want` is (want / strip_cols, sj_lo + want % strip_cols)
};

// The triangle's wanted blocks.  Block row si (GI_BR tiles of rows) wants the block columns from the one that holds the
// column tile of frame (first row frame of the block row) + 1 -- a unit holds fpu frames, a tile two units.
__host__ __device__ inline int gi_first_block_col(int si, int P, int fpu) {
    const long long row_frame = ((long long)si * GI_BR * GI_T) / P;
    return (int)(((row_frame + 1) / fpu / 2) / GI_BC);
}

// blk[nsup]: the wanted blocks numbered row by row (the workgroups of gram_i8_kernel look theirs up: r03 first had every
// workgroup walk the block rows itself -- up to nsm iterations of two 64-bit divisions on the scalar unit, ~10 us of a
// 38 us tile).  rowstart: [nsm + 1] scratch.  One workgroup of 256 threads.
__global__ void gram_blocks_kernel(int nsm, int nsn, int P, int fpu, int* rowstart, int2* blk) {
    for (int si = threadIdx.x; si < nsm; si += blockDim.x) {
        const int cnt = nsn - gi_first_block_col(si, P, fpu);
        rowstart[si + 1] = cnt > 0 ? cnt : 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        rowstart[0] = 0;
        for (int si = 0; si < nsm; ++si) rowstart[si + 1] += rowstart[si];
    }
    __syncthreads();
    for (int si = threadIdx.x; si < nsm; si += blockDim.x) {
        const int first = gi_first_block_col(si, P, fpu), o = rowstart[si];
        for (int sj = first; sj < nsn; ++sj) blk[o + sj - first] = make_int2(si, sj);
    }
}

// ---- the accumulators: FIXED accumulation registers a0 .. a191, named in the instructions themselves ---------------------
// Class c (0: C2, 1: C3, 2: C4), column group j, row group i: a[B : B + 3], B = ((c * 4 + j) * 4 + i) * 4.  hipcc neither
// allocates nor moves them: as "+a" operands of the inline-asm MFMAs it did both -- at the seams between the k loop and the
// steps behind it it shuffled accumulators between registers (v_accvgpr_read / _mov / _write) DIRECTLY behind the MFMAs that
// wrote them.  An MFMA in inline asm is opaque to hipcc's hazard recognizer, so no wait states went between them, and an
// accumulator read that early returns the value from before the MFMA: a build whose allocation happened to do that to the
// youngest accumulators lost part of the last k-step.  The wait states are now written out (gi_acc_settle), and `make`
// checks the object code (check_m0.py): no v_accvgpr_* instruction in the kernel but the 192 zero writes and the 192 reads
// of the epilogue.
#define GI_CL10(d) "a" #d "0", "a" #d "1", "a" #d "2", "a" #d "3", "a" #d "4", "a" #d "5", "a" #d "6", "a" #d "7", "a" #d "8", "a" #d "9"
#define GI_CL_ALL "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", GI_CL10(1), GI_CL10(2), GI_CL10(3), GI_CL10(4),  \
    GI_CL10(5), GI_CL10(6), GI_CL10(7), GI_CL10(8), GI_CL10(9), GI_CL10(10), GI_CL10(11), GI_CL10(12), GI_CL10(13),            \
    GI_CL10(14), GI_CL10(15), GI_CL10(16), GI_CL10(17), GI_CL10(18), "a190", "a191"
template <int B> __device__ __forceinline__ void gi_mfma(const v4i& a, const v4i& b) {
    asm volatile("v_mfma_i32_16x16x64_i8 a[%2:%3], %0, %1, a[%2:%3]" : : "v"(a), "v"(b), "n"(B), "n"(B + 3));
}
template <int B> __device__ __forceinline__ void gi_acc_zero16() {          // a[B : B + 15] = 0
    asm volatile("v_accvgpr_write_b32 a[%0], 0\n\tv_accvgpr_write_b32 a[%1], 0\n\tv_accvgpr_write_b32 a[%2], 0\n\tv_accvgpr_write_b32 a[%3], 0\n\t"
                 "v_accvgpr_write_b32 a[%4], 0\n\tv_accvgpr_write_b32 a[%5], 0\n\tv_accvgpr_write_b32 a[%6], 0\n\tv_accvgpr_write_b32 a[%7], 0\n\t"
                 "v_accvgpr_write_b32 a[%8], 0\n\tv_accvgpr_write_b32 a[%9], 0\n\tv_accvgpr_write_b32 a[%10], 0\n\tv_accvgpr_write_b32 a[%11], 0\n\t"
                 "v_accvgpr_write_b32 a[%12], 0\n\tv_accvgpr_write_b32 a[%13], 0\n\tv_accvgpr_write_b32 a[%14], 0\n\tv_accvgpr_write_b32 a[%15], 0"
                 : : "n"(B), "n"(B + 1), "n"(B + 2), "n"(B + 3), "n"(B + 4), "n"(B + 5), "n"(B + 6), "n"(B + 7), "n"(B + 8), "n"(B + 9),
                     "n"(B + 10), "n"(B + 11), "n"(B + 12), "n"(B + 13), "n"(B + 14), "n"(B + 15));
}
// all 192 to zero; the one statement whose clobber list tells hipcc that the kernel uses a0 .. a191 at all
__device__ __forceinline__ void gi_acc_zero_all() {
    asm volatile("s_nop 0" : : : GI_CL_ALL);
    gi_acc_zero16<0>(); gi_acc_zero16<16>(); gi_acc_zero16<32>(); gi_acc_zero16<48>(); gi_acc_zero16<64>(); gi_acc_zero16<80>();
    gi_acc_zero16<96>(); gi_acc_zero16<112>(); gi_acc_zero16<128>(); gi_acc_zero16<144>(); gi_acc_zero16<160>(); gi_acc_zero16<176>();
    asm volatile("s_nop 7" : : : GI_CL_ALL);        // (v_accvgpr_write -> MFMA reading it as its C operand: wait states)
}
// behind the last MFMA, before the first accumulator read: longer than an MFMA's latency (4 passes = 16 cycles + write-back)
__device__ __forceinline__ void gi_acc_settle() { asm volatile("s_nop 15\n\ts_nop 15" : : : GI_CL_ALL); }
template <int B> __device__ __forceinline__ v4i gi_acc_read() {
    v4i r;
    asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\tv_accvgpr_read_b32 %2, a[%6]\n\tv_accvgpr_read_b32 %3, a[%7]"
                 : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : "n"(B), "n"(B + 1), "n"(B + 2), "n"(B + 3));
    return r;
}
#define GI_ACC(C, J, I) ((((C) * 4 + (J)) * 4 + (I)) * 4)

// One 1 KiB LDS-DMA piece: lane l fetches bytes l * 16 .. + 15 behind the wave-uniform base `src` (+ voff, which carries
// the k-step) into LDS at `lds` + l * 16.  Inline asm so that hipcc does not count it in vmcnt (it would wait for
// vmcnt(0) in front of every LDS read).  M0 carries the LDS destination and is declared CLOBBERED, not saved and restored
// (two more instructions per piece, twelve pieces per k-step of a wave that has nothing else to hide them behind): hipcc
// warns that M0 is a reserved register it may not preserve across the statement, so the BUILD checks what the claim rests
// on -- `make` runs csrc/check_m0.py, which disassembles this kernel and fails if any instruction outside these asm
// statements reads or writes M0.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma1(unsigned voff, const char* src, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(src), "s"(lds) : "memory", "m0");
}
#pragma clang diagnostic pop

__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}

// LDS stage (48 KiB): the row panel's 8 groups of 16 rows x 3 slices, a 1 KiB block each in MFMA operand order -- block
// (g * 3 + s), lane l of a block holds row l % 16, bytes (l / 16) * 16 .. + 15 of the k-step of slice s -- then the
// column panel's the same.  A fragment read is one ds_read_b128 at block + lane * 16.
//
// Pipeline: three stages; k-step t multiplies the fragments of stage t, which it read from LDS during k-step t-1, while
// it reads those of stage t+1 and issues the DMA of stage t+3 into stage t's slot -- ONE wave per SIMD (192 accumulators in
// a0 .. a191, 128 fragment registers), so the interleaving inside the wave is what hides the LDS and DMA latencies: one
// fragment read behind every second MFMA, one DMA piece behind every fourth.  Barrier t says "stage t+1 has landed
// everywhere and everybody is through reading stage t".
// One workgroup per tile.  A persistent form (256 workgroups walking a tile list, the stages one stream across the tiles:
// no prologue, no workgroup turnover) was built and measured in r03: per tile it saves 10 000 of 100 000 cycles and gives
// them back -- the chip holds 2.22 GHz under it instead of 2.35 GHz, the epilogue has to work in 16 KiB instead of in the
// dead stages (docs/LAB.md 9) -- 5.45 ms against this form's 5.27.
__global__ __launch_bounds__(256) void gram_i8_kernel(const GramI8Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem_i8[];
    // workgroup -> tile: ids go round-robin to the 8 XCDs; an XCD's 32 resident workgroups take one GI_BR x GI_BC block
    // of tiles (12 panels feed 32 tiles out of that XCD's L2).  Only blocks with a wanted tile are numbered, row by row,
    // and dealt to the XCDs in turn: dealt by block column, the triangle gave XCD 7 2.4 times the work of XCD 0.
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3, local = slot & 31;
    const int want = (slot >> 5) * 8 + xcd;           // index among the wanted blocks: p.blk (gram_blocks_kernel) names it
    if (want >= p.nsup) return;
    // the triangle's blocks come from the table; a strip's are every block row x its strip_cols block columns from sj_lo on
    const int2 blk = p.strip_cols ? make_int2(want / p.strip_cols, p.sj_lo + want % p.strip_cols) : p.blk[want];
    const int si = blk.x, sj = blk.y;
    const int tile_m = si * GI_BR + (local >> 3), tile_n = sj * GI_BC + (local & 7);
    if (tile_m >= p.tiles_m || tile_n >= p.tiles_n || tile_n < p.tn_lo || tile_n > p.tn_hi) return;
    const long long m0 = (long long)tile_m * GI_T, n0 = (long long)tile_n * GI_T;
    // the tile's last frame must lie behind its first row's frame, and its first frame must exist
    if ((long long)(2 * tile_n + 2) * p.fpu - 1 <= m0 / p.P || (long long)2 * tile_n * p.fpu >= p.nframes) return;
    if (p.keys[2]) return;                          // a NaN / infinity in the dataset, or the sample said "hopeless": not this form

    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wr = w >> 1, wc = w & 1;              // 64 row patches x 64 column patches per wave
    const unsigned lds_base = (unsigned)(unsigned long long)(lptr_t)smem_i8;
    const int n64 = p.kp / GI_KS;                   // k-steps (per slice): a multiple of 4, at least 4
    // this wave's DMA share of a stage: groups 2 w, 2 w + 1 of each panel, three slices each = 12 pieces.  Piece b:
    // panel b / 6, group (b % 6) / 3, slice b % 3; its source base is wave-uniform, the k-step rides in the lanes' offset.
    // Row pieces (b < 6): one uniform base each, all lanes at lane * 16 behind it.  Column pieces (b >= 6): column c of the
    // tile is row (c / 64) * fpu * P + c % 64 of the panel when c % 64 < fpu * P and its frame exists, a zero row otherwise
    // -- 16 consecutive rows that begin anywhere in a group, so every lane carries its own offset from X (the panel is
    // smaller than 4 GiB: sim_filter_fits): the same bytes serve as row and as column operands, there is no second panel.
    const char* src[6];
#pragma unroll
    for (int b = 0; b < 6; ++b)
        src[b] = uniform_ptr(p.X + (m0 / 16) * p.gpitch + (long long)(w * 2 + b / 3) * p.gpitch + (long long)(b % 3) * n64 * 1024);
    const char* xbase = uniform_ptr(p.X);
    unsigned yoff[2];                               // this lane's offset of slice 0, k-step 0 for the wave's two column groups
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
        const long long c = n0 + (w * 2 + gi) * 16 + (lane & 15);
        const long long u = c >> 6;
        const int o = (int)(c & 63);
        const bool valid = o < p.fpu * p.P && u * p.fpu + o / p.P < p.nframes;
        const long long xr = valid ? u * p.fpu * p.P + o : p.zrow;
        yoff[gi] = (unsigned)((xr >> 4) * p.gpitch + (((lane >> 4) * 16 + (int)(xr & 15)) * 16));
    }
    const unsigned slice_bytes = (unsigned)n64 * 1024u;
    // its LDS destination inside a stage: block ((2 w + group) * 3 + slice) of the panel's half
    const unsigned lds_w = __builtin_amdgcn_readfirstlane(lds_base + w * 6 * 1024);
    unsigned voff_issue = lane * 16;                // lanes' offset of the next stage to fetch: + 1 KiB per k-step
    unsigned kbytes = 0;                            // ... the same without the lane's part (uniform)
    int is_slot = 0;                                // ... and the slot it goes to
    auto issue_piece = [&](int b) {
        const unsigned lds = lds_w + is_slot * GI_STAGE + (b / 6) * GI_HALF + (b % 6) * 1024;
        if (b < 6) dma1(voff_issue, src[b], lds);
        else dma1(yoff[(b - 6) / 3] + (unsigned)((b - 6) % 3) * slice_bytes + kbytes, xbase, lds);
    };
    auto issue_done = [&]() {
        voff_issue += 1024;
        kbytes += 1024;
        is_slot = is_slot == GI_NSTAGE - 1 ? 0 : is_slot + 1;
    };

    gi_acc_zero_all();                              // the three classes' accumulators: a0 .. a191 (GI_ACC above)
    // fragments [slice][group]: ONE set of the row panel's and of the column panel's slice 2, reloaded in place where they
    // die; the column panel's slices 0 and 1 double-buffered (yb[buffer][slice]) -- 128 registers
    v4i fx[3][4], fy2[4], yb[2][2][4];

    // fragment (slice s, group g) of this wave: row panel block ((wr * 4 + g) * 3 + s), column panel the same with wc
    const char* sx = smem_i8 + (wr * 4) * 3 * 1024 + lane * 16;
    const char* sy = smem_i8 + GI_HALF + (wc * 4) * 3 * 1024 + lane * 16;
#define GI_RDX(S, SO) _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) fx[S][g_] = *(const v4i*)(sx + (SO) + (g_ * 3 + (S)) * 1024)
#define GI_RDY2(SO) _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) fy2[g_] = *(const v4i*)(sy + (SO) + (g_ * 3 + 2) * 1024)
#define GI_RDYB(BUF, S, SO) _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) yb[BUF][S][g_] = *(const v4i*)(sy + (SO) + (g_ * 3 + (S)) * 1024)
    // four MFMAs: column group J of FY against the four row groups of slice SX, into class C
#define GI_ROW2A(C, J, FY, SX) gi_mfma<GI_ACC(C, J, 0)>(FY[J], fx[SX][0]); gi_mfma<GI_ACC(C, J, 1)>(FY[J], fx[SX][1]);
#define GI_ROW2B(C, J, FY, SX) gi_mfma<GI_ACC(C, J, 2)>(FY[J], fx[SX][2]); gi_mfma<GI_ACC(C, J, 3)>(FY[J], fx[SX][3]);
    // one fragment read (a ds_read_b128), pinned where it stands between the MFMAs
#define GI_RD(DST, PTR) { __builtin_amdgcn_sched_barrier(0); DST = *(const v4i*)(PTR); __builtin_amdgcn_sched_barrier(0); }
#define GI_PX(S, G, SO) (sx + (SO) + ((G) * 3 + (S)) * 1024)
#define GI_PY(S, G, SO) (sy + (SO) + ((G) * 3 + (S)) * 1024)
#define GI_NONE(J)
    // sixteen MFMAs of one slice product: column fragments FY (four groups) against row slice SX into class C
#define GI_PROD(C, FY, SX)                                                                                      \
    GI_ROW2A(C, 0, FY, SX) GI_ROW2B(C, 0, FY, SX) GI_ROW2A(C, 1, FY, SX) GI_ROW2B(C, 1, FY, SX)                 \
    GI_ROW2A(C, 2, FY, SX) GI_ROW2B(C, 2, FY, SX) GI_ROW2A(C, 3, FY, SX) GI_ROW2B(C, 3, FY, SX)
    // ... with fragment reads and DMA pieces between them.  Behind every fourth MFMA a DMA piece of the stage three ahead (a
    // piece is three instructions, which fit in the shadow of the MFMA in front of them; two five-instruction pieces behind
    // every fourth MFMA left the matrix pipe idle for their issue time); behind the second (and, RD2, the fourth) MFMA of
    // every four a fragment read of the next stage -- NOT in bursts between the products: the four waves leave the barrier
    // together, sixteen reads each were 64 KiB queued at the LDS at once, and a wave whose read is not accepted yet cannot
    // issue the MFMA behind it either (k loop 88 -> 77 thousand cycles per tile).  RD1 / RD2: macros of the group J.
#define GI_PROD_IL1(C, J, FY, SX, B0, ISSUE, RD1, RD2)                                                          \
    GI_ROW2A(C, J, FY, SX) RD1(J) GI_ROW2B(C, J, FY, SX) RD2(J) if (ISSUE) issue_piece((B0) + (J));
#define GI_PROD_IL(C, FY, SX, B0, ISSUE, RD1, RD2)                                                              \
    GI_PROD_IL1(C, 0, FY, SX, B0, ISSUE, RD1, RD2) GI_PROD_IL1(C, 1, FY, SX, B0, ISSUE, RD1, RD2)               \
    GI_PROD_IL1(C, 2, FY, SX, B0, ISSUE, RD1, RD2) GI_PROD_IL1(C, 3, FY, SX, B0, ISSUE, RD1, RD2)
    // s_waitcnt immediate (gfx9: vmcnt [3:0] + [15:14], expcnt [6:4] left at 7, lgkmcnt [11:8]) with lgkmcnt(0); the
    // builtin, not inline asm, so that hipcc's own wait insertion knows the LDS reads are done
    constexpr int GI_WAIT_VM12 = 0x007c, GI_WAIT_VM0 = 0x0070;
    // One k-step.  The six slice products run in the order (y0 x0) (y1 x0) (y2 x0) | (y0 x1) (y0 x2) (y1 x1).  Behind the third
    // one x0 and y2 are dead and take the NEXT stage's fragments in place; y0 and y1 of the next stage go to the other
    // buffer of the pair; x2 is reloaded during the sixth product and x1 during the first of its own k-step.  Every
    // reload is issued at least 32 MFMAs (512 cycles) before its first use.  The workgroup's one barrier sits behind the
    // third product: "stage t+1 has landed everywhere, and everybody has read the last of stage t" (x1, during the first
    // product), so behind it the DMA of stage t+3 may overwrite stage t's slot.
    int cur = 0, nxt = 1;                           // LDS slots of stage t and stage t+1
#define GI_STEP(C, N, WAIT, ISSUE)                                                                              \
    {                                                                                                           \
        const int so_c = cur * GI_STAGE, so_n = nxt * GI_STAGE;                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        GI_PROD_IL(0, yb[C][0], 0, 0, false, GI_RD_X1, GI_NONE)                                                 \
        GI_PROD(1, yb[C][1], 0)                                                                                 \
        GI_PROD(2, fy2, 0)                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        __builtin_amdgcn_s_waitcnt(WAIT);           /* this wave's pieces of stage t+1 */                        \
        asm volatile("" ::: "memory");                                                                          \
        __builtin_amdgcn_s_barrier();                                                                           \
        asm volatile("" ::: "memory");                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        if (N == 0) {                                                                                           \
            GI_PROD_IL(1, yb[C][0], 1, 0, ISSUE, GI_RD_Y0A, GI_RD_X0)                                           \
            GI_PROD_IL(2, yb[C][0], 2, 4, ISSUE, GI_RD_Y1A, GI_RD_Y2)                                           \
        } else {                                                                                                \
            GI_PROD_IL(1, yb[C][0], 1, 0, ISSUE, GI_RD_Y0B, GI_RD_X0)                                           \
            GI_PROD_IL(2, yb[C][0], 2, 4, ISSUE, GI_RD_Y1B, GI_RD_Y2)                                           \
        }                                                                                                       \
        GI_PROD_IL(2, yb[C][1], 1, 8, ISSUE, GI_RD_X2, GI_NONE)                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        if (ISSUE) issue_done();                                                                                \
        cur = nxt;                                                                                              \
        nxt = nxt == GI_NSTAGE - 1 ? 0 : nxt + 1;                                                               \
    }
    // the reads of a k-step, by the group J they fetch (so_c / so_n: the LDS offsets of stage t / t+1 inside GI_STEP)
#define GI_RD_X1(J) GI_RD(fx[1][J], GI_PX(1, J, so_c))
#define GI_RD_X0(J) GI_RD(fx[0][J], GI_PX(0, J, so_n))
#define GI_RD_X2(J) GI_RD(fx[2][J], GI_PX(2, J, so_n))
#define GI_RD_Y2(J) GI_RD(fy2[J], GI_PY(2, J, so_n))
#define GI_RD_Y0A(J) GI_RD(yb[0][0][J], GI_PY(0, J, so_n))
#define GI_RD_Y1A(J) GI_RD(yb[0][1][J], GI_PY(1, J, so_n))
#define GI_RD_Y0B(J) GI_RD(yb[1][0][J], GI_PY(0, J, so_n))
#define GI_RD_Y1B(J) GI_RD(yb[1][1][J], GI_PY(1, J, so_n))
    // prologue: stages 0 .. 2 in flight; of stage 0 everything but x1 into the registers
#pragma unroll
    for (int t = 0; t < GI_NSTAGE; ++t) {
#pragma unroll
        for (int b = 0; b < 12; ++b) issue_piece(b);
        issue_done();
    }
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");                    // stage 0 (12 pieces per stage and wave)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    GI_RDX(0, 0); GI_RDX(2, 0); GI_RDYB(0, 0, 0); GI_RDYB(0, 1, 0); GI_RDY2(0);
    // steady state, two k-steps per trip (the y buffers alternate): k-steps 0 .. n64 - 4 fetch the stage three ahead, the last
    // three fetch nothing; n64 is a multiple of 4, so the loop leaves exactly four k-steps
    int t = 0;
#pragma unroll 1
    for (; t + 2 <= n64 - GI_NSTAGE; t += 2) {
        GI_STEP(0, 1, GI_WAIT_VM12, true)
        GI_STEP(1, 0, GI_WAIT_VM12, true)
    }
    GI_STEP(0, 1, GI_WAIT_VM12, true)
    GI_STEP(1, 0, GI_WAIT_VM12, false)
    GI_STEP(0, 1, GI_WAIT_VM0, false)
    GI_STEP(1, 0, GI_WAIT_VM0, false)               // (its reads of "stage n64" fetch a slot nobody writes any more: unused)
    gi_acc_settle();                                // (wait states: see the accumulators' comment above the kernel)
#undef GI_STEP
#undef GI_PROD_IL
#undef GI_PROD_IL1
#undef GI_ROW2A
#undef GI_ROW2B
#undef GI_NONE
#undef GI_PROD
#undef GI_RDX
#undef GI_RDY2
#undef GI_RDYB
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue: the patch arg-min of every (row patch, column frame) of the tile, decided here -- the products never
    // leave the chip (r02 / early r03 wrote them out, 2 GB, for a second kernel to read back).
    // acc = C2 + floor((C3 + floor(C4 / 256)) / 256) in units of 2^-16 of v . v'; d2 = |v_b|^2 - 2 acc 2^-16 in units of 2^-15.
    // D[m][n] of MFMA (j, i): m = column j * 16 + (lane / 16) * 4 + v of this wave's unit, n = row patch (wr * 4 + i) * 16 + lane % 16.
    // A row patch's columns are spread over four lanes and sixteen registers; rather than merge (best, index, runner-up)
    // triples across lanes -- a chain of 64 dependent cross-lane moves per tile, 7 us -- the wave turns its 64 x 64 block
    // of d2 through LDS (the stages are dead by now) so that lane r owns row r: one sequential scan per frame, as the pair
    // kernels of the fp64 form do it.
    __syncthreads();                                                     // every wave is through its last fragment reads
    constexpr int DP = 65;                                               // row pitch (ints): lanes on different rows, same column -> different banks
    int* d2s = (int*)smem_i8 + w * 64 * DP;
    const int quad = lane >> 4;
    const long long unit = (long long)tile_n * 2 + wc;                   // this wave's 64 columns: frames unit * fpu .. + fpu - 1
    {
        int nbl[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const v4i t4 = *(const v4i*)(p.nbp + unit * 64 + j * 16 + quad * 4);
#pragma unroll
            for (int v = 0; v < 4; ++v) nbl[j][v] = t4[v];
        }
#define GI_EPI(I, J)                                                                                            \
        {                                                                                                       \
            const v4i a2 = gi_acc_read<GI_ACC(0, J, I)>(), a3 = gi_acc_read<GI_ACC(1, J, I)>(), a4 = gi_acc_read<GI_ACC(2, J, I)>(); \
            const v4i acc = a2 + ((a3 + (a4 >> 8)) >> 8);                                                       \
            int* dst = d2s + ((I) * 16 + (lane & 15)) * DP + (J) * 16 + quad * 4;                               \
            _Pragma("unroll") for (int v = 0; v < 4; ++v) dst[v] = nbl[J][v] - acc[v];                          \
        }
        GI_EPI(0, 0) GI_EPI(0, 1) GI_EPI(0, 2) GI_EPI(0, 3) GI_EPI(1, 0) GI_EPI(1, 1) GI_EPI(1, 2) GI_EPI(1, 3)
        GI_EPI(2, 0) GI_EPI(2, 1) GI_EPI(2, 2) GI_EPI(2, 3) GI_EPI(3, 0) GI_EPI(3, 1) GI_EPI(3, 2) GI_EPI(3, 3)
#undef GI_EPI
    }
    __builtin_amdgcn_s_waitcnt(0x0070);                                  // lgkmcnt(0): this wave's LDS writes (its own region)
    __builtin_amdgcn_wave_barrier();
    const long long window = dlc_sim_window(p.keys, p.H);
    const unsigned wu = window > 0xffffffffll ? 0xffffffffu : (unsigned)window, pmask = p.P >= 32 ? ~0u : (1u << p.P) - 1u;
    const long long a = m0 + wr * 64 + lane;                             // lane r owns row patch r of the wave's 64
    const int* drow = d2s + lane * DP;
    for (int fs = 0; fs < p.fpu; ++fs) {
        const long long fj = unit * p.fpu + fs;
        if (!(a < p.nrows && fj < p.nframes && a < fj * p.P)) continue;  // (row frame < column frame)  <=>  a < fj * P
        const int* dr = drow + fs * p.P;
        int dv[32];                                                      // P <= 32 (sim_use_filter): all reads in flight at once
#pragma unroll
        for (int b = 0; b < 32; ++b) dv[b] = dr[b < p.P ? b : 0];
#pragma unroll
        for (int b = 1; b < 32; ++b) dv[b] = b < p.P ? dv[b] : 0x7fffffff;
        int best = dv[0], second = 0x7fffffff, bi = 0;
#pragma unroll
        for (int b = 1; b < 32; ++b) {
            second = min(second, max(best, dv[b]));                      // the smaller of the two that are not the new minimum
            bi = dv[b] < best ? b : bi;                                  // strict: the first minimum keeps its index
            best = min(best, dv[b]);
        }
        unsigned cand = 0;
        if ((unsigned)second - (unsigned)best <= wu) {                   // (every dv >= best: the unsigned difference is exact)
#pragma unroll
            for (int b = 0; b < 32; ++b) cand |= ((unsigned)dv[b] - (unsigned)best <= wu ? 1u : 0u) << b;
            cand &= pmask;
            if ((cand & (cand - 1)) == 0) cand = 0;                      // P = 1, or a window wider than the padding's distance
        }
        p.abi[(fj - p.fj_base) * p.rp + a] = (unsigned char)bi;
        p.acand[(fj - p.fj_base) * p.rp + a] = cand;                    // 0 = decided (one patch inside the window)
    }
}

// ---- column extremes, centres, quantisation ------------------------------------------------------------------------
// The dataset's column statistics travel as `range` words (include/dlc.h: dlc_sdav_distinctive_score leaves them when it
// has walked the same descriptors, else sim_colrange_kernel does): [0], [1] reserved, [2] flag (a NaN / infinity seen),
// [3 + k] ordered key of column k's minimum, [3 + H + k] of its maximum.

__global__ void sim_range_init_kernel(unsigned long long* range, int H) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 3) range[i] = 0ull;
    if (i < H) { range[3 + i] = ~0ull; range[3 + H + i] = 0ull; }
}

// Column extremes of x [rows, H]: a workgroup takes 128 columns (a wave's row segment: 64 lanes x 16 bytes = 1 KiB) of one
// chunk of rows, its four waves every fourth row, eight rows in flight per wave; ordered-key atomics fold the chunks.
constexpr int CR_ROWS = 512;                // rows per workgroup
__global__ __launch_bounds__(256) void sim_colrange_kernel(const double* __restrict__ x, long long rows, int H,
                                                           unsigned long long* __restrict__ range) {
    __shared__ double red[4][128][2];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c0 = blockIdx.x * 128 + lane * 2;
    const long long r0 = (long long)blockIdx.y * CR_ROWS, r1 = r0 + CR_ROWS < rows ? r0 + CR_ROWS : rows;
    const bool two = c0 + 1 < H, one = c0 < H;
    const bool vec = (H & 1) == 0 && ((unsigned long long)x & 15) == 0;
    double lo0 = INFINITY, hi0 = -INFINITY, lo1 = INFINITY, hi1 = -INFINITY;
    bool bad = false;
    auto take = [&](double a, double b) {
        bad |= !(fabs(a) < INFINITY) || !(fabs(b) < INFINITY);
        lo0 = fmin(lo0, a); hi0 = fmax(hi0, a); lo1 = fmin(lo1, b); hi1 = fmax(hi1, b);
    };
    if (one) {
        long long r = r0 + w;
        if (vec && two) {
            for (; r + 28 < r1; r += 32) {
                double2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *(const double2*)(x + (r + 4 * u) * H + c0);
#pragma unroll
                for (int u = 0; u < 8; ++u) take(v[u].x, v[u].y);
            }
            for (; r < r1; r += 4) { const double2 v = *(const double2*)(x + r * H + c0); take(v.x, v.y); }
        } else {
            for (; r < r1; r += 4) { const double a = x[r * H + c0]; take(a, two ? x[r * H + c0 + 1] : a); }
        }
    }
    red[w][lane * 2][0] = lo0; red[w][lane * 2][1] = hi0; red[w][lane * 2 + 1][0] = lo1; red[w][lane * 2 + 1][1] = hi1;
    const bool any_bad = __ballot(bad) != 0;
    if (lane == 0 && any_bad) atomicMax(&range[2], 1ull);
    __syncthreads();
    if (threadIdx.x < 128) {
        const int c = blockIdx.x * 128 + threadIdx.x;
        const double lo = fmin(fmin(red[0][threadIdx.x][0], red[1][threadIdx.x][0]), fmin(red[2][threadIdx.x][0], red[3][threadIdx.x][0]));
        const double hi = fmax(fmax(red[0][threadIdx.x][1], red[1][threadIdx.x][1]), fmax(red[2][threadIdx.x][1], red[3][threadIdx.x][1]));
        if (c < H && lo <= hi) { atomicMin(&range[3 + c], dlc_f64_key(lo)); atomicMax(&range[3 + H + c], dlc_f64_key(hi)); }
    }
}

// keys (DLC_SIM_KEYS words) of a call and the columns' centres: cc[k] = the midpoint of column k's extremes, keys[0] = the
// ordered key of the largest column range (0 when every column is constant: all rows the same, every distance 0).
// One workgroup of 256 threads.
__global__ __launch_bounds__(256) void sim_keys_init_kernel(unsigned long long* keys, const unsigned long long* __restrict__ range,
                                                            int H, double* __restrict__ cc) {
    __shared__ double red[256];
    double s = 0.0;
    for (int k = threadIdx.x; k < H; k += 256) {
        const double lo = dlc_f64_unkey(range[3 + k]), hi = dlc_f64_unkey(range[3 + H + k]);
        const bool ok = lo <= hi && fabs(lo) < INFINITY && fabs(hi) < INFINITY;
        cc[k] = ok ? lo + 0.5 * (hi - lo) : 0.0;
        if (ok) s = fmax(s, hi - lo);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x < DLC_SIM_KEYS) keys[threadIdx.x] = 0ull;
    __syncthreads();
    if (threadIdx.x == 0) {
        // (a range that overflows, two finite extremes 1e308 apart, cannot be scaled: the fp64 form takes such data)
        keys[0] = dlc_f64_key(red[0] < INFINITY ? red[0] : 0.0);
        keys[2] = (range[2] || !(red[0] < INFINITY)) ? 1ull : 0ull;
    }
}

// v = (x - c) * inv clamped to the digits' range (values outside it exist only in streams, whose range is fixed at
// creation, and are flagged: `outside`), and its 24-bit fixed-point value
constexpr double SIM_VMAX = 0.498;
__device__ __forceinline__ double sim_centred(double x, double c, double inv, bool& outside) {
    const double v = (x - c) * inv;
    outside |= !(fabs(v) <= SIM_VMAX * (1.0 + 0x1p-40));
    return v > -SIM_VMAX ? (v < SIM_VMAX ? v : SIM_VMAX) : -SIM_VMAX;       // (a NaN lands on -SIM_VMAX, flagged)
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

// The projections p = dot(score, row) are kept as double-double values (hi, lo): what the similarity takes from them is the
// DIFFERENCE of two rows' projections, |dot(score, m_a - m_b)| (SimilarityCalculator.py:42-43), and for rows that lie close
// together -- low-contrast descriptors, every column within 1e-3 of its mean: a quarter of the matched pairs of real
// frames through 1/sqrt(fan_in) weights -- the difference of two fp64 sums of magnitude 600 had lost seven of its digits,
// so the pair kernels re-evaluated it from the rows (one wave, 60 KB, per matched pair: 4 M of them, 20 of the call's
// 33 ms).  With 106 bits in each projection the difference is good to 1e-20 of the projections whatever the rows are.
struct dd_t { double hi, lo; };
__device__ __forceinline__ void dd_add_prod(dd_t& s, double a, double b) {       // s += a * b (a * b exactly: fma)
#pragma clang fp contract(off)
    const double p = a * b;
    const double e = fma(a, b, -p);
    const double t = s.hi + p;
    const double bb = t - s.hi;
    const double err = (s.hi - (t - bb)) + (p - bb);
    s.hi = t;
    s.lo += err + e;
}
__device__ __forceinline__ dd_t dd_add(dd_t x, dd_t y) {
#pragma clang fp contract(off)
    const double t = x.hi + y.hi;
    const double bb = t - x.hi;
    const double err = (x.hi - (t - bb)) + (y.hi - bb);
    dd_t r;
    r.hi = t;
    r.lo = (x.lo + y.lo) + err;
    return r;
}

// The pass over the descriptors that both forms of the similarity share, a workgroup per group of 16 patch rows, a wave
// per k-step of 64 elements (w, w + 4, ..), lane l on row l % 16, elements ks * 64 + (l / 16) * 16 .. + 15 -- 128
// contiguous bytes per lane:
//   |x|^2 (the fp64 Gram form's norms) and p = dot(score, row) (double-double: proj[2 r], proj[2 r + 1]) for every row --
//   per lane in k order, then the row's four lanes (xor 16, 32), then its four waves in order: one fixed summation order
//   for both forms, whose projections must agree bit for bit;
//   QUANT (the filter): sum |v| (its largest value into keys[3], an ordered key), |v|^2, and the three signed digits of
//   the 24-bit fixed-point value, 16 bytes of each per lane at lane * 16 of that (group, slice, k-step) block -- 1 KiB per
//   store instruction.  Rows past the last one and k >= H are zeros there and take no part in the sums.
constexpr int SR_WAVES = 8;                // waves of a workgroup: the k-steps of a group's 16 rows are dealt round them
template <bool QUANT>
__global__ __launch_bounds__(64 * SR_WAVES) void sim_rows_kernel(const double* __restrict__ desc, long long rows, int H, int kp,
                                                       const double* __restrict__ score, unsigned long long* keys,
                                                       const double* __restrict__ cc,
                                                       char* __restrict__ X, double* __restrict__ nrm2,
                                                       double* __restrict__ nu2, double* __restrict__ proj,
                                                       unsigned long long* __restrict__ rowhash, long long g0, int P, int fpu,
                                                       int* __restrict__ nbp) {
    __shared__ double red[SR_WAVES][16][5];
    __shared__ unsigned long long redh[SR_WAVES][16][2];
    unsigned long long h1 = 0, h2 = 0;
    const long long g = g0 + blockIdx.x;          // (g0 > 0: the groups a stream's new frames touch)
    bool outside = false;                         // a value outside the digits' range: only possible with a FIXED range (streams)
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, rr = lane & 15, chunk = lane >> 4;
    double inv = 0.0;
    if constexpr (QUANT) {
        const double s_ = dlc_f64_unkey(keys[0]);
        inv = s_ > 0.0 ? 2.0 * SIM_VMAX / s_ : 0.0;
    }
    const long long r = g * 16 + rr;
    const bool row_ok = r < rows;
    const int nks = QUANT ? kp / 64 : (H + 63) / 64;
    const double* x = desc + (row_ok ? r : 0) * H;
    const bool vec = (H & 1) == 0 && ((unsigned long long)desc & 15) == 0 && ((unsigned long long)score & 15) == 0 &&
                     (!QUANT || ((unsigned long long)cc & 15) == 0);
    char* xg = X + g * (3ll * nks * 1024) + lane * 16;
    double n2 = 0.0, su = 0.0, s2 = 0.0;
    dd_t pr = {0.0, 0.0};
    for (int ks = w; ks < nks; ks += SR_WAVES) {
        const int k0 = ks * 64 + chunk * 16;
        unsigned w1[4] = {0, 0, 0, 0}, w2[4] = {0, 0, 0, 0}, w3[4] = {0, 0, 0, 0};
        auto put = [&](int e, double v, double sc, double c) {
            n2 = fma(v, v, n2);
            dd_add_prod(pr, sc, v);
            if (rowhash) {
                const unsigned long long bits = (unsigned long long)__double_as_longlong(v), pos = (unsigned long long)(k0 + e);
                h1 += sim_mix64(bits + pos * 0x9e3779b97f4a7c15ull);
                h2 += sim_mix64((bits ^ 0xd6e8feb86659fd93ull) + pos * 0xc2b2ae3d27d4eb4full);
            }
            if constexpr (QUANT) {
                const double u = sim_centred(v, c, inv, outside);
                su += fabs(u); s2 = fma(u, u, s2);
                const int q = (int)rint(u * 16777216.0);                // |q| <= 0.498 * 2^24 = 8 355 054: digits in [-128, 127]
                const int q1 = (q + 128) >> 8;                           // (q - s3) / 256, s3 = the signed low byte of q
                w1[e >> 2] |= (unsigned)(((q1 + 128) >> 8) & 255) << (8 * (e & 3));
                w2[e >> 2] |= (unsigned)(q1 & 255) << (8 * (e & 3));
                w3[e >> 2] |= (unsigned)(q & 255) << (8 * (e & 3));
            }
        };
        if (row_ok) {
            if (vec && k0 + 16 <= H) {                                  // 128 contiguous, 16-byte aligned bytes per lane
                double2 v[8], c[8], m[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[e] = *(const double2*)(x + k0 + 2 * e); c[e] = *(const double2*)(score + k0 + 2 * e);
                    if constexpr (QUANT) m[e] = *(const double2*)(cc + k0 + 2 * e); else m[e] = make_double2(0.0, 0.0);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) { put(2 * e, v[e].x, c[e].x, m[e].x); put(2 * e + 1, v[e].y, c[e].y, m[e].y); }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (k0 + e < H) put(e, x[k0 + e], score[k0 + e], QUANT ? cc[k0 + e] : 0.0);
            }
        }
        if constexpr (QUANT) {
            const uint4 v1 = make_uint4(w1[0], w1[1], w1[2], w1[3]), v2 = make_uint4(w2[0], w2[1], w2[2], w2[3]),
                        v3 = make_uint4(w3[0], w3[1], w3[2], w3[3]);
            *(uint4*)(xg + (0ll * nks + ks) * 1024) = v1;
            *(uint4*)(xg + (1ll * nks + ks) * 1024) = v2;
            *(uint4*)(xg + (2ll * nks + ks) * 1024) = v3;
        }
    }
    for (int o = 16; o <= 32; o <<= 1) {
        n2 += __shfl_xor(n2, o);
        dd_t other;
        other.hi = __shfl_xor(pr.hi, o); other.lo = __shfl_xor(pr.lo, o);
        pr = dd_add(pr, other);
        if constexpr (QUANT) { su += __shfl_xor(su, o); s2 += __shfl_xor(s2, o); }
        h1 += __shfl_xor(h1, o); h2 += __shfl_xor(h2, o);
    }
    if constexpr (QUANT) {
        if (__ballot(outside) != 0 && lane == 0) atomicOr(&keys[2], 1ull);
    }
    if (chunk == 0) {
        red[w][rr][0] = n2; red[w][rr][1] = pr.hi; red[w][rr][2] = su; red[w][rr][3] = s2; red[w][rr][4] = pr.lo;
        redh[w][rr][0] = h1; redh[w][rr][1] = h2;
    }
    __syncthreads();
    if (w == 0 && chunk == 0 && row_ok) {
        double t[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            t[c] = red[0][rr][c];
#pragma unroll
            for (int ww = 1; ww < SR_WAVES; ++ww) t[c] += red[ww][rr][c];
        }
        if (nrm2) nrm2[r] = t[0];
        dd_t pt = {red[0][rr][1], red[0][rr][4]};
#pragma unroll
        for (int ww = 1; ww < SR_WAVES; ++ww) { const dd_t o = {red[ww][rr][1], red[ww][rr][4]}; pt = dd_add(pt, o); }
        {                                                                // normalised: hi = the sum rounded to fp64, lo = the rest
#pragma clang fp contract(off)
            const double hi = pt.hi + pt.lo;
            proj[2 * r] = hi;
            proj[2 * r + 1] = pt.lo - (hi - pt.hi);
        }
        if constexpr (QUANT) {
            nu2[r] = t[3];
            atomicMax(&keys[3], dlc_f64_key(t[2]));
            if (nbp) { const long long f = r / P; nbp[(f / fpu) * 64 + (f % fpu) * P + (r - f * P)] = (int)llrint(t[3] * 32768.0); }
        }
        if (rowhash) {
            unsigned long long ha = 0, hb = 0;
#pragma unroll
            for (int ww = 0; ww < SR_WAVES; ++ww) { ha += redh[ww][rr][0]; hb += redh[ww][rr][1]; }
            rowhash[2 * r] = ha;
            rowhash[2 * r + 1] = hb;
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

// rows of the row panel: whole tiles and a tile of slack
static int64_t sim_panel_rows(int64_t rows) { return (int64_t)dlc::align_up((size_t)rows, (size_t)GI_T) + GI_T + 16; }

size_t sim_filter_panel_bytes(int64_t rows, int64_t H) {
    const size_t kp = dlc::align_up((size_t)H, (size_t)GI_KPAD);
    return (size_t)sim_panel_rows(rows) * 3 * kp;
}

// The column side of the products: units of 64 columns = fpu = 64 / P whole frames + zero rows; whole tiles of two units.
int sim_frames_per_unit(int64_t P) { return (int)(64 / P); }
static int64_t sim_col_tiles(int64_t N, int64_t P) { return dlc::cdiv(dlc::cdiv(N, (int64_t)sim_frames_per_unit(P)), (int64_t)2); }
int64_t sim_col_rows(int64_t N, int64_t P) { return sim_col_tiles(N, P) * GI_T; }                      // columns in unit layout: entries of nbp
int64_t sim_col_frames(int64_t N, int64_t P) { return sim_col_tiles(N, P) * 2 * sim_frames_per_unit(P); }
int64_t sim_argmin_pitch(int64_t N, int64_t P) { return (int64_t)dlc::align_up((size_t)(N * P), 64); }   // nfp: frames per row of abi / acand
// the block table of gram_argmin_i8 (`blocks`)
size_t gram_blocks_bytes(int64_t N, int64_t P) {
    const size_t tiles_m = (size_t)dlc::cdiv((N > 1 ? N - 1 : 1) * P, (int64_t)GI_T), tiles_n = (size_t)sim_col_tiles(N, P);
    const size_t nsm = (tiles_m + GI_BR - 1) / GI_BR, nsn = (tiles_n + GI_BC - 1) / GI_BC;
    return dlc::align_up(nsm + 1, 2) * 4 + nsm * nsn * 8;
}
// the product kernel's column lanes address the panel with 32-bit offsets
bool sim_filter_fits(int64_t N, int64_t P, int64_t H) { return sim_filter_panel_bytes(N * P, H) < (1ull << 32); }

// keys (DLC_SIM_KEYS words): [0] ordered key of the largest column range, [2] flag (1: a NaN / infinity or a value outside a
// stream's fixed range, 2: the sample said the filter would not decide -- match_ref.hip), [3] ordered key of the largest
// row sum of |v|, [4] direct evaluations (a count), [5] length of prog (up to 1023 int2 entries for H <= 32768), [6] / [7]
// the sample's undecided / finished cells.  cc: H doubles (16-byte aligned), ws_range: sim_range_words(H) words for the
// column extremes when the caller brings none.  X: sim_filter_panel_bytes (its rows behind the last patch are zeros);
// nbp: sim_col_rows ints.
size_t sim_range_words(int64_t H) { return 3 + 2 * (size_t)H; }

static int sim_column_extremes(dlc_ctx* ctx, const double* desc, int64_t rows, int64_t H, unsigned long long* range, hipStream_t st) {
    hipLaunchKernelGGL(sim_range_init_kernel, dim3((unsigned)dlc::cdiv(H > 3 ? H : 3, (int64_t)256)), dim3(256), 0, st, range, (int)H);
    hipLaunchKernelGGL(sim_colrange_kernel, dim3((unsigned)dlc::cdiv(H, (int64_t)128), (unsigned)dlc::cdiv(rows, (int64_t)CR_ROWS)), dim3(256),
                       0, st, desc, (long long)rows, (int)H, range);
    DLC_LAUNCH_CHECK(ctx, "sim_colrange_kernel");
    return DLC_OK;
}

int sim_filter_prepare(dlc_ctx* ctx, const double* desc, int64_t N, int64_t P, int64_t H, const double* score,
                       unsigned long long* keys, double* cc, unsigned long long* ws_range, char* X, int* nbp, double* nu2,
                       double* proj, unsigned long long* rowhash, void* prog, const unsigned long long* range, hipStream_t st) {
    const int64_t rows = N * P;
    const int kp = (int)dlc::align_up((size_t)H, (size_t)GI_KPAD);
    if (!range) {
        const int rc = sim_column_extremes(ctx, desc, rows, H, ws_range, st);
        if (rc != DLC_OK) return rc;
        range = ws_range;
    }
    hipLaunchKernelGGL(sim_keys_init_kernel, dim3(1), dim3(256), 0, st, keys, range, (int)H, cc);
    hipLaunchKernelGGL(sim_pairwise_program_kernel, dim3(1), dim3(64), 0, st, (int)H, (int2*)prog, keys + 5);
    DLC_LAUNCH_CHECK(ctx, "sim_keys_init_kernel");
    DLC_HIP_CHECK(ctx, hipMemsetAsync(nbp, 0, (size_t)sim_col_rows(N, P) * 4, st));
    hipLaunchKernelGGL(sim_rows_kernel<true>, dim3((unsigned)(sim_panel_rows(rows) / 16)), dim3(64 * SR_WAVES), 0, st, desc, (long long)rows,
                       (int)H, kp, score, keys, (const double*)cc, X, (double*)nullptr, nu2, proj, rowhash, 0ll, (int)P,
                       sim_frames_per_unit(P), nbp);
    DLC_LAUNCH_CHECK(ctx, "sim_rows_kernel");
    return DLC_OK;
}

// ---- the streaming form (match_ref.hip: dlc_sdav_stream_*): a resident, append-only panel over a range FIXED at creation:
// x - centre[k] in [lo, hi] for every column k (centre: DEVICE, H doubles, or null for zeros)
__global__ __launch_bounds__(256) void sim_stream_keys_kernel(unsigned long long* keys, double lo, double hi,
                                                              const double* __restrict__ centre, int H, double* __restrict__ cc) {
    if (threadIdx.x < DLC_SIM_KEYS) keys[threadIdx.x] = 0ull;
    __syncthreads();
    if (threadIdx.x == 0) keys[0] = dlc_f64_key(hi - lo);
    for (int k = threadIdx.x; k < H; k += 256) cc[k] = (centre ? centre[k] : 0.0) + (lo + 0.5 * (hi - lo));
}

int sim_stream_init(dlc_ctx* ctx, unsigned long long* keys, double* cc, void* prog, int64_t H, double lo, double hi,
                    const double* centre, hipStream_t st) {
    hipLaunchKernelGGL(sim_stream_keys_kernel, dim3(1), dim3(256), 0, st, keys, lo, hi, centre, (int)H, cc);
    hipLaunchKernelGGL(sim_pairwise_program_kernel, dim3(1), dim3(64), 0, st, (int)H, (int2*)prog, keys + 5);
    DLC_LAUNCH_CHECK(ctx, "sim_stream_keys_kernel");
    return DLC_OK;
}

// quantise the 16-row groups g_first .. g_first + g_count - 1 of desc[rows_total, H] into the panel and the per-row arrays
// (a group shared with older rows is rewritten with the same values)
int sim_stream_quantise(dlc_ctx* ctx, const double* desc, int64_t rows_total, int64_t H, const double* score,
                        unsigned long long* keys, const double* cc, char* X, double* nu2, double* proj,
                        unsigned long long* rowhash, int64_t g_first, int64_t g_count, int64_t P, int* nbp, hipStream_t st) {
    const int kp = (int)dlc::align_up((size_t)H, (size_t)GI_KPAD);
    if (g_count < 1) return DLC_OK;
    // nbp: |v|^2 of every patch in the product kernel's unit layout (the batched query's strip takes the patches as columns)
    hipLaunchKernelGGL(sim_rows_kernel<true>, dim3((unsigned)g_count), dim3(64 * SR_WAVES), 0, st, desc, (long long)rows_total, (int)H, kp, score,
                       keys, cc, X, (double*)nullptr, nu2, proj, rowhash, (long long)g_first, (int)P, sim_frames_per_unit(P), nbp);
    DLC_LAUNCH_CHECK(ctx, "sim_rows_kernel");
    return DLC_OK;
}

size_t sim_stream_panel_bytes(int64_t rows, int64_t H) {
    const size_t kp = dlc::align_up((size_t)H, (size_t)GI_KPAD);
    return (size_t)(dlc::cdiv(rows, (int64_t)16) + 8) * 3 * kp * 16;      // whole groups + the queries' five-group window past the end
}

// |x|^2, dot(score, x) and the content hash of every patch row (the fp64 Gram form; the filter's prepare computes the same
// projections and hashes), and NumPy's pairwise-summation program for rows of H elements (prog: sim_pairwise_program_bytes,
// its length to *prog_len) -- what the pair kernels need to decide the arg-mins the Gram matrix cannot.
size_t sim_pairwise_program_bytes(int64_t H) { return dlc::align_up((size_t)(H / 32 + 64) * 8, 256); }

int sim_row_sums(dlc_ctx* ctx, const double* desc, int64_t rows, int64_t H, const double* score, double* nrm2, double* proj,
                 unsigned long long* rowhash, void* prog, unsigned long long* prog_len, hipStream_t st) {
    hipLaunchKernelGGL(sim_pairwise_program_kernel, dim3(1), dim3(64), 0, st, (int)H, (int2*)prog, prog_len);
    hipLaunchKernelGGL(sim_rows_kernel<false>, dim3((unsigned)dlc::cdiv(rows, (int64_t)16)), dim3(64 * SR_WAVES), 0, st, desc, (long long)rows,
                       (int)H, 0, score, (unsigned long long*)nullptr, (const double*)nullptr, (char*)nullptr, nrm2, (double*)nullptr, proj,
                       rowhash, 0ll, 1, 1, (int*)nullptr);
    DLC_LAUNCH_CHECK(ctx, "sim_rows_kernel");
    return DLC_OK;
}

// The patch arg-min of every (row patch a, column frame j) with frame(a) < j, for all N frames at once: abi / acand are
// [sim_col_frames(N, P), sim_argmin_pitch(N, P)] (bytes / 32-bit words); entries with frame(a) >= j are not written.
int gram_argmin_i8(dlc_ctx* ctx, int64_t N, int64_t P, int64_t H, const char* X, const int* nbp,
                   const unsigned long long* keys, unsigned char* abi, unsigned* acand, void* blocks, hipStream_t st) {
    const int kp = (int)dlc::align_up((size_t)H, (size_t)GI_KPAD);
    GramI8Args a;
    a.gpitch = 3ll * kp * 16; a.kp = kp; a.H = (int)H; a.P = (int)P; a.fpu = sim_frames_per_unit(P);
    a.X = X; a.zrow = (int64_t)dlc::align_up((size_t)(N * P), 16); a.nbp = nbp; a.keys = keys; a.abi = abi; a.acand = acand;
    a.nfp = sim_col_frames(N, P); a.rp = sim_argmin_pitch(N, P); a.nrows = N * P; a.nframes = N;
    a.tn_lo = 0; a.tn_hi = 0x7fffffff; a.fj_base = 0; a.strip_cols = 0; a.sj_lo = 0;
    a.tiles_m = (int)dlc::cdiv((N - 1) * P, (int64_t)GI_T);              // the last frame's patches have no later frame
    a.tiles_n = (int)sim_col_tiles(N, P);
    if (a.tiles_m < 1) return DLC_OK;
    a.nsm = (a.tiles_m + GI_BR - 1) / GI_BR;
    a.nsn = (a.tiles_n + GI_BC - 1) / GI_BC;
    a.nsup = 0;                               // blocks with a wanted tile (gram_blocks_kernel numbers them the same way)
    for (int si = 0; si < a.nsm; ++si) {
        const int cnt = a.nsn - gi_first_block_col(si, (int)P, a.fpu);
        if (cnt > 0) a.nsup += cnt;
    }
    if (a.nsup == 0) return DLC_OK;
    int* rowstart = (int*)blocks;
    a.blk = (const int2*)(rowstart + dlc::align_up((size_t)a.nsm + 1, 2));
    hipLaunchKernelGGL(gram_blocks_kernel, dim3(1), dim3(256), 0, st, a.nsm, a.nsn, (int)P, a.fpu, rowstart, (int2*)a.blk);
    DLC_LAUNCH_CHECK(ctx, "gram_blocks_kernel");
    const size_t lds = (size_t)GI_NSTAGE * GI_STAGE;
    if (!(ctx->func_attr_set & (1ull << DLC_ATTR_GRAM_I8))) {
        DLC_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)gram_i8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ctx->func_attr_set |= 1ull << DLC_ATTR_GRAM_I8;
    }
    const unsigned grid = (unsigned)(((a.nsup + 7) / 8) * 8 * 32);
    // bench.py's kernel-only timing (dlc_set_profiling): an event pair around the kernel on its stream
    const int prof_slot = (int)(ctx->prof_calls % DLC_PROFILE_RING);
    if (ctx->profiling) DLC_HIP_CHECK(ctx, hipEventRecord(ctx->ev_start[prof_slot], st));
    hipLaunchKernelGGL(gram_i8_kernel, dim3(grid), dim3(256), lds, st, a);
    DLC_LAUNCH_CHECK(ctx, "gram_i8_kernel");
    if (ctx->profiling) {
        DLC_HIP_CHECK(ctx, hipEventRecord(ctx->ev_stop[prof_slot], st));
        ctx->prof_calls++;
    }
    return DLC_OK;
}

// ---- a strip of the triangle: the column frames f_first .. f_last against every older row patch ----------------------
// (the streaming form's batches: the frames of a batch are the strip's columns, the resident panel its rows -- the SAME
// kernel, so a batch of 32 frames costs its share of the matrix call instead of 16 passes over the panel.)
// frames of the strip's abi / acand (whole block columns)
int64_t gram_strip_frames(int64_t f_first, int64_t f_last, int64_t P) {
    const int64_t fpb = (int64_t)GI_BC * 2 * sim_frames_per_unit(P);     // frames per block column
    return (f_last / fpb - f_first / fpb + 1) * fpb;
}

// X: a panel whose rows are frame-major patches (frame f's patch p = row f P + p) with an all-zero group at row `zrow`,
// nbp in unit layout over the same frame numbering.  abi / acand: [gram_strip_frames, rp]; the entry of (column frame fj,
// row patch a) is row fj - *fj_base_out.  Written: f_first <= fj <= f_last (and the other frames of their tiles), a < fj P.
int gram_argmin_i8_strip(dlc_ctx* ctx, int64_t f_first, int64_t f_last, int64_t P, int64_t H, const char* X, int64_t zrow,
                         const int* nbp, const unsigned long long* keys, unsigned char* abi, unsigned* acand, int64_t rp,
                         int64_t* fj_base_out, hipStream_t st, bool launch) {
    const int kp = (int)dlc::align_up((size_t)H, (size_t)GI_KPAD);
    const int64_t N = f_last + 1;
    GramI8Args a;
    a.gpitch = 3ll * kp * 16; a.kp = kp; a.H = (int)H; a.P = (int)P; a.fpu = sim_frames_per_unit(P);
    a.X = X; a.zrow = zrow; a.nbp = nbp; a.keys = keys; a.abi = abi; a.acand = acand;
    a.nfp = 0; a.rp = rp; a.nrows = N * P; a.nframes = N;
    a.tiles_m = (int)dlc::cdiv(f_last * P, (int64_t)GI_T);                // rows of the frames older than the newest
    a.tiles_n = (int)sim_col_tiles(N, P);
    a.tn_lo = (int)(f_first / a.fpu / 2); a.tn_hi = (int)(f_last / a.fpu / 2);
    const int sj_lo = a.tn_lo / GI_BC, sj_hi = a.tn_hi / GI_BC;
    a.fj_base = (int64_t)sj_lo * GI_BC * 2 * a.fpu;
    *fj_base_out = a.fj_base;
    if (a.tiles_m < 1 || !launch) return DLC_OK;                          // (!launch: the caller only wants the strip's base)
    a.nsm = (a.tiles_m + GI_BR - 1) / GI_BR;
    a.nsn = (a.tiles_n + GI_BC - 1) / GI_BC;
    a.nsup = a.nsm * (sj_hi - sj_lo + 1);
    a.blk = nullptr; a.strip_cols = sj_hi - sj_lo + 1; a.sj_lo = sj_lo;
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
