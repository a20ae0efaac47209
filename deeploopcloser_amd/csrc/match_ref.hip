// Reference-semantics match kernels:
//   * cnn_vtl distance matrix: sum_k popcount(|a_k ^ b_k|) on signed int8
//     (src/cnn_vtl/similarity/DistanceCalculator.py:4-12, loop of
//     src/cnn_vtl/create_distance_matrix.py:30-36).  HBM/VALU-bound byte work:
//     LDS-tiled 64x64 output blocks, 4 bytes per VALU op (SWAR |x|, v_bcnt).
//   * SDAV similarity matrix (src/sdav/similarity/SimilarityCalculator.py:12-49,
//     loop of src/sdav/create_similarity_matrix.py:29-38): patch-to-patch
//     distances from one fp64 MFMA Gram GEMM (||a||^2+||b||^2-2a.b), then one
//     wave per frame pair for argmin / weighted distance / log-sum; or, for up to 32 patches per frame, the arg-min
//     from exact integer products of column-centred 24-bit fixed-point descriptors with a direct fp64 evaluation wherever their
//     error bound cannot separate the candidates (gram_i8.hip).
#include <cstdlib>
#include <cstring>
#include "gemm_internal.h"

namespace dlc_gemm {
int gram_upper_f64(dlc_ctx* ctx, int blayout, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda,
                   const double* B, int64_t ldb, double* C, int64_t ldc, int patches, int64_t row0, int64_t col0,
                   hipStream_t st);
}

namespace {

// ---------------------------------------------------------------------------
// cnn_vtl distance
// ---------------------------------------------------------------------------
// popcount(|x|) for the four signed bytes of a word w:  |x| = (x ^ m) + s with m = 0xFF and s = 1 for negative bytes
// (~x <= 127, so the +1 never carries out of a byte; -128 -> 128 -> 1 bit, as bin(-128) has):
//     s = (w >> 7) & 0x01010101,  m = (s << 8) - s,  popcount((w ^ m) + s).
// The kernel below stages x ^ m(x) and s(x) per OPERAND word (see there).

constexpr int DT = 64;      // output tile (frames x frames)
constexpr int DCH = 64;     // descriptor bytes per step (16 words)

// popcount(|a ^ b|) is symmetric, so only tiles on or above the diagonal are computed and every off-diagonal
// tile is stored twice (the reference's loop evaluates both orders, create_distance_matrix.py:33-36: same
// integers).  The descriptor width is cut into gridDim.z chunks -- 153 tile pairs alone would leave a third of
// the chip idle with one wave per SIMD -- whose partial sums are added with 64-bit integer atomics onto a
// zeroed matrix: integer addition, so the result does not depend on the order.  WORDS: rows are 4-byte
// aligned (base and ldd), loaded a word at a time; otherwise byte by byte.
template <bool WORDS>
__global__ __launch_bounds__(256) void distance_matrix_kernel(const int8_t* __restrict__ desc, long long n, long long d,
                                                              long long ldd, long long kchunk,
                                                              unsigned long long* __restrict__ out) {
    if (blockIdx.x < blockIdx.y) return;                  // below the diagonal: the mirror of another tile
    // staged per operand word, once: x' = x ^ m(x) and s(x) (the mask and the carry-in above).  For w = a ^ b the sign bytes are
    // s(w) = s(a) ^ s(b) and m(w) = m(a) ^ m(b), so |w| = (a' ^ b') + (s(a) ^ s(b)): an xor, a v_xad_u32 and the counting add
    // per word pair instead of six instructions (the kernel is VALU-issue-bound: 1.15 G word pairs at 1063 frames).
    __shared__ unsigned As[DT][DCH / 4 + 1];
    __shared__ unsigned Bs[DT][DCH / 4 + 1];
    __shared__ unsigned Asg[DT][DCH / 4 + 1];
    __shared__ unsigned Bsg[DT][DCH / 4 + 1];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const long long i0 = (long long)blockIdx.y * DT, j0 = (long long)blockIdx.x * DT;
    const long long k_lo = (long long)blockIdx.z * kchunk;
    const long long k_hi = k_lo + kchunk < d ? k_lo + kchunk : d;
    int acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0;
    const int lrow = tid >> 2, lw0 = (tid & 3) * 4;   // 4 threads per row, 4 words each
    const bool a_ok = i0 + lrow < n, b_ok = j0 + lrow < n;
    const int8_t* arow = desc + (a_ok ? i0 + lrow : 0) * ldd;
    const int8_t* brow = desc + (b_ok ? j0 + lrow : 0) * ldd;
    // a step's words are fetched into registers while the step before is being counted (a workgroup has only ~13 steps
    // at 1063 frames and two or three workgroups share a CU: the fetch latency was in the open)
    unsigned va[4], vb[4];
    auto fetch = [&](long long k0) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            va[w] = 0; vb[w] = 0;
            const long long k = k0 + (lw0 + w) * 4;
            if constexpr (WORDS) {
                if (k < k_hi) {
                    const long long left = k_hi - k;
                    const unsigned mask = left >= 4 ? 0xffffffffu : ((1u << (8 * (int)left)) - 1u);
                    if (a_ok) va[w] = *(const unsigned*)(arow + k) & mask;
                    if (b_ok) vb[w] = *(const unsigned*)(brow + k) & mask;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (k + e < k_hi) {
                        if (a_ok) va[w] |= ((unsigned)(unsigned char)arow[k + e]) << (8 * e);
                        if (b_ok) vb[w] |= ((unsigned)(unsigned char)brow[k + e]) << (8 * e);
                    }
                }
            }
        }
    };
    fetch(k_lo);
    for (long long k0 = k_lo; k0 < k_hi; k0 += DCH) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const unsigned sa = (va[w] >> 7) & 0x01010101u, sb = (vb[w] >> 7) & 0x01010101u;
            As[lrow][lw0 + w] = va[w] ^ ((sa << 8) - sa);
            Bs[lrow][lw0 + w] = vb[w] ^ ((sb << 8) - sb);
            Asg[lrow][lw0 + w] = sa;
            Bsg[lrow][lw0 + w] = sb;
        }
        __syncthreads();
        if (k0 + DCH < k_hi) fetch(k0 + DCH);
#pragma unroll
        for (int w = 0; w < DCH / 4; ++w) {
            unsigned a[4], b[4], sa[4], sb[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a[r] = As[ty * 4 + r][w]; b[r] = Bs[tx * 4 + r][w];
                sa[r] = Asg[ty * 4 + r][w]; sb[r] = Bsg[tx * 4 + r][w];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r][c] += __popc((a[r] ^ b[c]) + (sa[r] ^ sb[c]));
        }
        __syncthreads();
    }
    const bool diag = blockIdx.x == blockIdx.y;           // a diagonal tile holds both orders itself
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const long long i = i0 + ty * 4 + r, j = j0 + tx * 4 + c;
            if (i < n && j < n) {
                atomicAdd(&out[i * n + j], (unsigned long long)acc[r][c]);
                if (!diag) atomicAdd(&out[j * n + i], (unsigned long long)acc[r][c]);
            }
        }
}

// ---------------------------------------------------------------------------
// SDAV similarity
// ---------------------------------------------------------------------------
// Column mean of desc viewed as [rows, H], summed row by row in order (what
// np.average(axis=0) does on a C-contiguous array, SimilarityCalculator.py:20-23),
// then the distinctive score exp(-(avg-mu)^2 / (2 sigma^2)) (:25-27).
//
// The row-ordered fp64 add chain of a column IS the specification (bit parity with NumPy), the
// serial LOAD chain is not: a workgroup owns DS_COLS = 8 columns (one 64-byte segment of every
// row) and all of its 256 threads fetch -- 32 lane groups x DS_U rows each = a batch of 512 rows =
// 32 KiB, and TWO batches ahead (two register sets): 64 KiB in flight per workgroup, 313 workgroups
// at H = 2500 (16 columns per workgroup were 157 -- fewer than the chip has CUs: 0.38 ms; 8: 0.33; 4 columns
// or smaller batches: slower; measured r04, scripts/exp/ds_variants.sh) -- while the first 8 lanes add
// the batch that has landed out of LDS in row order.  One
// barrier per batch (double-buffered LDS).  With one batch ahead a batch took as long as its loads
// (3.2 us, the add chain 1 us): 2 TB/s.  (512 threads and one batch of 64 KiB ahead: slower still --
// the same latency per batch, twice the batch.)  HBM-bound (rows*H*8 bytes read once) down to the
// latency of the add chain itself (rows x one v_add_f64).
#ifndef DLC_DS_COLS
#define DLC_DS_COLS 8
#define DLC_DS_U 16
#endif
constexpr int DS_COLS = DLC_DS_COLS;      // columns per workgroup
constexpr int DS_U = DLC_DS_U;            // rows per thread and batch
constexpr int DS_G = 256 / DS_COLS;       // lane groups: rows fetched side by side
constexpr int DS_ROWS = DS_G * DS_U;      // rows per batch
// range (may be null): the pass sees every element once, so it also leaves what the similarity's filter form needs to know
// about the dataset -- every COLUMN's minimum and maximum as ordered keys, and whether it holds a NaN / infinity: the layout
// of gram_i8.hip's sim_colrange_kernel, 3 + 2 H words -- and dlc_sdav_similarity_matrix then skips its own pass over the
// 638 MB.  A workgroup owns its columns, so the extremes are plain stores; only the flag is shared (range_init_kernel).
__global__ void range_init_kernel(unsigned long long* range) {
    if (threadIdx.x < 3) range[threadIdx.x] = 0ull;
}
__global__ __launch_bounds__(256) void distinctive_score_kernel(const double* __restrict__ desc, long long rows, int H,
                                                                double mu, double sigma, double* __restrict__ score,
                                                                unsigned long long* __restrict__ range) {
    __shared__ double buf[2][DS_ROWS][DS_COLS];
    double lo = INFINITY, hi = -INFINITY;
    bool bad = false;
    const int tid = threadIdx.x, c = tid % DS_COLS, g = tid / DS_COLS;
    // workgroup ids go round-robin to the 8 XCDs: XCD x takes the x-th CONTIGUOUS eighth of the column groups, so that the
    // groups that share a cache line (and a DRAM page) run side by side under one L2 instead of on eight of them
    const int groups = (H + DS_COLS - 1) / DS_COLS, per = (groups + 7) / 8;
    const int grp = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (grp >= groups) return;
    const int col = grp * DS_COLS + c;
    const bool col_ok = col < H;
    const double* src = desc + (col_ok ? col : 0);
    double va[DS_U], vb[DS_U];
    auto fetch = [&](double (&v)[DS_U], long long r0) {
#pragma unroll
        for (int u = 0; u < DS_U; ++u) {
            const long long r = r0 + u * DS_G + g;
            v[u] = (col_ok && r < rows) ? src[r * H] : 0.0;
        }
    };
    double s = 0.0;
    // one batch: its values (landed by now) into LDS, the fetch of the batch two ahead into the same registers, the add
    auto batch = [&](double (&v)[DS_U], int b, long long r0) {
#pragma unroll
        for (int u = 0; u < DS_U; ++u) buf[b][u * DS_G + g][c] = v[u];
        if (range) {                                           // (here, where the loads have landed -- not where they are issued)
#pragma unroll
            for (int u = 0; u < DS_U; ++u)
                if (col_ok && r0 + u * DS_G + g < rows) { bad |= !(fabs(v[u]) < INFINITY); lo = fmin(lo, v[u]); hi = fmax(hi, v[u]); }
        }
        __syncthreads();
        if (r0 + 2 * DS_ROWS < rows) fetch(v, r0 + 2 * DS_ROWS);
        if (tid < DS_COLS) {
            const long long left = rows - r0;
            const int m = left < DS_ROWS ? (int)left : DS_ROWS;
            if (m == DS_ROWS) {
#pragma unroll 32
                for (int r = 0; r < DS_ROWS; ++r) s += buf[b][r][c];
            } else {
                for (int r = 0; r < m; ++r) s += buf[b][r][c];
            }
        }
    };
    fetch(va, 0);
    if (DS_ROWS < rows) fetch(vb, DS_ROWS);
    for (long long r0 = 0; r0 < rows; r0 += 2 * DS_ROWS) {
        batch(va, 0, r0);
        if (r0 + DS_ROWS < rows) batch(vb, 1, r0 + DS_ROWS);
    }
    if (tid < DS_COLS && col_ok) {
        const double avg = s / (double)rows;
        const double e = -((avg - mu) * (avg - mu)) / (2.0 * sigma * sigma);
        score[col] = exp(e);
    }
    if (range) {
        // thread (c, g) has seen rows g, g + 16, .. of column c: the 16 lane groups of a column through LDS (buf is free:
        // the last batch's adds are behind the barrier)
        __syncthreads();
        buf[0][g][c] = lo; buf[1][g][c] = hi;
        const bool any_bad = __ballot(bad) != 0;
        if ((tid & 63) == 0 && any_bad) atomicOr(&range[2], 1ull);
        __syncthreads();
        if (tid < DS_COLS && col_ok) {
            double l = buf[0][0][c], h = buf[1][0][c];
            for (int q = 1; q < DS_G; ++q) { l = fmin(l, buf[0][q][c]); h = fmax(h, buf[1][q][c]); }
            range[3 + col] = dlc_f64_key(l);
            range[3 + H + col] = dlc_f64_key(h);
        }
    }
}

// The same pass with the loads as LDS-DMA (H even, 16-byte aligned rows; the form above stays for the rest).  What set the
// pace of the form above was neither memory nor the add chain's own latency (a dependent v_add_f64 issues every 9 cycles,
// scripts/micro/add_f64_chain.hip: 0.13 ms for 31 890 rows) but the LDS round trips hipcc left between the adds -- "read two
// rows, wait for them, add them": 22 cycles a row, 0.34 ms, 1.9 TB/s however the columns were dealt.
// Here a workgroup owns 16 columns (one 128-byte line of every row: 157 workgroups at H = 2500, one per CU -- two workgroups
// on a CU put two chains on one SIMD: 8 columns per workgroup 0.27 ms); waves 1-3 fetch -- a batch of 384 rows = 48 KiB = 48
// DMA instructions (eight rows x eight 16-byte pieces each), three batches in the ring (144 KiB, two in flight) -- and take
// the extremes of what has landed (all of a batch's reads first, then the comparisons: one LDS round trip, not one per
// value -- that alone was 0.29 -> 0.20 ms); wave 0 does nothing but the chain: sixteen lanes, one column each, 32 rows at a
// time out of LDS into registers with the next 32 on their way.  One barrier per batch: "batch t has landed, batch t - 1 is
// consumed".  0.19 ms = 3.4 TB/s (batches of 192 rows, six in the ring: 0.20).
#ifndef DLC_DSD_COLS
#define DLC_DSD_COLS 16
#endif
constexpr int DSD_RB = 384, DSD_NB = 3;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dsd_dma(unsigned voff, const char* sbase, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds) : "memory");
}
#pragma clang diagnostic pop
template <int DSD_C>
__global__ __launch_bounds__(256) void distinctive_score_dma_kernel(const double* __restrict__ desc, long long rows, int H,
                                                                    double mu, double sigma, double* __restrict__ score,
                                                                    unsigned long long* __restrict__ range) {
    extern __shared__ __attribute__((aligned(16))) char dsd_smem[];
    constexpr int DSD_BB = DSD_RB * DSD_C * 8;                 // bytes of a batch
    constexpr int PPR = DSD_C / 2, RPI = 64 / PPR;             // 16-byte pieces per row; rows per DMA instruction
    constexpr int IPW = DSD_RB / RPI / 3;                      // DMA instructions per loader wave and batch
    constexpr int NQ = 192 / DSD_C;                            // threads of waves 1-3 per column (the extremes)
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const int groups = (H + DSD_C - 1) / DSD_C, per = (groups + 7) / 8;
    const int grp = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);      // XCD x: the x-th contiguous eighth of the groups
    if (grp >= groups) return;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col0 = grp * DSD_C;
    const long long rowb = (long long)H * 8, nb = (rows + DSD_RB - 1) / DSD_RB;
    const unsigned lds_base = (unsigned)(unsigned long long)(lds_ptr_t)dsd_smem;
    // loader lanes: piece (16 bytes = two columns) lane & 7 of row lane >> 3 of an instruction's eight rows; a piece past
    // the last column fetches piece 0 again (nobody reads it)
    const unsigned pc16 = (unsigned)((col0 + (lane % PPR) * 2 < H ? (lane % PPR) : 0) * 16);
    auto issue = [&](long long t, int slot) {                  // batch t (past the end: the last one again, into a dead slot)
        const long long tc = t < nb ? t : nb - 1, r0 = tc * DSD_RB;
        const long long last = rows - 1 - r0;                  // rows past the end re-read the last one
        const unsigned long long a = (unsigned long long)(desc + r0 * H + col0);
        const unsigned a_lo = __builtin_amdgcn_readfirstlane((unsigned)a), a_hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        const char* base = (const char*)(((unsigned long long)a_hi << 32) | a_lo);      // (unsigned halves: the builtin returns int)
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)slot * DSD_BB + (unsigned)(w - 1) * IPW * 1024);
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
            const long long ri = ((w - 1) * IPW + j) * RPI + lane / PPR;
            dsd_dma((unsigned)((ri < last ? ri : last) * rowb) + pc16, base, dst + j * 1024);
        }
    };
    // extremes: thread (c, q) of waves 1-3 sees rows q, q + 12, .. of column c of every batch
    const int rc = (tid - 64) % DSD_C, rq = (tid - 64) / DSD_C;
    const bool rcol_ok = col0 + rc < H;
    double lo = INFINITY, hi = -INFINITY, s = 0.0;
    bool bad = false;
    if (w > 0)
        for (int t = 0; t < DSD_NB - 1; ++t) issue(t, t);
    int slot = 0, free_slot = DSD_NB - 1;
    for (long long t = 0; t < nb; ++t) {
        if (w > 0) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(IPW * (DSD_NB - 2)) : "memory");    // this wave's pieces of batch t
        __syncthreads();                                       // batch t is whole; batch t - 1 is consumed
        const double* b = (const double*)(dsd_smem + slot * DSD_BB);
        const long long r0 = t * DSD_RB;
        const int m = rows - r0 < DSD_RB ? (int)(rows - r0) : DSD_RB;
        if (w > 0) {
            issue(t + DSD_NB - 1, free_slot);
            if (range && rcol_ok) {
                double v[DSD_RB / NQ];                         // all reads first: one LDS round trip, not one per value
#pragma unroll
                for (int u = 0; u < DSD_RB / NQ; ++u) v[u] = b[(rq + NQ * u) * DSD_C + rc];
#pragma unroll
                for (int u = 0; u < DSD_RB / NQ; ++u) {        // (rows past the end hold the last row again: no predicate)
                    bad |= !(fabs(v[u]) < INFINITY); lo = fmin(lo, v[u]); hi = fmax(hi, v[u]);
                }
            }
        } else if (lane < DSD_C) {
            // the chain: 32 rows at a time out of LDS into registers, the NEXT 32 on their way while these are added (left
            // to itself hipcc read two rows, waited for them, added them: an LDS round trip per pair, 40 cycles a row, and
            // the chain -- not the memory -- set the pace of both forms of this pass)
            if (m == DSD_RB) {
                double va[32], vb[32];
#pragma unroll
                for (int r = 0; r < 32; ++r) va[r] = b[r * DSD_C + lane];
                // "two adds, then one LDS read (two rows)": a dependent v_add_f64 issues every 9 cycles (scripts/micro/
                // add_f64_chain.hip), the reads of the next 32 rows ride in the gaps
#define DSD_PAIRS _Pragma("unroll") for (int z_ = 0; z_ < 16; ++z_) { __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#pragma unroll
                for (int blk = 0; blk < DSD_RB / 32; blk += 2) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 32; ++r) vb[r] = b[((blk + 1) * 32 + r) * DSD_C + lane];
#pragma unroll
                    for (int r = 0; r < 32; ++r) s += va[r];
                    DSD_PAIRS
                    __builtin_amdgcn_sched_barrier(0);
                    if (blk + 2 < DSD_RB / 32) {
#pragma unroll
                        for (int r = 0; r < 32; ++r) va[r] = b[((blk + 2) * 32 + r) * DSD_C + lane];
                    }
#pragma unroll
                    for (int r = 0; r < 32; ++r) s += vb[r];
                    if (blk + 2 < DSD_RB / 32) { DSD_PAIRS }
                    __builtin_amdgcn_sched_barrier(0);
                }
#undef DSD_PAIRS
            } else {
                for (int r = 0; r < m; ++r) s += b[r * DSD_C + lane];
            }
        }
        free_slot = slot;
        slot = slot == DSD_NB - 1 ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the redundant tail DMAs must not outlive the workgroup's LDS
    __syncthreads();
    if (w == 0 && lane < DSD_C && col0 + lane < H) {
        const double avg = s / (double)rows;
        const double e = -((avg - mu) * (avg - mu)) / (2.0 * sigma * sigma);
        score[col0 + lane] = exp(e);
    }
    if (range) {
        double* red = (double*)dsd_smem;                       // [2][NQ][DSD_C]
        if (w > 0) { red[rq * DSD_C + rc] = lo; red[192 + rq * DSD_C + rc] = hi; }
        const bool any_bad = __ballot(bad) != 0;
        if (lane == 0 && any_bad) atomicOr(&range[2], 1ull);
        __syncthreads();
        if (w == 0 && lane < DSD_C && col0 + lane < H) {
            double l = red[lane], h = red[192 + lane];
            for (int q = 1; q < NQ; ++q) { l = fmin(l, red[q * DSD_C + lane]); h = fmax(h, red[192 + q * DSD_C + lane]); }
            range[3 + col0 + lane] = dlc_f64_key(l);
            range[3 + H + col0 + lane] = dlc_f64_key(h);
        }
    }
}

__device__ __forceinline__ long long f64_to_i64_trunc(double v) {
    if (!(fabs(v) < 9.2233720368547758e18)) return (long long)0x8000000000000000ull;   // inf / nan / overflow
    return (long long)v;   // truncation toward zero
}

// |dot(score, m_a - m_b)| (SimilarityCalculator.py:42-43) from the two rows' double-double projections (gram_i8.hip:
// sim_rows_kernel): good to 1e-20 of the projections however close the rows lie -- bit-identical rows (a frame seen twice,
// a blank patch in both frames) have identical projections and give exactly 0 (log -> -inf, the score +inf, as the
// reference's).  r03 kept plain fp64 projections and walked both rows again wherever their difference had cancelled.
__device__ __forceinline__ double proj_diff(const double* __restrict__ proj, long long ra, long long rb) {
#pragma clang fp contract(off)
    const double2 a = *(const double2*)(proj + 2 * ra), b = *(const double2*)(proj + 2 * rb);
    const double d = a.x - b.x;
    const double bb = d - a.x;
    const double err = (a.x - (d - bb)) + (-b.x - bb);
    return fabs(d + (err + (a.y - b.y)));
}

// How close two squared distances |a|^2 + |b|^2 - 2 a.b of the fp64 Gram form may be before their order is not to be
// trusted: each of the three terms is an fp64 sum of H products, off by at most H 2^-53 times the sum of the products'
// magnitudes whatever the order (<= |a|^2, |b|^2, |a||b|), so one distance is off by at most ~2 H 2^-53 (|a|^2 + |b|^2) and two
// of them compare to within twice that (nb: the largest |b|^2 of the frame); 2^-50 relative covers roots that round to
// one double.  Candidates inside the window are evaluated directly (direct_argmin_wave).
__device__ __forceinline__ double gram_window(int H, double na, double nb, double best) {
    return 4.1 * (double)H * 0x1p-53 * (na + nb) + best * 0x1p-49;
}

// |x_c - x_a|^2 of NB candidate rows in ONE pass over k, all 64 lanes: per candidate the fma chain of a lane (k = lane,
// lane + 64, ..) and the xor tree are the same whatever NB is.  frac: some difference is not an integer below 2^18.
template <int NB>
__device__ __forceinline__ void direct_pass(const double* __restrict__ xa, const double* const (&xs)[4], int H, int lane,
                                            double (&ss)[4], bool& frac) {
    double s_[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) s_[c] = 0.0;
#pragma unroll 4
    for (int k = lane; k < H; k += 64) {
        const double av = xa[k];
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            const double d = xs[c][k] - av;
            s_[c] = fma(d, d, s_[c]);
            frac |= !(d == rint(d) && fabs(d) < 262144.0);
        }
    }
#pragma unroll
    for (int c = 0; c < NB; ++c) {
        for (int o = 32; o > 0; o >>= 1) s_[c] += __shfl_xor(s_[c], o);
        ss[c] = s_[c];
    }
}

// The arg-min of |x_b - x_a| over the candidate patches b of one frame (bit b of cm), decided as the reference decides
// it (SimilarityCalculator.py:30-37: np.argmin of np.linalg.norm, first minimum) when products of the descriptors -- the
// filter's integers or the fp64 Gram matrix -- cannot tell the candidates apart.  Called by a whole wave (uniform
// arguments); every lane returns the index.  stacks: 8 * PF_STACK_DEPTH doubles of LDS owned by this wave.
constexpr int PF_STACK_DEPTH = 16;                           // value stack of the pairwise-summation program (H <= 4 M: 16 levels)
constexpr int PF_STACK_BYTES = 4 * 8 * PF_STACK_DEPTH * 8;  // one per 8-lane group of each of a workgroup's 4 waves
__device__ __forceinline__ int direct_argmin_wave(const double* __restrict__ xa, const double* __restrict__ xj,
                                                  unsigned long long cm, int P, int H, int lane,
                                                  const int2* __restrict__ prog, int prog_len, double* stacks) {
    // (1) all 64 lanes on each candidate's 2 H doubles: squared distances to ~5e-15 (relative); lane b keeps
    // candidate b's
    double mine = INFINITY, emin = INFINITY;
    bool frac = false;                                  // some difference is not an integer below 2^18
    // (four candidates per pass over k -- a candidate's sum is the same fma chain per lane and the same xor tree as if it
    // went alone, but the passes' memory round trips are shared: an undecided arg-min has two or three candidates, and one
    // pass each made a direct evaluation 15-25 us on its one wave, the tail of the pair kernels and of the streaming query)
    for (unsigned long long rest = cm & (P >= 64 ? ~0ull : (1ull << P) - 1ull); rest;) {
        int bs[4];
        int nb = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bs[c] = bs[0];
            if (rest) { bs[c] = __ffsll((long long)rest) - 1; rest &= rest - 1; nb = c + 1; }
        }
        const double* xs[4] = {xj + (long long)bs[0] * H, xj + (long long)bs[1] * H, xj + (long long)bs[2] * H, xj + (long long)bs[3] * H};
        double ss[4] = {0.0, 0.0, 0.0, 0.0};
        if (nb == 1) direct_pass<1>(xa, xs, H, lane, ss, frac);
        else if (nb == 2) direct_pass<2>(xa, xs, H, lane, ss, frac);
        else if (nb == 3) direct_pass<3>(xa, xs, H, lane, ss, frac);
        else direct_pass<4>(xa, xs, H, lane, ss, frac);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < nb) {
                if (lane == bs[c]) mine = ss[c];
                emin = fmin(emin, ss[c]);
            }
    }
    // (only the candidates take part: the other lanes hold +inf, which would pass the test when every candidate's
    // squared distance overflows -- finite descriptors ~1e154 apart -- and send stages 1b / 2 to rows past the frame;
    // with no finite distance at all np.argmin takes the first candidate)
    unsigned long long close = __ballot(mine <= emin * (1.0 + 1e-11)) & cm;
    if (!(emin < INFINITY)) close = cm & (0ull - cm);
    int ebi = __ffsll((long long)close) - 1;
    bool same_rows = false;
    if ((close & (close - 1)) && __ballot(frac) == 0) {
        // (1a) integer differences (binary or integer-valued descriptors): squares and their sums below 2^51 are
        // exact in any order, NumPy's included, and distinct sums have distinct roots -- the first smallest sum
        close = __ballot(mine == emin) & cm;
        ebi = __ffsll((long long)close) - 1;
        close = 0;
    }
    if (close & (close - 1)) {
        // (1b) the commonest tie: the close candidates are the SAME patch (a key-point found twice, a blank patch
        // repeated) -- bit-identical rows have identical norms however they are summed, the first one wins
        const double* x0 = xj + (long long)ebi * H;
        bool differ = false;
        for (unsigned long long m = close & (close - 1); m && !differ; m &= m - 1) {
            const double* xb = xj + (long long)(__ffsll((long long)m) - 1) * H;
            bool d_ = false;
#pragma unroll 8
            for (int k = lane; k < H; k += 64) d_ |= __double_as_longlong(xb[k]) != __double_as_longlong(x0[k]);
            differ = __ballot(d_) != 0;
        }
        same_rows = !differ;
    }
    if ((close & (close - 1)) && !same_rows) {
        // (2) still closer than either summation resolves: the candidates' norms exactly as NumPy forms them
        // (np.linalg.norm: sqrt(np.add.reduce((x - m) ** 2)) with pairwise summation), eight candidates at a
        // time -- 8 lanes per candidate, one per strided accumulator of a leaf
        const int grp = lane >> 3, q = lane & 7;
        double* stack = stacks + grp * PF_STACK_DEPTH;                    // (this wave's eight)
        double nbest = 0.0;
        bool first = true;
        while (close) {
            int b_mine = -1, b_first = __ffsll((long long)close) - 1, nb_ = 0;
            for (int c = 0; c < 8 && close; ++c) {
                const int b = __ffsll((long long)close) - 1;
                close &= close - 1;
                if (grp == c) b_mine = b;
                ++nb_;
            }
            const double* xb = xj + (long long)(b_mine >= 0 ? b_mine : b_first) * H;
            // (HIP's __dmul_rn / __dadd_rn are plain * and + that hipcc contracts into fma: the pragma is what
            // keeps every product and every sum rounded on its own, as NumPy's are)
            auto sq = [&](int k) {
#pragma clang fp contract(off)
                const double d = xb[k] - xa[k];
                const double d2 = d * d;
                return d2;
            };
            int sp = 0;
            for (int e = 0; e < prog_len; ++e) {
#pragma clang fp contract(off)
                const int2 op = prog[e];
                if (op.x < 0) {
                    const double rhs = stack[sp - 1], lhs = stack[sp - 2];
                    sp -= 2;
                    const double r = __dadd_rn(lhs, rhs);
                    if (q == 0) stack[sp] = r;
                    ++sp;
                    continue;
                }
                double r;
                if (op.y < 8) {
                    r = 0.0;
                    for (int t = 0; t < op.y; ++t) r = __dadd_rn(r, sq(op.x + t));
                } else {
                    const int m = op.y - (op.y & 7);
                    r = sq(op.x + q);
#pragma unroll 4
                    for (int t = 8; t < m; t += 8) r = __dadd_rn(r, sq(op.x + t + q));
                    r = __dadd_rn(r, __shfl_xor(r, 1));
                    r = __dadd_rn(r, __shfl_xor(r, 2));
                    r = __dadd_rn(r, __shfl_xor(r, 4));
                    for (int t = m; t < op.y; ++t) r = __dadd_rn(r, sq(op.x + t));
                }
                if (q == 0) stack[sp] = r;
                ++sp;
            }
            const double dist = sqrt(0.0 + stack[0]);
            for (int c = 0; c < nb_; ++c) {
                const double dc = __shfl(dist, c * 8);
                const int bc = __shfl(b_mine, c * 8);
                if (first || dc < nbest) { nbest = dc; ebi = bc; first = false; }    // np.argmin: first minimum
            }
        }
    }
    return ebi;
}

// One wave per frame pair (i, j), i in [i_lo, i_hi), j in (i, N).  G is the Gram
// block  desc[i_lo*P .. i_hi*P) . desc[col0 ..)^T  with leading dimension ldg.
__global__ __launch_bounds__(256) void pair_score_kernel(const double* __restrict__ desc, const double* __restrict__ G,
                                                         long long ldg, long long col0, const double* __restrict__ nrm2,
                                                         const double* __restrict__ proj,
                                                         const double* __restrict__ score, long long N, int P, int H,
                                                         long long i_lo, long long i_hi, double ca, double cb,
                                                         double* __restrict__ out_f64, long long* __restrict__ out_i64,
                                                         const int2* __restrict__ prog, const unsigned long long* __restrict__ prog_len,
                                                         const unsigned long long* __restrict__ rowhash) {
    __shared__ double stacks[4][8 * PF_STACK_DEPTH];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long j = (long long)blockIdx.x * 4 + w;
    const long long i = i_lo + blockIdx.y;
    if (i >= i_hi || j >= N || j <= i) return;
    double term = 0.0, wd = 1.0;
    long long rb = 0;
    int bi = 0;
    unsigned long long cand = 0;
    const long long ra = i * P + (lane < P ? lane : 0);
    if (lane < P) {
        const double* grow = G + (ra - i_lo * P) * ldg + (j * P - col0);
        const double na = nrm2[ra];
        double best = INFINITY, nbmax = 0.0;
        for (int b = 0; b < P; ++b) {
            const double nbb = nrm2[j * P + b];
            const double d2 = fmax(na + nbb - 2.0 * grow[b], 0.0);
            nbmax = fmax(nbmax, nbb);
            if (d2 < best) { best = d2; bi = b; }               // strict: the first minimum keeps its index
        }
        // what the Gram form resolves: each of |a|^2, |b|^2, a.b is an fp64 sum of H products
        const double win = gram_window(H, na, nbmax, best);
        for (int b = 0; b < P; ++b) {
            const double d2 = fmax(na + nrm2[j * P + b] - 2.0 * grow[b], 0.0);
            if (!(d2 - best <= win)) continue;
            bool copy = false;                                  // a copy of an earlier candidate: np.argmin never takes it
            for (unsigned long long m = cand; m; m &= m - 1) {
                const long long e = j * P + (__ffsll((long long)m) - 1);
                copy |= rowhash[2 * e] == rowhash[2 * (j * P + b)] && rowhash[2 * e + 1] == rowhash[2 * (j * P + b) + 1];
            }
            if (!copy) cand |= 1ull << b;
        }
        if (!(cand & (cand - 1))) { bi = cand ? __ffsll((long long)cand) - 1 : bi; cand = 0; }
    }
    for (unsigned long long todo = __ballot(cand != 0); todo; todo &= todo - 1) {
        const int src = __ffsll((long long)todo) - 1;
        const unsigned long long cm = ((unsigned long long)(unsigned)__shfl((int)(cand >> 32), src) << 32) |
                                      (unsigned long long)(unsigned)__shfl((int)cand, src);
        const int ebi = direct_argmin_wave(desc + (i * P + src) * H, desc + j * P * H, cm, P, H, lane, prog, (int)*prog_len, stacks[w]);
        if (lane == src) bi = ebi;
    }
    if (lane < P) {
        rb = j * P + bi;
        wd = proj_diff(proj, ra, rb);                           // |dot(score, m_i - m_j*)|, :42-43
    }
    if (lane < P) term = ca + cb * log(wd);                     // :48
    for (int o = 32; o > 0; o >>= 1) term += __shfl_xor(term, o);
    if (lane == 0) {
        out_f64[i * N + j] = term;
        out_f64[j * N + i] = term;
        if (out_i64) {
            const long long t = f64_to_i64_trunc(term);
            out_i64[i * N + j] = t;
            out_i64[j * N + i] = t;
        }
    }
}

// The same for P <= 32 patches per frame (the reference: 30) with the Gram rows read as long contiguous runs: a workgroup takes frame i against PS_JT consecutive
// frames j, stages the P x (PS_JT * P) block of G through LDS (every row a run of PS_JT * P doubles -- 1920 bytes at
// P = 30 -- instead of 240-byte segments of two rows per load instruction, which read HBM at 1.8 TB/s), then thread
// (jj, a) walks the P distances of its patch a to frame j0 + jj in b order -- the general kernel's loop, so the first
// minimum and every rounding are the same -- and the 32-lane xor tree sums the P terms of a pair.
constexpr int PS_JT = 8;
constexpr int PS_GX = 8;        // workgroups per frame i: each walks every PS_GX-th run of PS_JT frames
__global__ __launch_bounds__(256) void pair_score_tile_kernel(const double* __restrict__ desc, const double* __restrict__ G,
                                                              long long ldg, long long col0, const double* __restrict__ nrm2,
                                                              const double* __restrict__ proj,
                                                              const double* __restrict__ score, long long N, int P, int H,
                                                              long long i_lo, long long i_hi, double ca, double cb,
                                                              double* __restrict__ out_f64, long long* __restrict__ out_i64,
                                                              const int2* __restrict__ prog,
                                                              const unsigned long long* __restrict__ prog_len,
                                                              const unsigned long long* __restrict__ rowhash) {
    extern __shared__ double ps_lds_raw[];
    double* ps_lds = ps_lds_raw + PF_STACK_BYTES / 8;            // in front: the summation program's value stacks, 8 per wave
    const long long i = i_lo + blockIdx.y;
    if (i >= i_hi) return;
    const int tid = threadIdx.x;
    const int width = PS_JT * P, row = width | 1;                // odd row pitch: 32 patches a read 32 different banks
    double* g = ps_lds;                                          // [P][row]
    double* nb = ps_lds + (size_t)P * row;                       // [PS_JT * P] squared norms of the frames' patches
    const int w = tid >> 6, lane = tid & 63;
    const int jj = tid >> 5, a = tid & 31;
    const long long ra = i * P + (a < P ? a : 0);
    const double na = nrm2[ra];
    const double* grow0 = G + (i * P - i_lo * P) * ldg - col0;   // column 0 of the matrix in row a = 0 (left of the block: only frames > i are read)
    // wave w stages rows w, w + 4, ..; a lane columns lane, lane + 64, ..: the (up to 32) loads of a run are issued
    // together, and the NEXT run's are in flight while this one is scored
    double v[8][4];
    auto fetch = [&](long long j0) {
        const long long jlo = j0 > i + 1 ? j0 : i + 1;           // first frame of the run that is wanted
        const long long jhi = j0 + PS_JT < N ? j0 + PS_JT : N;   // one past the last
        const int c_lo = (int)((jlo - j0) * P), c_hi = (int)((jhi - j0) * P);
        const double* gb = grow0 + j0 * P;
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int ar = w + 4 * r, c = lane + 64 * cc;
                v[r][cc] = (ar < P && c >= c_lo && c < c_hi) ? gb[(long long)ar * ldg + c] : 0.0;
            }
    };
    long long j0 = ((i + 1) / PS_JT + blockIdx.x) * PS_JT;       // the first run with a frame > i, then every PS_GX-th
    if (j0 < N) fetch(j0);
    for (; j0 < N; j0 += (long long)PS_GX * PS_JT) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int ar = w + 4 * r, c = lane + 64 * cc;
                if (ar < P && c < width) g[ar * row + c] = v[r][cc];
            }
        for (int c = tid; c < width; c += 256) nb[c] = (j0 * P + c < N * P) ? nrm2[j0 * P + c] : 0.0;
        __syncthreads();
        const long long jn = j0 + (long long)PS_GX * PS_JT;
        if (jn < N) fetch(jn);
        const long long j = j0 + jj;
        double term = 0.0, wd = 1.0;
        const bool pair_ok = j > i && j < N;
        int bi = 0;
        unsigned cand = 0;
        if (pair_ok && a < P) {
            const double* grow = g + a * row + jj * P;
            const double* nbj = nb + jj * P;
            // np.argmin(np.linalg.norm(..)) (:34-35) without 30 square roots: sqrt is monotone, so the first minimum of
            // the distances is the first minimum of the SQUARED distances unless another patch's square lies so close
            // above the smallest that the Gram form cannot order them (gram_window).  One pass keeps the two smallest
            // squares; only when the second is inside the window are the candidates collected -- copies of an earlier
            // candidate dropped by their content hashes -- and handed to the direct evaluation.
            double best = 0.0, second = INFINITY, nbmax = 0.0;
            auto scan = [&](int b) {                            // min / max instructions instead of compare-select chains
                const double d2 = fmax(na + nbj[b] - 2.0 * grow[b], 0.0);
                nbmax = fmax(nbmax, nbj[b]);
                if (b == 0) { best = d2; bi = 0; return; }
                second = fmin(second, fmax(best, d2));          // the smaller of the two that are not the new minimum
                bi = d2 < best ? b : bi;                        // strict: the first minimum keeps its index
                best = fmin(best, d2);
            };
            if (P == 30) {                                      // the reference's patch count: fully unrolled, all LDS reads up front
#pragma unroll
                for (int b = 0; b < 30; ++b) scan(b);
            } else {
#pragma unroll 4
                for (int b = 0; b < P; ++b) scan(b);
            }
            const double win = gram_window(H, na, nbmax, best);
            if (second - best <= win) {
                const unsigned long long* hbj = rowhash + 2 * (j * P);
                for (int b = 0; b < P; ++b) {
                    if (!(fmax(na + nbj[b] - 2.0 * grow[b], 0.0) - best <= win)) continue;
                    bool copy = false;
                    for (unsigned m = cand; m; m &= m - 1) {
                        const int e = __ffs((int)m) - 1;
                        copy |= hbj[2 * e] == hbj[2 * b] && hbj[2 * e + 1] == hbj[2 * b + 1];
                    }
                    if (!copy) cand |= 1u << b;
                }
                if (!(cand & (cand - 1))) { bi = __ffs((int)cand) - 1; cand = 0; }       // one patch left: decided
            }
        }
        for (unsigned long long todo = __ballot(cand != 0); todo; todo &= todo - 1) {    // this wave's undecided arg-mins
            const int src = __ffsll((long long)todo) - 1;
            const unsigned cm = (unsigned)__shfl((int)cand, src);
            const long long j_s = j0 + w * 2 + (src >> 5);
            const int ebi = direct_argmin_wave(desc + (i * P + (src & 31)) * H, desc + j_s * P * H, (unsigned long long)cm, P, H, lane,
                                               prog, (int)*prog_len, ps_lds_raw + (size_t)w * 8 * PF_STACK_DEPTH);
            if (lane == src) bi = ebi;
        }
        if (pair_ok && a < P) wd = proj_diff(proj, ra, j * P + bi);     // |dot(score, m_i - m_j*)|, :42-43
        if (pair_ok && a < P) term = ca + cb * log(wd);         // :48
        for (int o = 16; o > 0; o >>= 1) term += __shfl_xor(term, o);      // the 32 lanes of this pair (lanes >= P hold 0)
        if (pair_ok && a == 0) {
            out_f64[i * N + j] = term;
            out_f64[j * N + i] = term;
            if (out_i64) {
                const long long t = f64_to_i64_trunc(term);
                out_i64[i * N + j] = t;
                out_i64[j * N + i] = t;
            }
        }
        __syncthreads();                                         // g / nb are rewritten for the next run
    }
}

// The filter form (gram_i8.hip): the int8 product kernel has decided, for every row patch a and every later frame j, which
// patch b of frame j is nearest -- exact integer products of the column-centred descriptors' 24-bit fixed-point values v,
// |2 v_a . v_b - acc 2^-15| <= E, arg-min of |v_b|^2 - acc 2^-15 -- and left in abi / acand [nfp, rp] the index and, where
// the runner-up lay within 2 E (dlc_sim_window), the set of patches inside that window.  This kernel turns them into scores: undecided sets
// lose the copies of an earlier member (equal content hashes: the same distance, a later index) and what is left is
// evaluated directly -- |x_b - x_a| in fp64 from the descriptors, square roots compared as the reference compares them
// (np.argmin of np.linalg.norm, first minimum: direct_argmin_wave) -- then the P terms of the pair and their sum.
// Every wave works on its own: frame i against runs of TWO frames j (its two half-waves take one each, lane a of a half
// the patch a) -- no workgroup barrier, so a wave that has to evaluate candidates directly holds up nobody else.
__global__ __launch_bounds__(256) void pair_score_amin_kernel(const double* __restrict__ desc,
                                                              const unsigned char* __restrict__ abi,
                                                              const unsigned* __restrict__ acand, long long rp,
                                                              const double* __restrict__ proj, const double* __restrict__ score,
                                                              unsigned long long* __restrict__ keys, long long N, int P, int H,
                                                              double ca, double cb, double* __restrict__ out_f64,
                                                              long long* __restrict__ out_i64, const int2* __restrict__ prog,
                                                              const unsigned long long* __restrict__ rowhash,
                                                              unsigned char* __restrict__ direct_map) {
    extern __shared__ double ps_lds_all[];                  // the summation program's value stacks, 8 per wave
    unsigned long long* n_fallback = keys + 4;              // a count of the direct evaluations (-> stats[0] of the call)
    if (keys[2]) return;                                    // a NaN / infinity in the dataset: this form does not apply
    const int prog_len = (int)keys[5];
    const long long i = blockIdx.y;
    if (i >= N - 1) return;
    const int tid = threadIdx.x;
    const int w = tid >> 6, lane = tid & 63;
    const int jj = lane >> 5, a = lane & 31;
    const long long ra = i * P + (a < P ? a : 0);
    const bool flat = !(dlc_f64_unkey(keys[0]) > 0.0);         // every column constant -- all patches the same row: all distances 0
    const long long nwaves = (long long)PS_GX * 4;               // waves per frame i: wave q takes the runs q, q + nwaves, ..
    unsigned long long fallbacks = 0;
    for (long long j0 = ((i + 1) / 2 + blockIdx.x * 4 + w) * 2; j0 < N; j0 += nwaves * 2) {
        const long long j = j0 + jj;
        const bool pair_ok = j > i && j < N;
        const bool live = pair_ok && a < P;
        int bi = 0;
        unsigned cand = 0;
        if (live && !flat) {
            bi = abi[j * rp + ra];
            cand = acand[j * rp + ra];
            if (cand) {
                // a copy of an earlier candidate (equal content hashes: sim_mix64) has that candidate's distance and a
                // later index: np.argmin never takes it
                const unsigned long long* hbj = rowhash + 2 * (j * P);
                unsigned kept = 0;
                for (unsigned m = cand; m; m &= m - 1) {
                    const int b = __ffs((int)m) - 1;
                    bool copy = false;
                    for (unsigned k2 = kept; k2; k2 &= k2 - 1) {
                        const int e = __ffs((int)k2) - 1;
                        copy |= hbj[2 * e] == hbj[2 * b] && hbj[2 * e + 1] == hbj[2 * b + 1];
                    }
                    if (!copy) kept |= 1u << b;
                }
                cand = kept;
                if (!(cand & (cand - 1))) { bi = __ffs((int)cand) - 1; cand = 0; }       // one patch left: decided
            }
        }
        // the undecided arg-mins of this wave, one after the other
        unsigned long long todo = __ballot(cand != 0);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const unsigned cm = (unsigned)__shfl((int)cand, src);
            const int a_s = src & 31;
            const long long j_s = j0 + (src >> 5);
            const int ebi = direct_argmin_wave(desc + (i * P + a_s) * H, desc + j_s * P * H, (unsigned long long)cm, P, H, lane,
                                               prog, prog_len, ps_lds_all + (size_t)w * 8 * PF_STACK_DEPTH);
            if (lane == src) bi = ebi;
            if (lane == 0) {
                ++fallbacks;
                if (direct_map) direct_map[i * N + j_s] = 1;
            }
        }
        double term = 0.0, wd = 1.0;
        if (live) wd = proj_diff(proj, ra, j * P + bi);         // |dot(score, m_i - m_j*)|, :42-43
        if (live) term = ca + cb * log(wd);                     // :48
        for (int o = 16; o > 0; o >>= 1) term += __shfl_xor(term, o);
        if (pair_ok && a == 0) {
            out_f64[i * N + j] = term;
            out_f64[j * N + i] = term;
            if (out_i64) {
                const long long t = f64_to_i64_trunc(term);
                out_i64[i * N + j] = t;
                out_i64[j * N + i] = t;
            }
        }
    }
    if (lane == 0 && fallbacks) atomicAdd(n_fallback, fallbacks);
}

// The filter's EXIT.  The int8 products decide an arg-min when the runner-up lies outside the error window; data whose
// distances sit inside it (one column with a range a thousand times the others', say) would send every arg-min to the
// one-wave direct evaluation -- 15-25 us each, 17 M of them at 1063 frames -- where the fp64 Gram form takes 39 ms.
// Before the product kernel starts, SIM_SAMPLES cells (row patch a, later frame j), picked by a fixed hash, are worked out
// exactly -- |x_b - x_a|^2 in fp64 for the P patches of frame j, copies of an earlier patch dropped as the pair kernel
// drops them -- and the gap between the two smallest is held against the window; when more than an eighth of the cells
// would be undecided the last workgroup sets bit 1 of keys[2]: the filter's kernels leave at once, and the host, which
// reads that word anyway (the NaN flag), sends the call down the fp64 route.  A workgroup per cell, eight candidates per
// wave; 160 MB of descriptor rows, ~40 us.  (Not launched under DLC_SIM_NO_HOST_SYNC: nobody could act on the verdict.)
constexpr int SIM_SAMPLES = 256;
__global__ __launch_bounds__(256) void sim_sample_kernel(const double* __restrict__ desc,
                                                         const unsigned long long* __restrict__ rowhash,
                                                         unsigned long long* __restrict__ keys, long long N, int P, int H) {
    __shared__ double d2s[32];
    if (keys[2]) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long z = ((unsigned long long)blockIdx.x + 1) * 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    z ^= z >> 31;
    const long long i = (long long)(z % (unsigned long long)(N - 1));
    const int a = (int)((z >> 24) % (unsigned)P);
    const long long j = i + 1 + (long long)((z >> 40) % (unsigned long long)(N - 1 - i));
    const double* xa = desc + (i * P + a) * H;
    const double* xj = desc + j * P * H;
    for (int pass = 0; pass < 2; ++pass) {
        const int b0 = w * 8 + pass * 4;
        if (b0 >= P) break;                                       // (whole wave)
        const double* xs[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) xs[c] = xj + (long long)(b0 + c < P ? b0 + c : b0) * H;
        double ss[4] = {0.0, 0.0, 0.0, 0.0};
        bool frac = false;
        direct_pass<4>(xa, xs, H, lane, ss, frac);
        if (lane < 4 && b0 + lane < P) d2s[b0 + lane] = lane == 0 ? ss[0] : lane == 1 ? ss[1] : lane == 2 ? ss[2] : ss[3];
    }
    __syncthreads();
    if (w != 0) return;
    const double s_ = dlc_f64_unkey(keys[0]);
    const double inv = s_ > 0.0 ? 0.996 / s_ : 0.0;
    double d = lane < P ? d2s[lane] * inv * inv : INFINITY;       // in the filter's units (|v_b - v_a|^2)
    if (lane < P) {
        const unsigned long long* hb = rowhash + 2 * (j * P);
        bool copy = false;
        for (int e = 0; e < lane; ++e) copy |= hb[2 * e] == hb[2 * lane] && hb[2 * e + 1] == hb[2 * lane + 1];
        if (copy) d = INFINITY;
    }
    double best = d;
    for (int o = 32; o > 0; o >>= 1) best = fmin(best, __shfl_xor(best, o));
    const int fb = __ffsll((long long)__ballot(d == best)) - 1;
    double second = lane == fb ? INFINITY : d;
    for (int o = 32; o > 0; o >>= 1) second = fmin(second, __shfl_xor(second, o));
    const double window = (double)dlc_gemm::dlc_sim_window(keys, H) * 0x1p-15;
    if (lane == 0) {
        if (second - best <= window) atomicAdd(&keys[6], 1ull);
        __threadfence();
        const unsigned long long done = atomicAdd(&keys[7], 1ull);
        if (done == (unsigned long long)gridDim.x - 1 && atomicAdd(&keys[6], 0ull) * 8 > (unsigned long long)gridDim.x)
            atomicOr(&keys[2], 2ull);
    }
}

__global__ void fill_diag_kernel(long long N, double* out_f64, long long* out_i64) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    out_f64[i * N + i] = -1.0;
    if (out_i64) out_i64[i * N + i] = -1;
}

// out[c, r] = in[r, c] through a padded LDS tile (32 x 32)
__global__ __launch_bounds__(256) void transpose_f64_kernel(const double* __restrict__ in, long long rows, long long cols,
                                                            double* __restrict__ out) {
    __shared__ double tile[32][33];
    const long long r0 = (long long)blockIdx.x * 32, c0 = (long long)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = in[(r0 + i) * cols + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < cols && r0 + tx < rows) out[(c0 + i) * rows + r0 + tx] = tile[tx][i];
}

// ---------------------------------------------------------------------------------------------------------------------
// Streaming form of the similarity (SimilarityCalculator.similarity_score for ONE new frame, the shape the loop of
// create_similarity_matrix.py:34-38 takes when a robot adds a frame): frame f against every OLDER resident frame j < f --
// row[j] = score(h_j, h_f) = entry [j, f] of the matrix call, bit for bit.  The database side is resident: descriptors
// [capacity, P, H] fp64 and their quantised panel, |u|^2, projections and hashes, appended frame by frame over a range
// FIXED at creation (no re-quantisation of older frames).  Two kernels per query:
//   stream_argmin_kernel  every wave streams two 16-patch groups of the database panel ONCE (245 MB at 1063 frames: the
//                         HBM-bound part, plain 16-byte loads in MFMA operand order -- no LDS) against the three groups
//                         that hold frame f's patches (L2-resident), three class accumulators, and decides for each of
//                         its 32 database patches a which patch b of frame f is nearest -- the filter's rule: integer
//                         arg-min, candidates inside the error window evaluated directly (direct_argmin_wave);
//   stream_score_kernel   the P terms of every older frame and their sum in the pair kernels' order.
typedef __attribute__((ext_vector_type(4))) int sr_v4i;
constexpr int SR_G = 2;                 // database groups per wave

__device__ __forceinline__ void sr_merge(int& best, int& bidx, int& second, int ob, int oi, int os) {
    // (best, index of the FIRST minimum, runner-up) of two disjoint candidate sets
    const bool take = ob < best || (ob == best && oi < bidx);
    const int lose = take ? best : ob;
    second = min(min(second, os), lose);
    if (take) { best = ob; bidx = oi; }
}

// NQ: query frames per pass over the database panel.  NQ = 1 is the single query (two register rings four k-steps deep: the
// latency form).  NQ = 2 scores TWO consecutive frames per pass -- their 2 P patches lie in five consecutive groups of the
// panel -- so a batch of frames that arrived together streams the older frames' panel once per pair instead of once per
// frame (a batch of 32 one frame per pass: 47 us per query, all of it the 245 MB panel at 5.2 TB/s).  Query y of pass
// blockIdx.y is frame f + blockIdx.y * NQ + y (< f_end); its verdicts go to row blockIdx.y * NQ + y of bi_out.
template <int NQ>
__global__ __launch_bounds__(256) void stream_argmin_kernel(const char* __restrict__ X, long long gpitch, int n64,
                                                            const double* __restrict__ desc, const double* __restrict__ nu2,
                                                            const unsigned long long* __restrict__ rowhash,
                                                            unsigned long long* __restrict__ keys, long long f, long long f_end, int P,
                                                            int H, unsigned char* __restrict__ bi_out, long long bi_pitch,
                                                            const int2* __restrict__ prog) {
    constexpr int NG = 2 * NQ + 1;                                 // query-side groups: NQ * P + 15 patches at most
    constexpr int DEPTH = NQ == 1 ? 4 : 2;                         // k-steps in flight (registers: rings of DEPTH buffers)
    __shared__ double stacks[4][8 * PF_STACK_DEPTH];
    f += (long long)blockIdx.y * NQ;                               // the pass's first query frame
    bi_out += (long long)blockIdx.y * NQ * bi_pitch;
    const int nv = (int)(f_end - f < NQ ? f_end - f : NQ);         // query frames of this pass (>= 1)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, quad = lane >> 4, an = lane & 15;
    const long long rows_q0 = f * P;                               // first patch of the pass's first query frame
    const long long rows_last = (f + nv - 1) * P;                  // database patches of its LAST query: rows 0 .. rows_last - 1
    const long long gd0 = ((long long)blockIdx.x * 4 + w) * SR_G;
    // (the count of direct evaluations, keys[4], is zero: the stream's creation and every query's last kernel leave it so)
    if (gd0 * 16 >= rows_last) return;                             // (whole wave)
    const long long gq0 = rows_q0 / 16;                            // the query frames' patches lie in groups gq0 .. gq0 + NG - 1
    const int boff0 = (int)(rows_q0 - gq0 * 16);
    const char* xq = X + gq0 * gpitch + lane * 16;
    const char* xd = X + gd0 * gpitch + lane * 16;
    sr_v4i c2[NG][SR_G], c3[NG][SR_G], c4[NG][SR_G];
#pragma unroll
    for (int jq = 0; jq < NG; ++jq)
#pragma unroll
        for (int ig = 0; ig < SR_G; ++ig) { c2[jq][ig] = sr_v4i{0, 0, 0, 0}; c3[jq][ig] = sr_v4i{0, 0, 0, 0}; c4[jq][ig] = sr_v4i{0, 0, 0, 0}; }
    // The database fragments come from HBM (a wave's 6 KiB per k-step, read once, non-temporal), the queries' from L2
    // (3 KiB per group and k-step, the same for every wave); both are fetched DEPTH - 1 k-steps ahead through rings of
    // register buffers -- one k-step ahead, a wave finished a k-step per memory round trip: 25 us for the K loop alone.
    sr_v4i q[DEPTH][3][NG], d[DEPTH][3][SR_G];                     // [ring buffer][slice][group]
    auto fetch = [&](int buf, int t) {
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_) {
            const long long ko = ((long long)s_ * n64 + t) * 1024;
#pragma unroll
            for (int ig = 0; ig < SR_G; ++ig) d[buf][s_][ig] = __builtin_nontemporal_load((const sr_v4i*)(xd + ig * gpitch + ko));
#pragma unroll
            for (int jq = 0; jq < NG; ++jq) q[buf][s_][jq] = *(const sr_v4i*)(xq + jq * gpitch + ko);
        }
    };
#define SR_STEP(B)                                                                                                        \
    _Pragma("unroll") for (int jq = 0; jq < NG; ++jq)                                                                     \
        _Pragma("unroll") for (int ig = 0; ig < SR_G; ++ig) {                                                             \
            c2[jq][ig] = __builtin_amdgcn_mfma_i32_16x16x64_i8(q[B][0][jq], d[B][0][ig], c2[jq][ig], 0, 0, 0);            \
            c3[jq][ig] = __builtin_amdgcn_mfma_i32_16x16x64_i8(q[B][1][jq], d[B][0][ig], c3[jq][ig], 0, 0, 0);            \
            c3[jq][ig] = __builtin_amdgcn_mfma_i32_16x16x64_i8(q[B][0][jq], d[B][1][ig], c3[jq][ig], 0, 0, 0);            \
            c4[jq][ig] = __builtin_amdgcn_mfma_i32_16x16x64_i8(q[B][2][jq], d[B][0][ig], c4[jq][ig], 0, 0, 0);            \
            c4[jq][ig] = __builtin_amdgcn_mfma_i32_16x16x64_i8(q[B][1][jq], d[B][1][ig], c4[jq][ig], 0, 0, 0);            \
            c4[jq][ig] = __builtin_amdgcn_mfma_i32_16x16x64_i8(q[B][0][jq], d[B][2][ig], c4[jq][ig], 0, 0, 0);            \
        }
#pragma unroll
    for (int i = 0; i < DEPTH - 1; ++i) fetch(i, i);               // n64 >= 4 > DEPTH - 1
    for (int t = 0; t < n64; t += DEPTH) {                         // k-step s lives in ring buffer s % DEPTH
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
            if (t + i + DEPTH - 1 < n64) fetch((i + DEPTH - 1) % DEPTH, t + i + DEPTH - 1);
            if (t + i < n64) { SR_STEP(i) }
        }
    }
#undef SR_STEP
    // D[m][n] of MFMA (jq, ig): m = query-side patch jq * 16 + quad * 4 + v (patch b = that - boff of its frame), n = database
    // patch an of group ig
    const long long window = dlc_gemm::dlc_sim_window(keys, H);
    const int prog_len = (int)keys[5];
    unsigned long long directs = 0;
#pragma unroll
    for (int y = 0; y < NQ; ++y) {
        if (y >= nv) break;                                        // (uniform)
        const long long rows_old = rows_q0 + (long long)y * P;     // query frame f + y: database patches 0 .. rows_old - 1
        const int boff = boff0 + y * P;
        int nbv[NG][4];                                            // |v_b|^2 of this lane's query-side patches that belong to frame y, units of 2^-15
#pragma unroll
        for (int jq = 0; jq < NG; ++jq)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int b = jq * 16 + quad * 4 + v - boff;
                nbv[jq][v] = (b >= 0 && b < P) ? (int)llrint(nu2[rows_old + b] * 32768.0) : 0;
            }
#pragma unroll
        for (int ig = 0; ig < SR_G; ++ig) {
            const long long a = (gd0 + ig) * 16 + an;
            const bool a_ok = a < rows_old;
            int best = 0x7fffffff, second = 0x7fffffff, bidx = 0x7fffffff;
            int d2v[NG][4];
#pragma unroll
            for (int jq = 0; jq < NG; ++jq) {
                const sr_v4i acc = c2[jq][ig] + ((c3[jq][ig] + (c4[jq][ig] >> 8)) >> 8);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int b = jq * 16 + quad * 4 + v - boff;
                    d2v[jq][v] = nbv[jq][v] - acc[v];
                    if (b >= 0 && b < P) sr_merge(best, bidx, second, d2v[jq][v], b, 0x7fffffff);
                }
            }
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {                   // the four quads hold the rest of patch a's row
                const int ob = __shfl_xor(best, o), oi = __shfl_xor(bidx, o), os = __shfl_xor(second, o);
                sr_merge(best, bidx, second, ob, oi, os);
            }
            int bi = bidx < P ? bidx : 0;
            unsigned cand = 0;
            if ((long long)second - best <= window) {
#pragma unroll
                for (int jq = 0; jq < NG; ++jq)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int b = jq * 16 + quad * 4 + v - boff;
                        if (b >= 0 && b < P && (long long)d2v[jq][v] - best <= window) cand |= 1u << b;
                    }
            }
            cand |= (unsigned)__shfl_xor((int)cand, 16);
            cand |= (unsigned)__shfl_xor((int)cand, 32);
            if (cand & (cand - 1)) {
                // a copy of an earlier candidate (equal content hashes) has that candidate's distance and a later index
                const unsigned long long* hb = rowhash + 2 * rows_old;
                unsigned kept = 0;
                for (unsigned m = cand; m; m &= m - 1) {
                    const int b = __ffs((int)m) - 1;
                    bool copy = false;
                    for (unsigned k2 = kept; k2; k2 &= k2 - 1) {
                        const int e = __ffs((int)k2) - 1;
                        copy |= hb[2 * e] == hb[2 * b] && hb[2 * e + 1] == hb[2 * b + 1];
                    }
                    if (!copy) kept |= 1u << b;
                }
                cand = kept;
                if (!(cand & (cand - 1))) { bi = __ffs((int)cand) - 1; cand = 0; }
            } else cand = 0;
            for (unsigned long long todo = __ballot(cand != 0 && quad == 0 && a_ok); todo; todo &= todo - 1) {
                const int src = __ffsll((long long)todo) - 1;
                const unsigned cm = (unsigned)__shfl((int)cand, src);
                const long long a_s = (gd0 + ig) * 16 + (src & 15);
                const int ebi = direct_argmin_wave(desc + a_s * H, desc + rows_old * H, (unsigned long long)cm, P, H, lane, prog,
                                                   prog_len, stacks[w]);
                if (lane == src) bi = ebi;
                if (lane == 0) ++directs;
            }
            if (quad == 0 && a_ok) bi_out[(long long)y * bi_pitch + a] = (unsigned char)bi;
        }
    }
    if (lane == 0 && directs) atomicAdd(&keys[4], directs);
}

// row[j] = sum over the P patches a of frame j of  ca + cb log | dot(score, m_a - m_b*) |  (SimilarityCalculator.py:40-49),
// b* = bi[j P + a] the patch of frame f nearest to a: the tail of the pair kernels, term for term and in their summation
// order (32-lane xor tree), so the row equals the matrix call's column f bit for bit.
__global__ __launch_bounds__(256) void stream_score_kernel(const double* __restrict__ desc, const double* __restrict__ proj,
                                                           const double* __restrict__ score,
                                                           const unsigned char* __restrict__ bi_in, long long bi_pitch,
                                                           unsigned long long* __restrict__ keys, long long f, int P, int H,
                                                           double ca, double cb, double* __restrict__ row, long long ld_row,
                                                           long long* __restrict__ stats) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, jj = lane >> 5, a = lane & 31;
    const long long j = ((long long)blockIdx.x * 4 + w) * 2 + jj;
    f += blockIdx.y;                                            // a batch: query y is frame f + y, row y of the output
    bi_in += (long long)blockIdx.y * bi_pitch;
    row += (long long)blockIdx.y * ld_row;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {   // the call's count of direct evaluations: handed over and
        if (stats) { stats[0] = (long long)keys[4]; stats[1] = keys[2] ? 1 : 0; }
        keys[4] = 0ull;                                         // reset for the next query (no memset launch in front of it)
    }
    const bool pair_ok = j < f, live = pair_ok && a < P;
    double term = 0.0, wd = 1.0;
    if (live) {
        const long long ra = j * P + a;
        wd = proj_diff(proj, ra, f * P + bi_in[ra]);            // |dot(score, m_i - m_j*)|, :42-43
    }
    if (live) term = ca + cb * log(wd);                         // :48
    for (int o = 16; o > 0; o >>= 1) term += __shfl_xor(term, o);
    if (pair_ok && a == 0) row[j] = keys[2] ? __longlong_as_double(0x7ff8000000000000ll) : term;
}

// The batched query through the matrix call's product kernel (gram_argmin_i8_strip: the batch's frames are a strip of
// columns): its verdicts abi / acand [strip frame, rp] -> the final arg-mins bi_out [query, bi_pitch] that
// stream_score_kernel reads, as stream_argmin_kernel leaves them.  Undecided sets are treated as pair_score_amin_kernel
// treats them: copies of an earlier member dropped (equal content hashes), what is left evaluated directly in fp64
// (direct_argmin_wave).  Lane = one older patch; a wave that has to evaluate holds up nobody else.
__global__ __launch_bounds__(256) void strip_resolve_kernel(const double* __restrict__ desc, const unsigned char* __restrict__ abi,
                                                            const unsigned* __restrict__ acand, long long rp, long long fj_base,
                                                            const unsigned long long* __restrict__ rowhash,
                                                            unsigned long long* __restrict__ keys, long long f_first, int P, int H,
                                                            const int2* __restrict__ prog, unsigned char* __restrict__ bi_out,
                                                            long long bi_pitch) {
    extern __shared__ double rs_lds_all[];                  // the summation program's value stacks, 8 per wave
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const long long fj = f_first + blockIdx.y;
    const long long ra = (long long)blockIdx.x * 256 + tid;
    const bool live = ra < fj * P;
    if (keys[2]) {                                          // poisoned stream: the product kernel left at once; rows come back NaN
        if (live) bi_out[(long long)blockIdx.y * bi_pitch + ra] = 0;
        return;
    }
    const int prog_len = (int)keys[5];
    int bi = 0;
    unsigned cand = 0;
    if (live) {
        bi = abi[(fj - fj_base) * rp + ra];
        cand = acand[(fj - fj_base) * rp + ra];
        if (cand) {
            const unsigned long long* hbj = rowhash + 2 * (fj * P);
            unsigned kept = 0;
            for (unsigned m = cand; m; m &= m - 1) {
                const int b = __ffs((int)m) - 1;
                bool copy = false;
                for (unsigned k2 = kept; k2; k2 &= k2 - 1) {
                    const int e = __ffs((int)k2) - 1;
                    copy |= hbj[2 * e] == hbj[2 * b] && hbj[2 * e + 1] == hbj[2 * b + 1];
                }
                if (!copy) kept |= 1u << b;
            }
            cand = kept;
            if (!(cand & (cand - 1))) { bi = __ffs((int)cand) - 1; cand = 0; }           // one patch left: decided
        }
    }
    unsigned long long directs = 0;
    for (unsigned long long todo = __ballot(cand != 0); todo; todo &= todo - 1) {
        const int src = __ffsll((long long)todo) - 1;
        const unsigned cm = (unsigned)__shfl((int)cand, src);
        const long long ra_s = (long long)blockIdx.x * 256 + w * 64 + src;
        const int ebi = direct_argmin_wave(desc + ra_s * H, desc + fj * P * H, (unsigned long long)cm, P, H, lane, prog, prog_len,
                                           rs_lds_all + (size_t)w * 8 * PF_STACK_DEPTH);
        if (lane == src) bi = ebi;
        if (lane == 0) ++directs;
    }
    if (live) bi_out[(long long)blockIdx.y * bi_pitch + ra] = (unsigned char)bi;
    if (lane == 0 && directs) atomicAdd(&keys[4], directs);
}

// The k best entries of every row of an fp64 score matrix (the loop-closure candidates of a batch of streamed frames,
// loop_closure.py: SdavLoopClosureDetector): row r offers its first limit0 + r * limit_step entries; order: score
// descending, ties -> the lower index (the older frame); a NaN is never taken.  One workgroup per row, k rounds of
// "the best entry after the last one taken" -- rows of a few thousand scores, k a handful: microseconds.
__global__ __launch_bounds__(256) void topk_rows_f64_kernel(const double* __restrict__ scores, long long ld, long long limit0,
                                                            long long limit_step, int k, double* __restrict__ out_s,
                                                            long long* __restrict__ out_i,
                                                            const long long* __restrict__ poison) {
    __shared__ unsigned long long red_k[4];
    __shared__ long long red_i[4];
    const long long r = blockIdx.x;
    if (poison && *poison != 0) {                                // the caller's "these scores mean nothing" word: say so in the
        for (int t = threadIdx.x; t < k; t += 256) {             // output itself (NaN, -1), not with an empty list
            out_s[r * k + t] = __longlong_as_double(0x7ff8000000000000ll);
            out_i[r * k + t] = -1;
        }
        return;
    }
    long long n = limit0 + r * limit_step;
    if (n > ld) n = ld;
    const double* row = scores + r * ld;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long last_k = ~0ull;                            // key of the last entry taken (keys order like the doubles)
    long long last_i = -1;
    for (int t = 0; t < k; ++t) {
        unsigned long long bk = 0ull;                            // (0 is below every key of a number: "none")
        long long bi = 0x7fffffffffffffffll;
        for (long long i = threadIdx.x; i < n; i += 256) {
            const double v = row[i];
            if (v != v) continue;
            const unsigned long long kk = dlc_f64_key(v);
            const bool after = kk < last_k || (kk == last_k && i > last_i);
            if (after && (kk > bk || (kk == bk && i < bi))) { bk = kk; bi = i; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long ok = ((unsigned long long)(unsigned)__shfl_xor((int)(bk >> 32), o) << 32) | (unsigned)__shfl_xor((int)bk, o);
            const long long oi = (long long)(((unsigned long long)(unsigned)__shfl_xor((int)((unsigned long long)bi >> 32), o) << 32) |
                                             (unsigned)__shfl_xor((int)bi, o));
            if (ok > bk || (ok == bk && oi < bi)) { bk = ok; bi = oi; }
        }
        __syncthreads();                                         // (the previous round's reads of red_* are done)
        if (lane == 0) { red_k[w] = bk; red_i[w] = bi; }
        __syncthreads();
        bk = red_k[0]; bi = red_i[0];
        for (int ww = 1; ww < 4; ++ww)
            if (red_k[ww] > bk || (red_k[ww] == bk && red_i[ww] < bi)) { bk = red_k[ww]; bi = red_i[ww]; }
        const bool none = bk == 0ull;
        if (threadIdx.x == 0) {
            out_s[r * k + t] = none ? -INFINITY : dlc_f64_unkey(bk);
            out_i[r * k + t] = none ? -1 : bi;
        }
        if (none) { last_k = 0ull; last_i = 0x7fffffffffffffffll; }   // nothing is after "none": the remaining slots stay empty
        else { last_k = bk; last_i = bi; }
    }
}

struct StreamWs {
    size_t keys, prog, cc, nu2, proj, rowhash, bi, nbp, panel, total;
    long long zrow;            // an all-zero group of the panel (the last of the eight spare ones behind the capacity's rows)
    size_t panel_groups;
};
StreamWs stream_ws(int64_t capacity, int64_t P, int64_t H) {
    StreamWs w;
    size_t o = 0;
    const size_t rows = (size_t)capacity * P;
    w.keys = o; o += 256;
    w.prog = o; o += 8192;
    w.cc = o; o += dlc::align_up((size_t)H * 8, 256);
    w.nu2 = o; o += dlc::align_up(rows * 8, 256);
    w.proj = o; o += dlc::align_up(rows * 16, 256);               // double-double projections
    w.rowhash = o; o += dlc::align_up(rows * 16, 256);
    w.bi = o; o += dlc::align_up(rows, 256);
    w.nbp = o; o += dlc::align_up((size_t)dlc_gemm::sim_col_rows(capacity, P) * 4, 256);   // |v|^2 in the product kernel's unit layout
    w.panel = o; o += dlc::align_up(dlc_gemm::sim_stream_panel_bytes(rows, H), 256);
    w.panel_groups = (size_t)dlc::cdiv((int64_t)rows, (int64_t)16) + 8;
    w.zrow = (long long)(w.panel_groups - 1) * 16;
    w.total = o;
    return w;
}

int stream_check(dlc_ctx* ctx, const char* what, const void* state, size_t state_bytes, int64_t capacity, int64_t P, int64_t H) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!state || capacity < 1 || P < 1 || H < 1) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "%s: bad argument", what);
    if (P > 32 || H > 32768)
        return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "%s: the streaming form takes P <= 32 patches and H <= 32768 (the filter's shapes)", what);
    if (capacity * P > 0x7fffffffll) return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "%s: capacity too large", what);
    if (state_bytes < stream_ws(capacity, P, H).total)
        return dlc::fail(ctx, DLC_ERR_WORKSPACE, "%s: state %zu < %zu bytes", what, state_bytes, stream_ws(capacity, P, H).total);
    if ((uintptr_t)state & 255) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "%s: state must be 256-byte aligned", what);
    return DLC_OK;
}

// stats of a call that took the fp64 route after all: no direct evaluations of the filter's, and why (the flag bits)
__global__ void sim_stats_kernel(long long* stats, long long why) {
    if (threadIdx.x == 0) { stats[0] = 0; stats[1] = why; }
}

// keys[2] (flag bits) and keys[4] (direct evaluations) of a finished call -> the caller's stats; and, for a call
// that may not read the flag on the host (DLC_SIM_NO_HOST_SYNC), NaN over the matrix when the filter form did not apply.
__global__ __launch_bounds__(256) void sim_finish_kernel(const unsigned long long* __restrict__ keys, long long* __restrict__ stats,
                                                         long long nn, double* __restrict__ out_f64,
                                                         long long* __restrict__ out_i64, int poison) {
    const bool bad = keys[2] != 0;
    if (stats && blockIdx.x == 0 && threadIdx.x == 0) { stats[0] = (long long)keys[4]; stats[1] = (long long)keys[2]; }
    if (!poison || !bad) return;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < nn; e += (long long)gridDim.x * 256) {
        out_f64[e] = __longlong_as_double(0x7ff8000000000000ll);
        if (out_i64) out_i64[e] = (long long)0x8000000000000000ull;
    }
}

struct SimWs {
    size_t nrm2, proj, gram, gram_bytes, desc_t, keys, prog, nu2, rowhash, qx, nbp, abi, acand, blk, cc, range, total;
    long long chunk_frames, nfp;
};

// The arg-min filter (gram_i8.hip) takes the call when the tiled pair kernel does (P <= 32) and the integer
// accumulators cannot overflow; DLC_SIM_FORCE_F64 in the call's flags forces the fp64 Gram form.
bool sim_use_filter(int64_t P, int64_t H, int flags) {
    if (flags & DLC_SIM_FORCE_F64) return false;
    return P <= 32 && H <= 32768;
}
// ... and, for the all-vs-all matrix, the fixed-point panel stays under 4 GiB (the product kernel's column lanes address
// it with 32-bit offsets: 559 000 patches at H = 2500)
bool sim_use_filter(int64_t N, int64_t P, int64_t H, int flags) {
    return sim_use_filter(P, H, flags) && dlc_gemm::sim_filter_fits(N, P, H);
}

long long sim_chunk(int64_t N, int64_t P, size_t row_bytes, int64_t chunk_bytes) {
    // fp64 Gram row chunk: at most ~8 GiB of the 288 GB (the work is triangular, so every chunk's launch is smaller than
    // the one before and each pays its own last partial round of the chip), at least one frame
    size_t cap = 8ull << 30;
    if (chunk_bytes >= (1ll << 16)) cap = (size_t)chunk_bytes;     // the caller's bound (tests: several chunks at small sizes)
    long long cf = (long long)(cap / (row_bytes * (size_t)P));
    if (cf < 1) cf = 1;
    if (cf >= N - 1) {
        cf = N > 1 ? N - 1 : 1;                  // everything at once (the last frame has no later frame to pair with)
    } else {
        // whole 256-row tiles of the Gram GEMM where the chunk allows: cf * P a multiple of 256
        for (long long q = 256; q >= 8; q /= 2)
            if (cf >= q && (q * P) % 256 == 0) { cf = cf / q * q; break; }
    }
    if (cf > N) cf = N;
    return cf;
}

SimWs sim_ws(int64_t N, int64_t P, int64_t H, int flags, int64_t chunk_bytes) {
    SimWs w;
    size_t o = 0;
    w.nrm2 = o; o += dlc::align_up((size_t)N * P * 8, 256);
    w.proj = o; o += dlc::align_up((size_t)N * P * 16, 256);      // double-double projections
    const bool filter = sim_use_filter(N, P, H, flags);
    const size_t row_bytes = (size_t)N * P * 8;
    w.chunk_frames = sim_chunk(N, P, row_bytes, chunk_bytes);
    // (the filter's workspace keeps a small fp64 Gram region -- up to 64 frames' rows: a dataset with a NaN / infinity in
    // it takes the fp64 route after all, in as many chunks as that needs and without the transposed copy)
    const size_t small = (size_t)(N < 64 ? N : 64) * P * row_bytes;
    w.gram_bytes = filter ? (small < (size_t)w.chunk_frames * P * row_bytes ? small : (size_t)w.chunk_frames * P * row_bytes)
                          : (size_t)w.chunk_frames * P * row_bytes;
    w.gram = o; o += dlc::align_up(w.gram_bytes, 256);
    // both forms: the range / flag / program-length words, NumPy's pairwise-summation program for rows of H elements and
    // the rows' content hashes (the direct evaluation of arg-mins that products of the descriptors cannot decide)
    w.keys = o; o += 256;
    const size_t prog_bytes = dlc_gemm::sim_pairwise_program_bytes(H);
    w.prog = o; o += prog_bytes > 8192 ? prog_bytes : 8192;
    w.rowhash = o; o += dlc::align_up((size_t)N * P * 16, 256);
    w.nu2 = w.qx = w.nbp = w.abi = w.acand = w.blk = w.cc = w.range = 0;
    w.nfp = 0;
    if (filter) {
        // the fixed-point panel (row operands as it lies, column operands gathered from it in units of whole frames), |u|^2, and the product kernel's
        // verdicts: nearest patch + undecided set per (row patch, later frame) -- 5 bytes where r02 kept 120 of products
        w.nfp = dlc_gemm::sim_col_frames(N, P);
        w.nu2 = o; o += dlc::align_up((size_t)N * P * 8, 256);
        w.qx = o; o += dlc::align_up(dlc_gemm::sim_filter_panel_bytes(N * P, H), 256);
        w.nbp = o; o += dlc::align_up((size_t)dlc_gemm::sim_col_rows(N, P) * 4, 256);
        const size_t rp = (size_t)dlc_gemm::sim_argmin_pitch(N, P);
        w.abi = o; o += dlc::align_up(rp * w.nfp, 256);
        w.acand = o; o += dlc::align_up(rp * w.nfp * 4, 256);
        w.blk = o; o += dlc::align_up(dlc_gemm::gram_blocks_bytes(N, P), 256);
        w.cc = o; o += dlc::align_up((size_t)H * 8, 256);                      // the columns' centres
        w.range = o; o += dlc::align_up(dlc_gemm::sim_range_words(H) * 8, 256);  // their extremes, when the caller brings none
        w.desc_t = o;
    } else {
        // the descriptors transposed [H, N*P] (one extra pass over them): the fp64 Gram blocks then read their B operand as
        // [K,N] -- 1 KiB contiguous per k-row and tile instead of 128 scattered 128-byte row segments (an even N*P keeps
        // the rows 16-byte aligned for the LDS-DMA kernel; an odd one falls back to the [N,K] form)
        w.desc_t = o;
        if (((N * P) & 1) == 0) o += dlc::align_up((size_t)N * P * H * 8, 256);
    }
    w.total = o;
    return w;
}

}  // namespace

extern "C" int dlc_cnnvtl_distance_matrix(dlc_ctx* ctx, const int8_t* desc, int64_t N, int64_t D, int64_t ldd,
                                          int64_t* out, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!desc || !out || N < 1 || D < 1 || ldd < D) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "distance_matrix: bad argument");
    if (D > (1ll << 28)) return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "distance_matrix: D too large for int32 accumulation");
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    const unsigned t = (unsigned)dlc::cdiv(N, DT);
    if (t > 65535) return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "distance_matrix: N too large");
    // descriptor chunks: enough workgroups for ~3 per CU, each at least 4 steps of 64 bytes
    const long long pairs = (long long)t * (t + 1) / 2;
    long long z = dlc::cdiv((int64_t)768, pairs);
    const long long steps = dlc::cdiv(D, DCH);
    if (z > steps / 4) z = steps / 4;
    if (z < 1) z = 1;
    if (z > 64) z = 64;
    const long long kchunk = dlc::cdiv(steps, z) * DCH;
    z = dlc::cdiv(D, kchunk);
    DLC_HIP_CHECK(ctx, hipMemsetAsync(out, 0, (size_t)N * (size_t)N * 8, (hipStream_t)stream));
    const bool words = (((uintptr_t)desc) & 3) == 0 && (ldd & 3) == 0;
    if (words)
        hipLaunchKernelGGL(distance_matrix_kernel<true>, dim3(t, t, (unsigned)z), dim3(256), 0, (hipStream_t)stream, desc,
                           (long long)N, (long long)D, (long long)ldd, kchunk, (unsigned long long*)out);
    else
        hipLaunchKernelGGL(distance_matrix_kernel<false>, dim3(t, t, (unsigned)z), dim3(256), 0, (hipStream_t)stream, desc,
                           (long long)N, (long long)D, (long long)ldd, kchunk, (unsigned long long*)out);
    DLC_LAUNCH_CHECK(ctx, "distance_matrix_kernel");
    return DLC_OK;
}

extern "C" size_t dlc_sdav_similarity_workspace_bytes(int64_t N, int64_t P, int64_t H, int flags, int64_t chunk_bytes) {
    if (N < 1 || P < 1 || H < 1) return 0;
    return sim_ws(N, P, H, flags, chunk_bytes).total;
}

extern "C" size_t dlc_sdav_range_words(int64_t H) { return H < 1 ? 0 : dlc_gemm::sim_range_words(H); }

extern "C" int dlc_sdav_distinctive_score(dlc_ctx* ctx, const double* dataset, int64_t rows, int64_t H, double mu,
                                         double sigma, double* score, uint64_t* range, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!dataset || !score || rows < 1 || H < 1 || H > 0x7fffffff)
        return dlc::fail(ctx, DLC_ERR_BAD_ARG, "distinctive_score: bad argument");
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    if (range) hipLaunchKernelGGL(range_init_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)range);
    if ((H & 1) == 0 && ((uintptr_t)dataset & 15) == 0 && (long long)DSD_RB * H * 8 < 0x7fffffffll) {
        constexpr int C = DLC_DSD_COLS, lds = DSD_NB * DSD_RB * C * 8;
        if (!(ctx->func_attr_set & (1ull << DLC_ATTR_DS_DMA))) {
            DLC_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)distinctive_score_dma_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            ctx->func_attr_set |= 1ull << DLC_ATTR_DS_DMA;
        }
        hipLaunchKernelGGL(distinctive_score_dma_kernel<C>, dim3((unsigned)(dlc::cdiv(dlc::cdiv(H, (int64_t)C), (int64_t)8) * 8)), dim3(256),
                           lds, (hipStream_t)stream, dataset, (long long)rows, (int)H, mu, sigma, score, (unsigned long long*)range);
        DLC_LAUNCH_CHECK(ctx, "distinctive_score_dma_kernel");
        return DLC_OK;
    }
    hipLaunchKernelGGL(distinctive_score_kernel, dim3((unsigned)(dlc::cdiv(dlc::cdiv(H, (int64_t)DS_COLS), (int64_t)8) * 8)), dim3(256), 0, (hipStream_t)stream,
                       dataset, (long long)rows, (int)H, mu, sigma, score, (unsigned long long*)range);
    DLC_LAUNCH_CHECK(ctx, "distinctive_score_kernel");
    return DLC_OK;
}

extern "C" int dlc_sdav_similarity_matrix(dlc_ctx* ctx, const double* desc, int64_t N, int64_t P, int64_t H,
                                          const double* score, double a, double b, double* out_f64, int64_t* out_i64,
                                          int flags, int64_t chunk_bytes, const uint64_t* range, int64_t* stats,
                                          uint8_t* direct_pairs, void* workspace, size_t workspace_bytes, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!desc || !score || !out_f64 || N < 1 || P < 1 || H < 1 || (flags & ~(DLC_SIM_FORCE_F64 | DLC_SIM_NO_HOST_SYNC)))
        return dlc::fail(ctx, DLC_ERR_BAD_ARG, "similarity_matrix: bad argument");
    if (P > 64) return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "similarity_matrix: P=%lld patches per frame > 64", (long long)P);
    if (H > 0x7fffffff || N > 65535) return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "similarity_matrix: N or H too large");
    const SimWs w = sim_ws(N, P, H, flags, chunk_bytes);
    if (!workspace || workspace_bytes < w.total)
        return dlc::fail(ctx, DLC_ERR_WORKSPACE, "similarity_matrix: workspace %zu < %zu bytes", workspace_bytes, w.total);
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    double* nrm2 = (double*)(ws + w.nrm2);
    double* proj = (double*)(ws + w.proj);
    double* gram = (double*)(ws + w.gram);
    const long long rows = N * P;
    bool filter = sim_use_filter(N, P, H, flags);
    if (!filter && stats) DLC_HIP_CHECK(ctx, hipMemsetAsync(stats, 0, 16, st));
    if (direct_pairs) DLC_HIP_CHECK(ctx, hipMemsetAsync(direct_pairs, 0, (size_t)N * (size_t)N, st));

    hipLaunchKernelGGL(fill_diag_kernel, dim3((unsigned)dlc::cdiv(N, 256)), dim3(256), 0, st, (long long)N, out_f64,
                       (long long*)out_i64);
    DLC_LAUNCH_CHECK(ctx, "fill_diag_kernel");

    if (filter) {
        unsigned long long* keys = (unsigned long long*)(ws + w.keys);
        double* nu2 = (double*)(ws + w.nu2);
        char* qx = ws + w.qx;
        int* nbp = (int*)(ws + w.nbp);
        unsigned char* abi = (unsigned char*)(ws + w.abi);
        unsigned* acand = (unsigned*)(ws + w.acand);
        int2* prog = (int2*)(ws + w.prog);
        unsigned long long* rowhash = (unsigned long long*)(ws + w.rowhash);
        int rc = dlc_gemm::sim_filter_prepare(ctx, desc, N, P, H, score, keys, (double*)(ws + w.cc),
                                              (unsigned long long*)(ws + w.range), qx, nbp, nu2, proj, rowhash, prog,
                                              (const unsigned long long*)range, st);
        if (rc != DLC_OK) return rc;
        // did the range pass meet a NaN or an infinity?  (Their distances are NaN in the reference too, np.argmin then
        // takes the first of them: the fp64 kernels reproduce that, a fixed-point fraction cannot.)  The ONE host read of
        // this library's stream-ordered calls -- unless the caller rules it out (DLC_SIM_NO_HOST_SYNC): the filter form's
        // kernels then leave at once on such data and the matrix comes back as NaN / INT64_MIN with stats[1] = 1.
        // The read does not hold the stream up: the word is copied to page-locked memory behind the quantisation pass, the
        // filter form's kernels are enqueued right behind it (on such data they leave at once), and the host waits for
        // the COPY's event only -- while the product kernel runs -- before it decides whether the fp64 form has to follow.
        const bool no_sync = (flags & DLC_SIM_NO_HOST_SYNC) != 0;
        if (!no_sync) {
            // would the filter decide?  (sim_sample_kernel: bit 1 of the same word when it would not)
            if (N > 1) {
                hipLaunchKernelGGL(sim_sample_kernel, dim3(SIM_SAMPLES), dim3(256), 0, st, desc, (const unsigned long long*)rowhash, keys,
                                   (long long)N, (int)P, (int)H);
                DLC_LAUNCH_CHECK(ctx, "sim_sample_kernel");
            }
            DLC_HIP_CHECK(ctx, hipMemcpyAsync(ctx->host_flag, keys + 2, 8, hipMemcpyDeviceToHost, st));
            DLC_HIP_CHECK(ctx, hipEventRecord(ctx->ev_flag, st));
        }
        if (N > 1) {
            // all frames in ONE launch: the product kernel keeps its 31 890 x 31 890 products on the chip and emits
            // the arg-mins (r02: row chunks of an 8 GiB int32 block)
            rc = dlc_gemm::gram_argmin_i8(ctx, N, P, H, qx, nbp, keys, abi, acand, ws + w.blk, st);
            if (rc != DLC_OK) return rc;
            hipLaunchKernelGGL(pair_score_amin_kernel, dim3(PS_GX, (unsigned)(N - 1)), dim3(256), PF_STACK_BYTES, st, desc,
                               (const unsigned char*)abi, (const unsigned*)acand, (long long)dlc_gemm::sim_argmin_pitch(N, P), proj, score, keys,
                               (long long)N, (int)P, (int)H, a, b, out_f64, (long long*)out_i64, prog, rowhash, direct_pairs);
            DLC_LAUNCH_CHECK(ctx, "pair_score_amin_kernel");
        }
        if (stats || no_sync) {
            hipLaunchKernelGGL(sim_finish_kernel, dim3(no_sync ? 256u : 1u), dim3(256), 0, st, keys, (long long*)stats,
                               (long long)N * N, out_f64, (long long*)out_i64, no_sync ? 1 : 0);
            DLC_LAUNCH_CHECK(ctx, "sim_finish_kernel");
        }
        if (no_sync) return DLC_OK;
        DLC_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev_flag));
        const unsigned long long why = *(volatile unsigned long long*)ctx->host_flag;
        if (!why) return DLC_OK;
        filter = false;                 // NaN / infinity in the data, or distances the filter's window swallows: the fp64 form after all
        if (stats) hipLaunchKernelGGL(sim_stats_kernel, dim3(1), dim3(64), 0, st, (long long*)stats, (long long)why);
    }

    // the fp64 Gram route
    {
        const int rc = dlc_gemm::sim_row_sums(ctx, desc, rows, H, score, nrm2, proj, (unsigned long long*)(ws + w.rowhash),
                                              ws + w.prog, (unsigned long long*)(ws + w.keys) + 5, st);
        if (rc != DLC_OK) return rc;
    }
    long long chunk_frames = w.chunk_frames;
    bool use_t = (rows & 1) == 0;
    if (sim_use_filter(N, P, H, flags)) {        // the filter's workspace: no transposed copy, the chunk that fits its Gram region
        use_t = false;
        chunk_frames = (long long)(w.gram_bytes / ((size_t)P * rows * 8));
    }
    double* desc_t = (double*)(ws + w.desc_t);
    if (use_t) {
        hipLaunchKernelGGL(transpose_f64_kernel, dim3((unsigned)dlc::cdiv(rows, 32), (unsigned)dlc::cdiv(H, 32)), dim3(256), 0, st,
                           desc, rows, (long long)H, desc_t);
        DLC_LAUNCH_CHECK(ctx, "transpose_f64_kernel");
    }
    for (long long i_lo = 0; i_lo + 1 < N; i_lo += chunk_frames) {
        long long i_hi = i_lo + chunk_frames;
        if (i_hi > N - 1) i_hi = N - 1;          // the last frame has no j > i
        if (i_hi <= i_lo) break;
        // columns: frames j > i_lo, i.e. from frame i_lo+1 on
        const long long col0 = (i_lo + 1) * P;
        const long long ncols = rows - col0;
        const long long mrows = (i_hi - i_lo) * P;
        // only entries with (row frame < column frame) are read below: tiles under the diagonal are skipped
        int rc = use_t ? dlc_gemm::gram_upper_f64(ctx, DLC_B_KN, mrows, ncols, H, desc + i_lo * P * H, H, desc_t + col0, rows, gram,
                                                  ncols, (int)P, i_lo * P, col0, st)
                       : dlc_gemm::gram_upper_f64(ctx, DLC_B_NK, mrows, ncols, H, desc + i_lo * P * H, H, desc + col0 * H, H, gram,
                                                  ncols, (int)P, i_lo * P, col0, st);
        if (rc != DLC_OK) return rc;
        dim3 grid((unsigned)dlc::cdiv(N, 4), (unsigned)(i_hi - i_lo));
        const size_t tile_lds = PF_STACK_BYTES + ((size_t)P * ((PS_JT * P) | 1) + (size_t)PS_JT * P) * sizeof(double);
        const bool tiled = P <= 32;
        if (tiled) {
            if (tile_lds > 48 * 1024 && !(ctx->func_attr_set & (1ull << DLC_ATTR_PAIR_TILE))) {
                DLC_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)pair_score_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 76 * 1024));
                ctx->func_attr_set |= 1ull << DLC_ATTR_PAIR_TILE;
            }
            hipLaunchKernelGGL(pair_score_tile_kernel, dim3(PS_GX, (unsigned)(i_hi - i_lo)), dim3(256), tile_lds,
                               st, desc, gram, ncols, col0, nrm2, proj, score, (long long)N, (int)P, (int)H, i_lo, i_hi, a, b,
                               out_f64, (long long*)out_i64, (const int2*)(ws + w.prog),
                               (const unsigned long long*)(ws + w.keys) + 5, (const unsigned long long*)(ws + w.rowhash));
        } else
            hipLaunchKernelGGL(pair_score_kernel, grid, dim3(256), 0, st, desc, gram, ncols, col0, nrm2, proj, score,
                               (long long)N, (int)P, (int)H, i_lo, i_hi, a, b, out_f64, (long long*)out_i64,
                               (const int2*)(ws + w.prog), (const unsigned long long*)(ws + w.keys) + 5,
                               (const unsigned long long*)(ws + w.rowhash));
        DLC_LAUNCH_CHECK(ctx, "pair_score_kernel");
    }
    return DLC_OK;
}

// ---- the streaming similarity (include/dlc.h) ----------------------------------------------------------------------
extern "C" size_t dlc_sdav_stream_state_bytes(int64_t capacity, int64_t P, int64_t H) {
    if (capacity < 1 || P < 1 || P > 32 || H < 1 || H > 32768) return 0;
    return stream_ws(capacity, P, H).total;
}

extern "C" int dlc_sdav_stream_init(dlc_ctx* ctx, void* state, size_t state_bytes, int64_t capacity, int64_t P, int64_t H,
                                    double lo, double hi, const double* col_centre, void* stream) {
    int rc = stream_check(ctx, "sdav_stream_init", state, state_bytes, capacity, P, H);
    if (rc != DLC_OK) return rc;
    if (!(hi > lo) || !(hi - lo < INFINITY)) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "sdav_stream_init: need lo < hi, both finite");
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    const StreamWs w = stream_ws(capacity, P, H);
    char* ws = (char*)state;
    // the batched query's strip (gram_argmin_i8_strip) gathers padding columns from a zero group and reads |v|^2 = 0 for them
    const size_t gbytes = 3 * dlc::align_up((size_t)H, (size_t)256) * 16;
    DLC_HIP_CHECK(ctx, hipMemsetAsync(ws + w.panel + (w.panel_groups - 8) * gbytes, 0, 8 * gbytes, (hipStream_t)stream));
    DLC_HIP_CHECK(ctx, hipMemsetAsync(ws + w.nbp, 0, (size_t)dlc_gemm::sim_col_rows(capacity, P) * 4, (hipStream_t)stream));
    return dlc_gemm::sim_stream_init(ctx, (unsigned long long*)(ws + w.keys), (double*)(ws + w.cc), ws + w.prog, H, lo, hi,
                                     col_centre, (hipStream_t)stream);
}

extern "C" int dlc_sdav_stream_append(dlc_ctx* ctx, void* state, size_t state_bytes, int64_t capacity, int64_t P, int64_t H,
                                      const double* desc, int64_t n_old, int64_t n_total, const double* score, void* stream) {
    int rc = stream_check(ctx, "sdav_stream_append", state, state_bytes, capacity, P, H);
    if (rc != DLC_OK) return rc;
    if (!desc || !score || n_old < 0 || n_total < n_old || n_total > capacity)
        return dlc::fail(ctx, DLC_ERR_BAD_ARG, "sdav_stream_append: frames [%lld, %lld) outside the capacity %lld", (long long)n_old,
                         (long long)n_total, (long long)capacity);
    if (n_total == n_old) return DLC_OK;
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    const StreamWs w = stream_ws(capacity, P, H);
    char* ws = (char*)state;
    const int64_t g_first = n_old * P / 16, g_last = dlc::cdiv(n_total * P, (int64_t)16);
    return dlc_gemm::sim_stream_quantise(ctx, desc, n_total * P, H, score, (unsigned long long*)(ws + w.keys),
                                         (const double*)(ws + w.cc), ws + w.panel,
                                         (double*)(ws + w.nu2), (double*)(ws + w.proj), (unsigned long long*)(ws + w.rowhash),
                                         g_first, g_last - g_first, P, (int*)(ws + w.nbp), (hipStream_t)stream);
}

// A batch's workspace: the arg-mins [nq, pitch] | the strip's verdicts abi [frames, pitch] and acand [frames, pitch].
// frames: the whole block columns a batch of nq frames can touch, wherever it starts.
struct BatchWs {
    size_t bi, abi, acand, total;
    long long pitch;
};
constexpr int64_t STRIP_MIN_QUERIES = 8;                      // smaller batches: two query frames per pass over the panel
static BatchWs batch_ws(int64_t capacity, int64_t P, int64_t nq) {
    BatchWs w;
    w.pitch = (long long)dlc::align_up((size_t)capacity * (size_t)P, 256);
    size_t o = 0;
    w.bi = o; o += (size_t)nq * (size_t)w.pitch;
    w.abi = w.acand = o;
    if (nq >= STRIP_MIN_QUERIES) {
        const int64_t frames = dlc_gemm::gram_strip_frames(0, nq - 1, P) + dlc_gemm::gram_strip_frames(0, 0, P);
        w.abi = o; o += (size_t)frames * (size_t)w.pitch;
        w.acand = o; o += (size_t)frames * (size_t)w.pitch * 4;
    }
    w.total = o;
    return w;
}

static int stream_query_impl(dlc_ctx* ctx, const char* what, void* state, size_t state_bytes, int64_t capacity, int64_t P, int64_t H,
                             const double* desc, int64_t f, int64_t nq, const double* score, double a, double b, double* rows_out,
                             int64_t ld_rows, int64_t* stats, unsigned char* bi, int64_t bi_pitch, void* stream,
                             char* batch_base = nullptr, const BatchWs* bw = nullptr, int stage = 0) {
    hipStream_t st = (hipStream_t)stream;
    const StreamWs w = stream_ws(capacity, P, H);
    char* ws = (char*)state;
    unsigned long long* keys = (unsigned long long*)(ws + w.keys);
    const int64_t f_last = f + nq - 1;
    if (f_last == 0) {                                          // (one query, frame 0: nothing older)
        if (stats) DLC_HIP_CHECK(ctx, hipMemsetAsync(stats, 0, 16, st));
        return DLC_OK;
    }
    const long long kp = (long long)dlc::align_up((size_t)H, (size_t)256);
    const long long gpitch = 3 * kp * 16;
    if (bw && nq >= STRIP_MIN_QUERIES && dlc_gemm::sim_filter_fits(capacity, P, H)) {
        // the batch as a strip of the all-vs-all call: ONE launch of the product kernel (columns: the batch's frames, rows:
        // every older patch of the resident panel) + the resolution of its undecided cells + the scores.  At 32 frames
        // against 1062 older ones the two-frames-per-pass form streams the 245 MB panel 16 times (1.07 ms); the strip is
        // 8 of the matrix call's 266 column tiles.
        unsigned char* abi = (unsigned char*)(batch_base + bw->abi);
        unsigned* acand = (unsigned*)(batch_base + bw->acand);
        int64_t fj_base = 0;
        // (stage 2 -- the products were launched by an earlier stage-1 call on the same workspace: only the strip's base)
        int rc = dlc_gemm::gram_argmin_i8_strip(ctx, f, f_last, P, H, (const char*)(ws + w.panel), w.zrow, (const int*)(ws + w.nbp), keys,
                                                abi, acand, bw->pitch, &fj_base, st, /*launch=*/stage != 2);
        if (rc != DLC_OK) return rc;
        if (stage == 1) return DLC_OK;
        hipLaunchKernelGGL(strip_resolve_kernel, dim3((unsigned)dlc::cdiv(f_last * P, (int64_t)256), (unsigned)nq), dim3(256), PF_STACK_BYTES,
                           st, desc, (const unsigned char*)abi, (const unsigned*)acand, (long long)bw->pitch, (long long)fj_base,
                           (const unsigned long long*)(ws + w.rowhash), keys, (long long)f, (int)P, (int)H, (const int2*)(ws + w.prog), bi,
                           (long long)bi_pitch);
        DLC_LAUNCH_CHECK(ctx, "strip_resolve_kernel");
        hipLaunchKernelGGL(stream_score_kernel, dim3((unsigned)dlc::cdiv(f_last, (int64_t)8), (unsigned)nq), dim3(256), 0, st, desc,
                           (const double*)(ws + w.proj), score, (const unsigned char*)bi, (long long)bi_pitch, keys, (long long)f, (int)P,
                           (int)H, a, b, rows_out, (long long)ld_rows, (long long*)stats);
        DLC_LAUNCH_CHECK(ctx, what);
        return DLC_OK;
    }
    const long long groups = dlc::cdiv(f_last * P, (int64_t)16);          // of the newest query; an older one's extra blocks leave at once
    const dim3 agrid((unsigned)dlc::cdiv(groups, (long long)(4 * SR_G)), (unsigned)(nq == 1 ? 1 : dlc::cdiv(nq, (int64_t)2)));
    if (nq == 1)
        hipLaunchKernelGGL(stream_argmin_kernel<1>, agrid, dim3(256), 0, st, (const char*)(ws + w.panel), gpitch, (int)(kp / 64), desc,
                           (const double*)(ws + w.nu2), (const unsigned long long*)(ws + w.rowhash), keys, (long long)f,
                           (long long)(f + nq), (int)P, (int)H, bi, (long long)bi_pitch, (const int2*)(ws + w.prog));
    else                                                        // a batch: two query frames per pass over the panel
        hipLaunchKernelGGL(stream_argmin_kernel<2>, agrid, dim3(256), 0, st, (const char*)(ws + w.panel), gpitch, (int)(kp / 64), desc,
                           (const double*)(ws + w.nu2), (const unsigned long long*)(ws + w.rowhash), keys, (long long)f,
                           (long long)(f + nq), (int)P, (int)H, bi, (long long)bi_pitch, (const int2*)(ws + w.prog));
    DLC_LAUNCH_CHECK(ctx, "stream_argmin_kernel");
    hipLaunchKernelGGL(stream_score_kernel, dim3((unsigned)dlc::cdiv(f_last, (int64_t)8), (unsigned)nq), dim3(256), 0, st, desc,
                       (const double*)(ws + w.proj), score, (const unsigned char*)bi, (long long)bi_pitch, keys, (long long)f, (int)P,
                       (int)H, a, b, rows_out, (long long)ld_rows, (long long*)stats);
    DLC_LAUNCH_CHECK(ctx, what);
    return DLC_OK;
}

extern "C" int dlc_sdav_stream_query(dlc_ctx* ctx, void* state, size_t state_bytes, int64_t capacity, int64_t P, int64_t H,
                                     const double* desc, int64_t f, const double* score, double a, double b, double* row_out,
                                     int64_t* stats, void* stream) {
    int rc = stream_check(ctx, "sdav_stream_query", state, state_bytes, capacity, P, H);
    if (rc != DLC_OK) return rc;
    if (!desc || !score || (!row_out && f > 0) || f < 0 || f >= capacity)
        return dlc::fail(ctx, DLC_ERR_BAD_ARG, "sdav_stream_query: bad argument");
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    const StreamWs w = stream_ws(capacity, P, H);
    return stream_query_impl(ctx, "stream_score_kernel", state, state_bytes, capacity, P, H, desc, f, 1, score, a, b, row_out, 0, stats,
                             (unsigned char*)((char*)state + w.bi), 0, stream);
}

extern "C" size_t dlc_sdav_stream_query_batch_workspace_bytes(int64_t capacity, int64_t P, int64_t n_queries) {
    if (capacity < 1 || P < 1 || P > 32 || n_queries < 1) return 0;
    return batch_ws(capacity, P, n_queries).total;
}

extern "C" int dlc_sdav_stream_query_batch(dlc_ctx* ctx, void* state, size_t state_bytes, int64_t capacity, int64_t P, int64_t H,
                                           const double* desc, int64_t f_first, int64_t n_queries, const double* score, double a,
                                           double b, double* rows_out, int64_t ld_rows, int64_t* stats, void* workspace,
                                           size_t workspace_bytes, void* stream) {
    int rc = stream_check(ctx, "sdav_stream_query_batch", state, state_bytes, capacity, P, H);
    if (rc != DLC_OK) return rc;
    if (!desc || !score || !rows_out || f_first < 0 || n_queries < 1 || n_queries > 65535 || f_first + n_queries > capacity ||
        ld_rows < f_first + n_queries - 1)
        return dlc::fail(ctx, DLC_ERR_BAD_ARG, "sdav_stream_query_batch: bad argument (frames [%lld, %lld) of capacity %lld, ld_rows %lld)",
                         (long long)f_first, (long long)(f_first + n_queries), (long long)capacity, (long long)ld_rows);
    const size_t need = dlc_sdav_stream_query_batch_workspace_bytes(capacity, P, n_queries);
    if (!workspace || workspace_bytes < need)
        return dlc::fail(ctx, DLC_ERR_WORKSPACE, "sdav_stream_query_batch: workspace %zu < %zu bytes", workspace_bytes, need);
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    if ((uintptr_t)workspace & 255) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "sdav_stream_query_batch: workspace must be 256-byte aligned");
    const BatchWs bw = batch_ws(capacity, P, n_queries);
    return stream_query_impl(ctx, "stream_score_kernel", state, state_bytes, capacity, P, H, desc, f_first, n_queries, score, a, b, rows_out,
                             ld_rows, stats, (unsigned char*)workspace + bw.bi, bw.pitch, stream, (char*)workspace, &bw);
}

extern "C" int dlc_sdav_stream_query_batch_staged(dlc_ctx* ctx, void* state, size_t state_bytes, int64_t capacity, int64_t P, int64_t H,
                                                  const double* desc, int64_t f_first, int64_t n_queries, const double* score, double a,
                                                  double b, double* rows_out, int64_t ld_rows, int64_t* stats, void* workspace,
                                                  size_t workspace_bytes, int stage, void* stream) {
    int rc = stream_check(ctx, "sdav_stream_query_batch_staged", state, state_bytes, capacity, P, H);
    if (rc != DLC_OK) return rc;
    if (stage != 1 && stage != 2) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "sdav_stream_query_batch_staged: stage must be 1 or 2");
    if (!desc || !score || !rows_out || f_first < 0 || n_queries < 1 || n_queries > 65535 || f_first + n_queries > capacity ||
        ld_rows < f_first + n_queries - 1)
        return dlc::fail(ctx, DLC_ERR_BAD_ARG, "sdav_stream_query_batch_staged: bad argument (frames [%lld, %lld) of capacity %lld, ld_rows %lld)",
                         (long long)f_first, (long long)(f_first + n_queries), (long long)capacity, (long long)ld_rows);
    if (n_queries < STRIP_MIN_QUERIES || !dlc_gemm::sim_filter_fits(capacity, P, H))
        return dlc::fail(ctx, DLC_ERR_UNSUPPORTED, "sdav_stream_query_batch_staged: only the strip form (batches of %lld frames and more) "
                         "has two stages", (long long)STRIP_MIN_QUERIES);
    const size_t need = dlc_sdav_stream_query_batch_workspace_bytes(capacity, P, n_queries);
    if (!workspace || workspace_bytes < need)
        return dlc::fail(ctx, DLC_ERR_WORKSPACE, "sdav_stream_query_batch_staged: workspace %zu < %zu bytes", workspace_bytes, need);
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    if ((uintptr_t)workspace & 255) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "sdav_stream_query_batch_staged: workspace must be 256-byte aligned");
    if (f_first + n_queries - 1 == 0) return DLC_OK;              // (frame 0 alone cannot be a strip; kept for symmetry)
    const BatchWs bw = batch_ws(capacity, P, n_queries);
    return stream_query_impl(ctx, "stream_score_kernel", state, state_bytes, capacity, P, H, desc, f_first, n_queries, score, a, b, rows_out,
                             ld_rows, stats, (unsigned char*)workspace + bw.bi, bw.pitch, stream, (char*)workspace, &bw, stage);
}

extern "C" int dlc_topk_rows_f64(dlc_ctx* ctx, const double* scores, int64_t rows, int64_t ld, int64_t limit0, int64_t limit_step,
                                 int k, double* out_scores, int64_t* out_idx, const int64_t* poison, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!scores || !out_scores || !out_idx || rows < 1 || ld < 1 || k < 1 || k > DLC_MAX_K || rows > 0x7fffffffll)
        return dlc::fail(ctx, DLC_ERR_BAD_ARG, "topk_rows_f64: bad argument");
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    hipLaunchKernelGGL(topk_rows_f64_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, scores, (long long)ld,
                       (long long)limit0, (long long)limit_step, k, out_scores, (long long*)out_idx, (const long long*)poison);
    DLC_LAUNCH_CHECK(ctx, "topk_rows_f64_kernel");
    return DLC_OK;
}
