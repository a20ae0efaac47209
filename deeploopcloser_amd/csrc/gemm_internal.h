// Shared between the dense-GEMM translation units (gemm_dense.hip, gemm_dma_f64.hip).
#pragma once
#include "dlc_internal.h"

namespace dlc_gemm {

// Implicit im2col (tf.layers.conv2d on NHWC, src/cnn_vtl/network/cnn_vtl.py:33-93): row m of the
// A operand is output pixel (img, oy, ox), column k is (ky, kx, c) with c fastest; A then points
// at the NHWC input and lda is unused.
struct ConvGeom {
    int H, W, C, KW, stride, pad_t, pad_l, OH, OW;
    // optional: [images, 2] ordered keys (dlc_f64_key) into which the minimum / maximum of every image's outputs is
    // folded (cnn_vtl.py:110-112 takes them over the whole descriptor of a frame; dlc_cnnvtl_frame_minmax_init)
    unsigned long long* mm_keys;
};

// Triangular skip of a launch (the Gram blocks of the SDAV similarity, match_ref.hip): rows / columns are
// patches of frames of `p` patches, row r is global patch row0 + r, column c global patch col0 + c; only
// (row frame < column frame) entries are ever read, so a tile that holds none is not computed.  p = 0: off.
struct TriSkip {
    int p;
    long long row0, col0;
};

// The LDS-DMA form of the fp64 GEMM (gemm_dma_f64.hip).  Returns DLC_OK after launching, or 1 when the shape /
// alignment is not one it handles (the caller then takes the register-staged kernel of gemm_dense.hip).
// Kb (0 = K): the reduction length of the B operand when A has been zero-padded past it (an odd K such as SDAV's
// 1681 input columns, copied into rows of 1696): B's missing k-rows read as zeros.
int launch_dma_f64(dlc_ctx* ctx, int blayout, int act, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda,
                   const double* B, int64_t ldb, const double* bias, double* C, int64_t ldc, hipStream_t st,
                   const ConvGeom* cv, const TriSkip* tri, int64_t Kb = 0, double alpha = 0.0);
int gemm_axpy_dma_f64(dlc_ctx* ctx, int blayout, double alpha, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda,
                      const double* B, int64_t ldb, double* C, int64_t ldc, hipStream_t st);
// split-K on the kernel's 64-row tiles: chunks of kchunk (a multiple of 16) into partials[chunks][M][N] (gemm_dma_f64.hip)
int gemm_dma_f64_splitk(dlc_ctx* ctx, int blayout, int64_t M, int64_t N, int64_t K, int64_t Kb, const double* A, int64_t lda,
                        const double* B, int64_t ldb, double* partials, int64_t kchunk, hipStream_t st, bool dry);

// A zero-padded by the caller to lda = Kpad columns (columns K .. Kpad-1 are zeros): act(A[:, :K] . B + bias) with the
// LDS-DMA kernel when it applies (it then walks Kpad), else the register-staged kernel on the first K columns.
int gemm_bias_act_padded_f64(dlc_ctx* ctx, int act, int64_t M, int64_t N, int64_t K, int64_t Kpad, const double* A,
                             const double* B, int64_t ldb, const double* bias, double* C, int64_t ldc, hipStream_t st);

// The SDAV similarity's arg-min filter (gram_i8.hip): descriptors as three signed fixed-point digits of their offset from
// the column's centre, their exact integer products, and the bound of what the rounding misses.
constexpr int DLC_SIM_KEYS = 8;        // words of a call's `keys` (gram_i8.hip: sim_filter_prepare)
size_t sim_filter_panel_bytes(int64_t rows, int64_t H);
size_t sim_range_words(int64_t H);
int sim_frames_per_unit(int64_t P);
int64_t sim_col_rows(int64_t N, int64_t P);
int64_t sim_col_frames(int64_t N, int64_t P);
int64_t sim_argmin_pitch(int64_t N, int64_t P);
bool sim_filter_fits(int64_t N, int64_t P, int64_t H);
int sim_filter_prepare(dlc_ctx* ctx, const double* desc, int64_t N, int64_t P, int64_t H, const double* score,
                       unsigned long long* keys, double* cc, unsigned long long* ws_range, char* X, int* nbp, double* nu2,
                       double* proj, unsigned long long* rowhash, void* prog, const unsigned long long* range, hipStream_t st);
size_t sim_pairwise_program_bytes(int64_t H);
int sim_row_sums(dlc_ctx* ctx, const double* desc, int64_t rows, int64_t H, const double* score, double* nrm2, double* proj,
                 unsigned long long* rowhash, void* prog, unsigned long long* prog_len, hipStream_t st);
// the streaming form: a resident append-only panel quantised over a fixed range (match_ref.hip: dlc_sdav_stream_*)
int sim_stream_init(dlc_ctx* ctx, unsigned long long* keys, double* cc, void* prog, int64_t H, double lo, double hi,
                    const double* centre, hipStream_t st);
int sim_stream_quantise(dlc_ctx* ctx, const double* desc, int64_t rows_total, int64_t H, const double* score,
                        unsigned long long* keys, const double* cc, char* X, double* nu2, double* proj,
                        unsigned long long* rowhash, int64_t g_first, int64_t g_count, int64_t P, int* nbp, hipStream_t st);
size_t sim_stream_panel_bytes(int64_t rows, int64_t H);
int64_t gram_strip_frames(int64_t f_first, int64_t f_last, int64_t P);
int gram_argmin_i8_strip(dlc_ctx* ctx, int64_t f_first, int64_t f_last, int64_t P, int64_t H, const char* X, int64_t zrow,
                         const int* nbp, const unsigned long long* keys, unsigned char* abi, unsigned* acand, int64_t rp,
                         int64_t* fj_base_out, hipStream_t st, bool launch = true);
int gram_argmin_i8(dlc_ctx* ctx, int64_t N, int64_t P, int64_t H, const char* X, const int* nbp,
                   const unsigned long long* keys, unsigned char* abi, unsigned* acand, void* blocks, hipStream_t st);
size_t gram_blocks_bytes(int64_t N, int64_t P);

// How far apart two d2 = |v_b|^2 - 2 v_a . v_b of the filter (units of 2^-15, gram_i8.hip's header) must be before their
// order is the order of the true distances: twice the bound of one d2's error, + 1e-8 for the fp64 roundings of v and of
// |v|^2, rounded up, + 2.  keys[3]: the ordered key of the largest row sum of |v|.
__device__ __forceinline__ long long dlc_sim_window(const unsigned long long* __restrict__ keys, int H) {
    const double sv = dlc_f64_unkey(keys[3]);
    const double ed = 0x1p-23 * sv + (double)H * (0x1p-24 + 0x1p-33 + 0x1p-46) + 1.004 * 0x1p-15 + 0x1p-16;
    return (long long)ceil((2.0 * ed + 1e-8) * 32768.0) + 2;
}

}  // namespace dlc_gemm

namespace dlc_cnn {
// Folds min / max of x[img, per_img] (fp64, contiguous) into keys[img, 2] (cnnvtl.hip).
int fold_minmax_f64(dlc_ctx* ctx, const double* x, int64_t n_img, int64_t per_img, unsigned long long* keys, hipStream_t st);
}
