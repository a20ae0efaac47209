// Shared between the dense-GEMM translation units (gemm_dense.hip, gemm_dma_f64.hip).
#pragma once
#include "dlc_internal.h"

namespace dlc_gemm {

// Implicit im2col (tf.layers.conv2d on NHWC, src/cnn_vtl/network/cnn_vtl.py:33-93): row m of the
// A operand is output pixel (img, oy, ox), column k is (ky, kx, c) with c fastest; A then points
// at the NHWC input and lda is unused.
struct ConvGeom {
    int H, W, C, KW, stride, pad_t, pad_l, OH, OW;
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
int launch_dma_f64(dlc_ctx* ctx, int blayout, int act, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda,
                   const double* B, int64_t ldb, const double* bias, double* C, int64_t ldc, hipStream_t st,
                   const ConvGeom* cv, const TriSkip* tri);

}  // namespace dlc_gemm
