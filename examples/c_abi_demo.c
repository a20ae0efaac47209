/* Plain-C use of the C ABI (include/dlc.h): no Python, no torch.
 *
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ examples/c_abi_demo.c -Iinclude -I/opt/rocm/include \
 *       -Ldeeploopcloser_amd -ldlc_hip -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/deeploopcloser_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/c_abi_demo && /tmp/c_abi_demo
 *
 * Builds a small key-frame database on the device (float rows -> L2-normalised bf16 with
 * dlc_l2_normalize_rows), asks for the top-3 matches of two of its own rows and checks that
 * each row finds itself first with score ~1.  Device memory comes from the HIP runtime; the
 * library itself never allocates.
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "dlc.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_DLC(x) do { int rc_ = (x); if (rc_ != DLC_OK) { fprintf(stderr, "%s: %s (%s)\n", #x, dlc_status_string(rc_), dlc_last_error(ctx)); return 3; } } while (0)

int main(void) {
    enum { N = 1000, D = 256, Q = 2, K = 3 };
    dlc_ctx* ctx = NULL;
    if (dlc_create(0, &ctx) != DLC_OK) { fprintf(stderr, "no MI355X visible\n"); return 1; }
    if (dlc_abi_version() != DLC_ABI_VERSION) { fprintf(stderr, "header / library ABI mismatch\n"); return 1; }

    float* host = (float*)malloc(sizeof(float) * N * D);
    unsigned s = 12345u;
    for (long i = 0; i < (long)N * D; ++i) { s = s * 1664525u + 1013904223u; host[i] = (float)(s >> 8) / 16777216.0f - 0.5f; }

    float* rows_f32; void* rows; float* scores; int64_t* idx; void* ws;
    CHECK_HIP(hipMalloc((void**)&rows_f32, sizeof(float) * N * D));
    CHECK_HIP(hipMalloc(&rows, 2 * N * D));
    CHECK_HIP(hipMalloc((void**)&scores, sizeof(float) * Q * K));
    CHECK_HIP(hipMalloc((void**)&idx, sizeof(int64_t) * Q * K));
    CHECK_HIP(hipMemcpy(rows_f32, host, sizeof(float) * N * D, hipMemcpyHostToDevice));
    CHECK_DLC(dlc_l2_normalize_rows(ctx, DLC_F32, rows_f32, N, D, D, 0, DLC_BF16, rows, D, NULL));

    const size_t need = dlc_cosine_topk_workspace_bytes(Q, N, D, K);
    CHECK_HIP(hipMalloc(&ws, need));
    const char* queries = (const char*)rows + (size_t)500 * D * 2;        /* rows 500 and 501 as the queries */
    /* fp64 scores and the per-query certificate are optional outputs: NULL, NULL; tau_scale NULL: every row is
       dlc_l2_normalize_rows' output (rows from elsewhere: dlc_max_row_norm + dlc_cosine_tau_scale) */
    CHECK_DLC(dlc_cosine_topk(ctx, DLC_BF16, queries, Q, D, rows, N, D, D, K, 0, scores, NULL, idx, NULL, NULL, ws, need, NULL));
    CHECK_HIP(hipDeviceSynchronize());

    float hs[Q * K]; int64_t hi[Q * K];
    CHECK_HIP(hipMemcpy(hs, scores, sizeof(hs), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(hi, idx, sizeof(hi), hipMemcpyDeviceToHost));
    int ok = 1;
    for (int q = 0; q < Q; ++q) {
        printf("query %d:", q);
        for (int j = 0; j < K; ++j) printf("  (%lld, %.4f)", (long long)hi[q * K + j], hs[q * K + j]);
        printf("\n");
        ok = ok && hi[q * K] == 500 + q && hs[q * K] > 0.99f && hs[q * K] < 1.01f && hs[q * K + 1] < 0.5f;
    }
    hipFree(ws); hipFree(idx); hipFree(scores); hipFree(rows); hipFree(rows_f32); free(host);
    dlc_destroy(ctx);
    printf(ok ? "c_abi_demo ok\n" : "c_abi_demo FAILED\n");
    return ok ? 0 : 4;
}
