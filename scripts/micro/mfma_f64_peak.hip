// Microbenchmark (GPU box only): what the fp64 matrix pipe sustains with nothing else in the loop --
// v_mfma_f64_16x16x4_f64 back to back on 16 independent accumulators, operands in registers, one or two waves per
// SIMD, every CU.  Sets the ceiling the dense fp64 GEMM (gemm_dense.hip) is priced against.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_f64_peak.hip -o /tmp/mfma_f64_peak && /tmp/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) double f64x4;

template <int NACC>
__global__ __launch_bounds__(512) void peak(double* out, int iters, double a0, double b0) {
    f64x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f64x4){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)       // inline asm: keeps the accumulators in VGPRs (the builtin made hipcc shuttle them through AGPRs)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int wgs_per_cu, int threads, int iters) {
    int dev = 0;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, dev);
    const int cus = p.multiProcessorCount;
    double* out;
    hipMalloc(&out, (size_t)cus * wgs_per_cu * threads * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(peak<NACC>, dim3(cus * wgs_per_cu), dim3(threads), 0, 0, out, iters, 1.0001, 0.9999);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)cus * wgs_per_cu * (threads / 64) * (double)iters * NACC * 2048.0;
        if (rep == 2)
            printf("acc=%2d  %d WG/CU x %d threads (%.0f waves/SIMD): %.3f ms  %.1f TFLOP/s fp64\n", NACC, wgs_per_cu, threads,
                   wgs_per_cu * threads / 256.0, ms, flops / ms / 1e9);
    }
    hipFree(out);
}

int main() {
    run<16>(1, 256, 20000);
    run<16>(2, 256, 20000);
    run<16>(1, 512, 20000);
    run<8>(2, 256, 40000);
    run<4>(1, 256, 80000);
    run<4>(2, 256, 80000);
    run<1>(2, 256, 160000);
    return 0;
}
