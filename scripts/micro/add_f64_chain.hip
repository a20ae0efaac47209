// Microbenchmark (GPU box only): the issue-to-issue distance of DEPENDENT v_add_f64 -- the row-ordered column sum of the SDAV
// similarity's distinctive score (SimilarityCalculator.py:20-23) is one such chain of `rows` adds per column, whatever feeds it.
// One wave, 16 or 64 active lanes, 4096 dependent adds from registers, shader cycles by s_memtime.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/add_f64_chain.hip -o /tmp/add_f64_chain && /tmp/add_f64_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void chain(double* out, long long* cyc, int lanes, double x0) {
    double s = 0.0, x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = x0 + i + threadIdx.x;
    if ((int)threadIdx.x < lanes) {
        const long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < 256; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(s) : "v"(x[i]));
        }
        const long long t1 = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0) cyc[0] = t1 - t0;
    }
    out[threadIdx.x] = s;
}
int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
    for (int lanes : {16, 64}) {
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, out, cyc, lanes, 0.5);
        hipDeviceSynchronize();
        long long h = 0; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("%d active lanes: %lld s_memtime ticks for 4096 dependent v_add_f64 = %.2f per add (s_memtime counts at 100 MHz on gfx950: x shader clock / 100 MHz)\n", lanes, h, h / 4096.0);
    }
    return 0;
}
