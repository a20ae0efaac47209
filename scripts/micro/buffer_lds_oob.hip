// What a `buffer_load_dwordx4 ... offen lds` writes to LDS for lanes whose offset is past num_records (gfx950):
// build: hipcc --offload-arch=gfx950 -O3 scripts/micro/buffer_lds_oob.hip -o exp_build/buffer_lds_oob ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u4;
__global__ void k(const char* p, unsigned num_records, unsigned soff, float* out) {
    extern __shared__ char sm[];
    float* f = (float*)sm;
    for (int i = threadIdx.x; i < 512; i += 64) f[i] = -7.f;            // what LDS held before
    __syncthreads();
    // lanes 0-31 in range (16 B each), lanes 32-47 far out of range, lanes 48-63 straddle / just past the end
    unsigned off = threadIdx.x < 32 ? threadIdx.x * 16 : (threadIdx.x < 48 ? 0xfffffff0u : num_records - 8 + (threadIdx.x - 48) * 16);
    u4 rsrc;
    unsigned long long a = (unsigned long long)p;
    rsrc[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    rsrc[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffff);
    rsrc[2] = num_records;
    rsrc[3] = 0x00020000;
    unsigned m = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)sm;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds\n\ts_waitcnt vmcnt(0)" :: "v"(off), "s"(rsrc), "s"(m), "s"(soff) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = f[i];
}
int main() {
    const int n = 4096;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = 1.f + i;
    char* d; float* o;
    hipMalloc(&d, n * 4); hipMalloc(&o, 256 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    for (unsigned soff : {0u, 1024u}) {
        const unsigned nr = 1024;                                        // bytes: floats 0..255 are in range
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, nr, soff, o);
        std::vector<float> r(256);
        hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
        printf("num_records %u soffset %u\n", nr, soff);
        for (int lane : {0, 1, 31, 32, 47, 48, 49, 63})
            printf("  lane %2d: %g %g %g %g\n", lane, r[lane * 4], r[lane * 4 + 1], r[lane * 4 + 2], r[lane * 4 + 3]);
    }
    return 0;
}
