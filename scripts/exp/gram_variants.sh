#!/bin/bash
# Timing-only variants of csrc/gram_i8.hip (WRONG RESULTS by construction): what the K loop costs without its DMA, its
# barrier, its LDS fragment reads, with every tile reading tile 0's panels (L2-fed), and a build with s_memtime stamps.  The shipped source carries no experiment switches: each variant is a patched COPY of it,
# built into exp_build/lib_gram_<variant>.so.  Run on the GPU box with scripts/exp/gram_variants_run.py.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/deeploopcloser_amd/csrc
T=$(mktemp -d)
mkdir -p $R/exp_build
build() {   # name, python patch expression on the source text `s`
    name=$1
    mkdir -p $T/$name && cp $C/*.hip $C/*.h $T/$name/
    sed -i "s#../../include/dlc.h#$R/include/dlc.h#" $T/$name/dlc_internal.h
    python3 - "$T/$name/gram_i8.hip" "$2" <<'PY'
import sys
p, expr = sys.argv[1], sys.argv[2]
s = open(p).read()
for old, new in eval(expr):
    assert old in s, old
    s = s.replace(old, new)
open(p, "w").write(s)
PY
    (cd $T/$name && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -Wno-unused-function *.hip -o $R/exp_build/lib_gram_$name.so)
    echo built exp_build/lib_gram_$name.so
}
BAR='        __builtin_amdgcn_s_barrier();                                                                           \\'
NODMA="('if (ISSUE) issue_piece((B0) + (J));', ''), ('if (ISSUE) issue_done();', '')"
NOLDS="('#define GI_RD(DST, PTR) { __builtin_amdgcn_sched_barrier(0); DST = *(const v4i*)(PTR); __builtin_amdgcn_sched_barrier(0); }', '#define GI_RD(DST, PTR) {}')"
NOBAR="('''$BAR''', '        \\\\')"
L2FED="('src[b] = uniform_ptr(p.X + (m0 / 16) * p.gpitch + (long long)(w * 2 + b / 3) * p.gpitch + (long long)(b % 3) * n64 * 1024);', 'src[b] = uniform_ptr(p.X + (long long)(w * 2 + b / 3) * p.gpitch + (long long)(b % 3) * n64 * 1024);'), ('const long long c = n0 + (w * 2 + gi) * 16 + (lane & 15);', 'const long long c = (w * 2 + gi) * 16 + (lane & 15);')"
# s_memtime stamps summed over the tiles (cycles per tile on stderr after every launch); the check of the object code does
# not apply to patched copies (they are built without `make`)
PROF="('__global__ __launch_bounds__(256) void gram_i8_kernel(const GramI8Args p) {', '__device__ unsigned long long gi_prof[8];\n__global__ __launch_bounds__(256) void gram_i8_kernel(const GramI8Args p) {\n    const unsigned long long TS = __builtin_amdgcn_s_memtime();'), \
('    if (p.keys[2]) return;                          // a NaN', '    const unsigned long long T0 = __builtin_amdgcn_s_memtime();\n    if (p.keys[2]) return;                          // a NaN'), \
('    GI_RDX(0, 0); GI_RDX(2, 0); GI_RDYB(0, 0, 0);', '    const unsigned long long T1 = __builtin_amdgcn_s_memtime();\n    GI_RDX(0, 0); GI_RDX(2, 0); GI_RDYB(0, 0, 0);'), \
('    __syncthreads();                                                     // every wave is through', '    const unsigned long long T2 = __builtin_amdgcn_s_memtime();\n    __syncthreads();                                                     // every wave is through'), \
('        p.acand[fj * p.rp + a] = cand;                                  // 0 = decided (one patch inside the window)\n    }\n', '        p.acand[fj * p.rp + a] = cand;\n    }\n    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n    const unsigned long long T3 = __builtin_amdgcn_s_memtime();\n    if (threadIdx.x == 0) { atomicAdd(&gi_prof[0], T0 - TS); atomicAdd(&gi_prof[1], T1 - T0); atomicAdd(&gi_prof[2], T2 - T1); atomicAdd(&gi_prof[3], T3 - T2); atomicAdd(&gi_prof[4], 1ull); }\n'), \
('    DLC_LAUNCH_CHECK(ctx, \"gram_i8_kernel\");', '    DLC_LAUNCH_CHECK(ctx, \"gram_i8_kernel\");\n    { unsigned long long h[8], z[8] = {}; (void)hipDeviceSynchronize(); (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(gi_prof), 64); (void)hipMemcpyToSymbol(HIP_SYMBOL(gi_prof), z, 64); fprintf(stderr, \"gi_prof tiles %llu: lookup %.0f prologue %.0f loop %.0f epilogue %.0f cycles per tile\\\\n\", h[4], (double)h[0] / h[4], (double)h[1] / h[4], (double)h[2] / h[4], (double)h[3] / h[4]); }')"
build prof "[$PROF]" &
build base "[]" &
build nodma "[$NODMA]" &
wait
build nobarrier "[$NOBAR]" &
build nolds "[$NOLDS]" &
build l2fed "[$L2FED]" &
wait
NOSTORE="('        p.abi[fj * p.rp + a] = (unsigned char)bi;', '        if (bi == 77) p.abi[fj * p.rp + a] = (unsigned char)bi;'), ('        p.acand[fj * p.rp + a] = cand;', '        if (bi == 77) p.acand[fj * p.rp + a] = cand;')"
build nostore "[$NOSTORE]" &
build mfma_only "[$NODMA, $NOLDS, $NOBAR]"
wait
rm -rf $T
