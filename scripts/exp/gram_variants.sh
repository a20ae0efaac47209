#!/bin/bash
# Timing-only variants of csrc/gram_i8.hip (WRONG RESULTS by construction): what the K loop costs without its DMA, its
# barrier, its LDS fragment reads.  The shipped source carries no experiment switches: each variant is a patched COPY of it,
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
NODMA="('if (ISSUE) issue_piece((B0) + j);', ''), ('if (ISSUE) issue_done();', '')"
NOLDS="('GI_RDX(1, so_c);', ''), ('GI_RDX(0, so_n); GI_RDYB(N, 0, so_n); GI_RDYB(N, 1, so_n); GI_RDY2(so_n);', ''), ('GI_RDX(2, so_n);', '')"
NOLDS_IL="('#define GI_RD(DST, PTR) { __builtin_amdgcn_sched_barrier(0); DST = *(const v4i*)(PTR); __builtin_amdgcn_sched_barrier(0); }', '#define GI_RD(DST, PTR) {}')"
NOBAR="('''$BAR''', '        \\\\')"
PROF="('__global__ __launch_bounds__(256) void gram_i8_kernel(const GramI8Args p) {', '__device__ unsigned long long gi_prof[8];\n__global__ __launch_bounds__(256) void gram_i8_kernel(const GramI8Args p) {\n    unsigned TS = (unsigned)__builtin_amdgcn_s_memtime(), a1 = 0, a2 = 0, a3 = 0, nt = 0;'), \
('        const int cur_m = tile_m, cur_n = tile_n;', '        const unsigned T0 = (unsigned)__builtin_amdgcn_s_memtime();\n        const int cur_m = tile_m, cur_n = tile_n;'), \
('        // two k-steps per trip (the y buffers alternate).', '        const unsigned T1 = (unsigned)__builtin_amdgcn_s_memtime();\n        a1 += T1 - T0;\n        // two k-steps per trip (the y buffers alternate).'), \
('        int* d2s = (int*)(smem_i8 + GI_NSTAGE * GI_STAGE) + w * 16 * 64;', '        const unsigned T2 = (unsigned)__builtin_amdgcn_s_memtime();\n        a2 += T2 - T1;\n        int* d2s = (int*)(smem_i8 + GI_NSTAGE * GI_STAGE) + w * 16 * 64;'), \
('        if (!has_next) break;', '        a3 += (unsigned)__builtin_amdgcn_s_memtime() - T2; ++nt;\n        if (!has_next) break;'), \
('    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");                    // the stream', '    if (threadIdx.x == 0) { atomicAdd(&gi_prof[1], (unsigned long long)a1); atomicAdd(&gi_prof[2], (unsigned long long)a2); atomicAdd(&gi_prof[3], (unsigned long long)a3); atomicAdd(&gi_prof[4], (unsigned long long)nt); atomicAdd(&gi_prof[0], (unsigned long long)((unsigned)__builtin_amdgcn_s_memtime() - TS)); atomicAdd(&gi_prof[5], 1ull); atomicMax(&gi_prof[6], (unsigned long long)((unsigned)__builtin_amdgcn_s_memtime() - TS)); atomicMax(&gi_prof[7], (unsigned long long)(0xffffffffu - ((unsigned)__builtin_amdgcn_s_memtime() - TS))); }\n    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");                    // the stream'), \
('    DLC_LAUNCH_CHECK(ctx, \"gram_i8_kernel\");', '    DLC_LAUNCH_CHECK(ctx, \"gram_i8_kernel\");\n    { unsigned long long h[8], z[8] = {}; (void)hipDeviceSynchronize(); (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(gi_prof), 64); (void)hipMemcpyToSymbol(HIP_SYMBOL(gi_prof), z, 64); fprintf(stderr, \"gi_prof tiles %llu: top %.0f loop %.0f epilogue %.0f cycles per tile; %llu workgroups of %.0f cycles (min %llu max %llu)\\\\n\", h[4], (double)h[1] / h[4], (double)h[2] / h[4], (double)h[3] / h[4], h[5], (double)h[0] / (h[5] ? h[5] : 1), 0xffffffffull - h[7], h[6]); }')"
build prof "[$PROF]" &
if [ "$1" = prof ]; then wait; rm -rf $T; exit 0; fi
L2FED="('set_src(tile_m, tile_n);', 'set_src(0, 0);')"
if [ "$1" = l2fed ]; then build l2fed "[$L2FED]"; build nodma "[$NODMA]"; build nolds "[$NOLDS_IL]"; rm -rf $T; exit 0; fi
build base "[]" &
build nodma "[$NODMA]" &
build nobarrier "[$NOBAR]" &
wait
build nolds "[$NOLDS]" &
build mfma_only "[$NODMA, $NOLDS, $NOBAR]" &
wait
rm -rf $T
