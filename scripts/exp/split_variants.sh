#!/bin/bash
# Timing-only variants of csrc/gemm_split_f16.hip (WRONG RESULTS by construction): what the K loop costs without its DMA
# (every tile multiplies whatever the prologue left in LDS), without its barriers, without its fragment reads, and L2-fed
# (every workgroup streams tile (0, 0)'s panels).  Patched COPIES, built into exp_build/lib_split_<variant>.so; run
# scripts/prof_sdav_split.py with DLC_LIB-less argv: python scripts/exp/split_variants_run.py
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/deeploopcloser_amd/csrc
T=$(mktemp -d)
mkdir -p $R/exp_build
build() {
    name=$1
    mkdir -p $T/$name && cp $C/*.hip $C/*.h $T/$name/
    sed -i "s#../../include/dlc.h#$R/include/dlc.h#" $T/$name/dlc_internal.h
    python3 - "$T/$name/gemm_split_f16.hip" "$2" <<'PY'
import sys
p, expr = sys.argv[1], sys.argv[2]
s = open(p).read()
for old, new in eval(expr):
    assert old in s, old
    s = s.replace(old, new)
open(p, "w").write(s)
PY
    (cd $T/$name && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -Wno-unused-function *.hip -o $R/exp_build/lib_split_$name.so)
    echo built exp_build/lib_split_$name.so
}
NODMA="('        sp_dma4(voff[H], src_, lds_stage + (POS) + (H) * SP_HALF);                                               \\\\', '        if ((t2) < 2) sp_dma4(voff[H], src_, lds_stage + (POS) + (H) * SP_HALF);                                \\\\')"
NOBAR="('#define SP_RELEASE() SP_WAIT_LGKM0(); sp_barrier()', '#define SP_RELEASE() SP_WAIT_LGKM0()')"
NOLDS="('#define SP_READ_A(DST, RD, OFF) _Pragma(\"unroll\") for (int tt = 0; tt < 4; ++tt) DST[tt] = *(lds_u4p)(lbase + (RD) + (OFF) + tt * 512)', '#define SP_READ_A(DST, RD, OFF) _Pragma(\"unroll\") for (int tt = 0; tt < 4; ++tt) asm volatile(\"\" : \"+v\"(DST[tt]))'), ('#define SP_READ_B(DST, RD, OFF) _Pragma(\"unroll\") for (int c = 0; c < 2; ++c) DST[c] = *(lds_u4p)(lbase + (RD) + (OFF) + c * 2048)', '#define SP_READ_B(DST, RD, OFF) _Pragma(\"unroll\") for (int c = 0; c < 2; ++c) asm volatile(\"\" : \"+v\"(DST[c]))')"
L2FED="('p.W[0] + (long long)tile_n * SP_BM * p.ldw_b : p.X[0] + tile_m * SP_BN * p.ldx_b);', 'p.W[0] : p.X[0]);'), ('p.W[1] + (long long)tile_n * SP_BM * p.ldw_b : p.X[1] + tile_m * SP_BN * p.ldx_b);', 'p.W[1] : p.X[1]);')"
build base "[]" &
build nodma "[$NODMA]" &
build nobar "[$NOBAR]" &
build nolds "[$NOLDS]" &
build l2fed "[$L2FED]" &
wait
