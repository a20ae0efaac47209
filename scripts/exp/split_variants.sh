#!/bin/bash
# Timing-only variants of csrc/gemm_split_f16.hip (WRONG RESULTS by construction): what the k loop costs without its DMA
# (every tile multiplies whatever the prologue left in LDS), without its barrier, without its fragment reads, with none of
# the three (the MFMA stream + epilogue alone), and without the epilogue's arithmetic and stores.  Patched COPIES, built
# into var_build/lib_split_<variant>.so (git-ignored; delete after use); run: python scripts/exp/split_variants_run.py
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/deeploopcloser_amd/csrc
T=$(mktemp -d)
mkdir -p $R/var_build
build() {
    name=$1
    mkdir -p $T/$name && cp $C/*.hip $C/*.h $T/$name/
    sed -i "s#../../include/dlc.h#$R/include/dlc.h#" $T/$name/dlc_internal.h
    python3 - "$T/$name/gemm_split_f16.hip" "$2" <<'PY'
import sys
p, names = sys.argv[1], sys.argv[2].split(",")
s = open(p).read()
P = {
 "nodma": [("        sp_dma4(loader, voff, (SRC) + (long long)ss_ * slice_b", "        sp_dma4((S2) < 3 ? loader : 0u, voff, (SRC) + (long long)ss_ * slice_b")],
 "nobar": [("        sp_barrier();                                              /* slice s is dead; slice s + 1 is visible */  \\\n", "        \\\n")],
 "nolds": [('#define SP_READ_W(DST, BASE) _Pragma("unroll") for (int tt = 0; tt < 8; ++tt) DST[tt] = *(lds_u4p)(lbase + (BASE) + rdW + tt * 1024)',
            '#define SP_READ_W(DST, BASE) _Pragma("unroll") for (int tt = 0; tt < 8; ++tt) asm volatile("" : "+v"(DST[tt]))'),
           ('#define SP_READ_H(DST, BASE) _Pragma("unroll") for (int c = 0; c < 4; ++c) DST[c] = *(lds_u4p)(lbase + (BASE) + rdH + c * 1024)',
            '#define SP_READ_H(DST, BASE) _Pragma("unroll") for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(DST[c]))')],
 "noprio": [("        __builtin_amdgcn_s_setprio(1);                                                                           \\\n", "        \\\n"),
            ("        __builtin_amdgcn_s_setprio(0);                                                                           \\\n", "        \\\n")],
 # cycles a wave spends between arriving at the slice's barrier (its own waits done) and leaving it, summed over the k loop
 "barstamp": [("        sp_barrier();                                              /* slice s is dead; slice s + 1 is visible */  \\\n",
               "        { const unsigned long long b0_ = __builtin_amdgcn_s_memtime(); sp_barrier(); bar_ += __builtin_amdgcn_s_memtime() - b0_; } \\\n"),
              ("    f32x4_t acc[8][4];\n", "    unsigned long long bar_ = 0;\n    f32x4_t acc[8][4];\n"),
              ("d_[1] = (long long)(sr1_ - sr0_); }", "d_[1] = (long long)bar_; }")],
 # cycles from the end of the k loop to the end of the kernel (bias staging, sigmoid, re-split, stores), per wave
 "epistamp": [("    const unsigned long long st1_ = __builtin_amdgcn_s_memtime(), sr1_ = __builtin_amdgcn_s_memrealtime();", "    const unsigned long long st1_ = __builtin_amdgcn_s_memtime(), sr1_ = __builtin_amdgcn_s_memrealtime();\n    const long long ix_e_ = (tile_m * p.tiles_n + tile_n) * 8 + wid;"),
              ("            __builtin_amdgcn_wave_barrier();\n        }\n    }\n}", "            __builtin_amdgcn_wave_barrier();\n        }\n    }\n    if (!FINAL && lane == 0 && p.M == 31890) { asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\"); long long* d_ = (long long*)(p.O[0] + (23 + ix_e_ / 440) * p.oslice_b + (p.M + (ix_e_ % 440) / 4) * 64 + (ix_e_ % 4) * 16); d_[0] = (long long)(__builtin_amdgcn_s_memtime() - st1_); d_[1] = 0; }\n}")],
 "noepi": [("            if (m >= p.M) continue;\n            float hv[16];", "            if (m >= p.M || p.ns > 1) continue;\n            float hv[16];")],
 # in-kernel stamps of the k loop of the HIDDEN layers (shader cycles and 100 MHz ticks per wave), stored where nothing else
 # is: rows >= M of the output piece (440 entries of 16 bytes per k-slice); the epilogue stays (the MFMAs must stay alive)
 "stamp": [("    // ---- prologue: early waves issue (W2, h2)", "    const unsigned long long st0_ = __builtin_amdgcn_s_memtime(), sr0_ = __builtin_amdgcn_s_memrealtime();\n    // ---- prologue: early waves issue (W2, h2)"),
           ("    SP_WAIT_VMCNT(0);    // the clamped tail DMAs must not outlive the workgroup's LDS\n    SP_WAIT_LGKM0();",
            "    SP_WAIT_VMCNT(0);    // the clamped tail DMAs must not outlive the workgroup's LDS\n    SP_WAIT_LGKM0();\n    const unsigned long long st1_ = __builtin_amdgcn_s_memtime(), sr1_ = __builtin_amdgcn_s_memrealtime();"),
           ("    const int lg = lane >> 4;\n    // sigmoid(z)", "    if (!FINAL && lane == 0 && p.M == 31890) { const long long ix_ = (tile_m * p.tiles_n + tile_n) * 8 + wid; long long* d_ = (long long*)(p.O[0] + (ix_ / 440) * p.oslice_b + (p.M + (ix_ % 440) / 4) * 64 + (ix_ % 4) * 16); d_[0] = (long long)(st1_ - st0_); d_[1] = (long long)(sr1_ - sr0_); }\n    const int lg = lane >> 4;\n    // sigmoid(z)")],
}
for n in names:
    if not n:
        continue
    for old, new in P[n]:
        assert s.count(old) >= 1, (n, old)
        s = s.replace(old, new)
open(p, "w").write(s)
PY
    (cd $T/$name && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -Wno-unused-function *.hip -o $R/var_build/lib_split_$name.so)
    echo built var_build/lib_split_$name.so
}
build epistamp "stamp,epistamp" &
wait
