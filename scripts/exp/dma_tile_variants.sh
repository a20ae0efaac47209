#!/bin/bash
# Timing-only variants of csrc/gemm_dma_f64.hip: launch_dma_f64 forced onto one tile height (256 / 128 / 64 rows) for plain
# launches, to calibrate its choice on short-K shapes (the training step's weight-gradient product, 1681 x 2500 x 600).
# Patched COPIES of the sources, built into exp_build/lib_dma_tm<rows>.so; run scripts/exp/dma_tile_run.py on the GPU box.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/deeploopcloser_amd/csrc
T=$(mktemp -d)
mkdir -p $R/exp_build
for tm in 256 128 64; do
  (
    mkdir -p $T/$tm && cp $C/*.hip $C/*.h $T/$tm/
    sed -i "s#../../include/dlc.h#$R/include/dlc.h#" $T/$tm/dlc_internal.h
    python3 - "$T/$tm/gemm_dma_f64.hip" $tm <<'PY'
import sys
p, tm = sys.argv[1], sys.argv[2]
s = open(p).read()
old = "    return launch_dma_part(ctx, blayout, act, M, N, K, A, lda, B, ldb, bias, C, ldc, st, cv, tri, Kb, 0, tm, false);\n}"
assert old in s
s = s.replace(old, "    if (!tri && !cv && N > 96) tm = %s;\n" % tm + old)
# the split launch (whole rounds of 256-row tiles + 128-row rest) is switched off too
s = s.replace("if (t4 > 256 && t4 % 256 != 0) {", "if (false) {")
open(p, "w").write(s)
PY
    (cd $T/$tm && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -Wno-unused-function *.hip -o $R/exp_build/lib_dma_tm$tm.so)
    echo built exp_build/lib_dma_tm$tm.so
  ) &
done
wait
