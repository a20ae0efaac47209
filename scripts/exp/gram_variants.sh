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
NOBAR="('''$BAR''', '        \\\\')"
build base "[]" &
build nodma "[$NODMA]" &
build nobarrier "[$NOBAR]" &
wait
build nolds "[$NOLDS]" &
build mfma_only "[$NODMA, $NOLDS, $NOBAR]" &
wait
rm -rf $T
