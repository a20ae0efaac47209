#!/bin/bash
# Builds variants of distinctive_score_kernel's shape (columns per workgroup x rows per thread and batch) as separate
# libraries (deeploopcloser_amd/libdlc_ds_<cols>_<u>.so) for scripts/exp/ds_variants_run.py.  Run here (no GPU needed).
set -e
cd "$(dirname "$0")/../../deeploopcloser_amd/csrc"
make -s
for v in "16 16" "8 16" "8 8" "32 8" "4 16"; do
  set -- $v
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DDLC_DS_COLS=$1 -DDLC_DS_U=$2 -c match_ref.hip -o build/match_ref_ds.o
  objs=$(ls build/*.o | grep -v "build/match_ref.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../libdlc_ds_$1_$2.so $objs
  echo built libdlc_ds_$1_$2.so
done
rm -f build/match_ref_ds.o
