#!/bin/bash
# Build an experimental variant of libdlc_hip.so:  scripts/exp_build.sh <name> [-DFLAG ...]
set -e
cd "$(dirname "$0")/../deeploopcloser_amd/csrc"
name=$1; shift
mkdir -p ../../exp_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" api.hip cosine_topk.hip gemm_dense.hip gemm_dma_f64.hip match_ref.hip cnnvtl.hip frontend.hip train.hip -o ../../exp_build/lib_$name.so
echo built exp_build/lib_$name.so
