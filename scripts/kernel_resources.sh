#!/bin/bash
# Per-kernel register / spill / LDS summary of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
# usage: scripts/kernel_resources.sh deeploopcloser_amd/csrc/cosine_topk.hip [name filter]
f=$1; pat=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -c "$f" -o /dev/null \
    -Rpass-analysis=kernel-resource-usage 2>&1 | awk -v pat="$pat" '
  /Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-R.*/,"",name)}
  /    VGPRs:/ {v=$(NF-1)} /AGPRs:/ {a=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /VGPRs Spill/ {vs=$(NF-1)} /SGPRs Spill/ {ss=$(NF-1)}
  /LDS Size/ {l=$(NF-1); if (name ~ pat) printf "%-110s vgpr %3s agpr %3s scratch %4s vspill %3s sspill %3s lds %6s\n", name, v, a, s, vs, ss, l}
  /error/ {print}'
