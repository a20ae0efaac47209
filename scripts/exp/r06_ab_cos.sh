# A/B on one box: configs[1]'s cosine top-20 with the library as built vs other builds under csrc/exp_build/
R=$GRAFT_REPO_ROOT
for i in 1 2; do
  for v in "" $R/deeploopcloser_amd/csrc/exp_build/libdlc_*.so; do
    echo -n "$(basename ${v:-shipped}): "; python3 $R/scripts/prof_cos_topk.py 1063 20 $v 2>/dev/null | tail -1
  done
done
