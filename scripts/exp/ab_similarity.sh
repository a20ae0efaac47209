# A/B of two builds of the library on the similarity call: bash scripts/exp/ab_similarity.sh  (var_build/lib_old.so against the tree's)
for kind in saturated real real_fan_in; do
  for rep in 1 2; do
  for lib in var_build/lib_old.so deeploopcloser_amd/libdlc_hip.so; do
    echo "== $kind $lib: $(timeout -k 10 200 python scripts/prof_similarity.py $kind $lib | tail -2 | tr '\n' ' ')"
  done; done
done
