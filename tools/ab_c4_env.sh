# usage: tools/ab_c4_env.sh "VAR=a" "VAR=b" ... — rocprofv3 kernel times of the C4 bench (hetero, homo) per environment setting; on the GPU box
for E in "$@"; do
  echo "== env: $E"
  env $E bash tools/prof_fcn.sh ab 2>&1 | grep "k_bin_stream\|k_bin_acc" | cut -c1-60,82-130
done
