# usage: tools/ab_rank8_env.sh "VAR=a" ... — one rank of the 8-way cut of C4 (hetero, homo): ms/step and kernel times per environment setting
for E in "$@"; do
  echo "== env: $E"
  for h in "" "--homo"; do
    env $E bash tools/prof_strong.sh ab --workload fcn --no-secondary $h 2>&1 | grep "k_bin_\|ms_per_step\|k_compact\|k_pack" | cut -c1-60,82-130
  done
done
