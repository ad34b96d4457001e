#!/bin/bash
# round 5, first batch of experiments on the binned route (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05_exp1; mkdir -p $O
cd $R
python3 tools/exp_c4_two_streams.py > $O/two_streams.txt 2>&1
python3 tools/exp_c4_two_streams.py --homo >> $O/two_streams.txt 2>&1
python3 tools/exp_c4_two_streams.py --n-post 1250000 --k 125 >> $O/two_streams.txt 2>&1
cat $O/two_streams.txt
for E in "X=0" "BE_BIN_COUNT=128" "BE_BIN_COUNT=512" "BE_BIN_COUNT=128 BE_BIN_CAP=64" "BE_BIN_COUNT=320"; do
  echo "== env: $E"
  env $E bash tools/prof_strong.sh ab --workload fcn --no-secondary 2>&1 | grep "k_bin_\|ms_per_step\|k_compact" | cut -c1-60,82-130
done > $O/rank_env.txt 2>&1
cat $O/rank_env.txt
ONLY="c4 c4_rank_of_8" bash tools/run_profiles.sh r05 cnt stats 2>&1 | grep "k_bin\|ms_per" | cut -c1-150
