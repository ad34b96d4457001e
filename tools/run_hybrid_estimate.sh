#!/bin/bash
# pass B / pass C kernel times of tools/exp_hybrid_estimate.py over block sizes and slice counts (on the GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for cfg in "8 16" "8 32" "8 64" "8 128" "4 64" "16 128"; do
  set -- $cfg
  echo "== S=$1 cap=$2"
  BE_EXP_S=$1 BE_BIN_CAP=$2 BE_BIN_COUNT=$(( (611 + $1 - 1) / $1 )) bash tools/prof_any.sh hy_$1_$2 tools/exp_hybrid_estimate.py | grep "k_bin\|k_compact" | cut -c1-50,82-130
  grep "us/step" gpurun_out/prof_hy_$1_$2.log
done
