#!/bin/bash
# the rank step's schedules, one rank of eight on one GPU (plain timings; no profiler)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 600 python -m pytest tests/test_dist_gpu.py -x -q > gpurun_out/r06_rank_tests.log 2>&1; echo "dist tests rc=$?"; tail -3 gpurun_out/r06_rank_tests.log
for f in "" "--fcn"; do
  for s in seq ahead ahead_ids; do
    timeout -k 10 200 python3 tools/rank_step_lab.py $f --schedule $s --check --steps 500 2>&1 | tail -1
  done
done | tee gpurun_out/r06_rank_schedules.txt
