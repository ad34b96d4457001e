#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for pg in "" "--pg"; do for f in "" "--fcn"; do for s in seq ahead_auto; do
  echo -n "pg='$pg': "; timeout -k 10 200 python3 tools/rank_step_lab.py $f $pg --schedule $s --check --steps 300 2>&1 | grep "rank 0 of" | tail -1
done; done; done | tee gpurun_out/r06_rank_queues.txt
