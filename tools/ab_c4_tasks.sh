#!/bin/bash
# A/B (round 4): pass B's task size at C4 itself (rows of 1000 entries = 250 groups of four) and the homogeneous variant
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "1024 2048" "512 2048" "256 2048" "128 2048" "1024 2048"; do
  set -- $cfg
  for homo in "" "--homo"; do
    BRAINEVENT_AMD_TUNING="{\"binned_task_groups\": $1, \"binned_min_tasks\": $2}" timeout -k 10 300 python3 bench.py --workload fcn $homo --steps 100 --warmup 20 --no-cpu --no-secondary > gpurun_out/ab_c4tg_$1_$2$homo.log 2>&1 || { tail -5 gpurun_out/ab_c4tg_$1_$2$homo.log; exit 1; }
    echo "C4 $homo task_groups $1 min_tasks $2: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab_c4tg_$1_$2$homo.log | head -1)"
  done
done
