#!/bin/bash
# A/B (round 4): pass B's task size on the 1-of-8 post slice of C4 (rows of ~125 entries: 32 groups of four) through the tuning
# override (BRAINEVENT_AMD_TUNING; the library's BE_BIN_* variables only seed what the tuning then sets): rows per task =
# task_groups / 32, at least min_tasks tasks
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "1024 2048" "512 2048" "256 2048" "128 2048" "2048 1024" "1024 4096" "1024 2048"; do
  set -- $cfg
  BRAINEVENT_AMD_TUNING="{\"binned_task_groups\": $1, \"binned_min_tasks\": $2}" timeout -k 10 300 python3 bench.py --emulate-world 8 --workload fcn --steps 300 --warmup 50 --no-cpu --no-secondary > gpurun_out/ab_tg_$1_$2.log 2>&1 || { tail -5 gpurun_out/ab_tg_$1_$2.log; exit 1; }
  echo "task_groups $1 min_tasks $2: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab_tg_$1_$2.log | head -1) $(grep -o '"kernel_ms": [0-9.]*' gpurun_out/ab_tg_$1_$2.log | head -1)"
done
