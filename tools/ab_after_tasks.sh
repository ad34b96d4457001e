#!/bin/bash
# after the task-size change: the binned configurations of the bench line with the new defaults
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { timeout -k 10 300 python3 bench.py "$@" --no-cpu --no-secondary > gpurun_out/after_tasks.log 2>&1 || { tail -5 gpurun_out/after_tasks.log; exit 1; }
        echo "$*: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/after_tasks.log | head -1)"; }
run --emulate-world 8 --workload fcn --steps 300 --warmup 50
run --emulate-world 8 --workload fcn --homo --steps 300 --warmup 50
run --workload fcn --steps 100 --warmup 20
run --workload fcn --homo --steps 100 --warmup 20
run --workload fcn --acc32 --steps 100 --warmup 20
