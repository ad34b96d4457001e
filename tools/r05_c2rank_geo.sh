#!/bin/bash
# C2 as one rank of 8: plan geometry sweep (parts per slice, slice width) — ms/step from the bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05_c2rank_geo.txt; : > $O
for extra in "" "--parts 8" "--parts 10" "--parts 14" "--parts 16" "--parts 28" "--width 5208" "--width 5208 --parts 10" "--width 7816 --parts 14" "--width 15632" "--width 15632 --parts 32"; do
  python3 bench.py --emulate-world 8 --steps 200 --warmup 30 --no-cpu --no-secondary $extra 2>/dev/null | grep '^{' | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$extra'.ljust(28), d['ms_per_step'], d['config']['plan_slices'], 'kernel_ms', d['roofline']['kernel_ms'], 'parity', d['parity_check']['ok'])" | tee -a $O
done
