#!/bin/bash
# the reference tuner's operating point (n = 500k, 2000 / row, one weight, exactly 2000 active): plan geometry / route variants
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05_tuner_geo.txt; : > $O
for extra in "" "--parts 4" "--parts 8" "--parts 16" "--parts 20" "--width 20000" "--width 20000 --parts 10" "--width 40000 --parts 19" "--layout u16" "--layout h8" "--route direct"; do
  python3 bench.py --n 500000 --conn 0.004 --fire 0.004 --homo --exact-active --steps 300 --warmup 50 --no-cpu --no-secondary $extra 2>/dev/null | grep '^{' | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$extra'.ljust(28), round(d['ms_per_step']*1e3,2), 'us', d['config']['plan_slices'], 'kernel_ms', (d['roofline'] or {}).get('kernel_ms'), 'parity', d['parity_check']['ok'])" | tee -a $O
done
