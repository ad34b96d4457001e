#!/bin/bash
# one rank of an 8-way strong split, emulated on one GPU (bench.py --emulate-world 8): C2 and C4 shards, hetero + homo
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/strong_${1:-x}; mkdir -p $O
cd $R
for cfg in "c2:" "c2h:--homo" "c4:--workload fcn" "c4h:--workload fcn --homo"; do
  name=${cfg%%:*}; args=${cfg#*:}
  timeout -k 10 200 python bench.py --emulate-world 8 $args $BENCH_EXTRA --steps 100 --warmup 20 --no-cpu > $O/$name.log 2>&1
  python - "$O/$name.log" "$name" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith('{')]
if not l:
    print(sys.argv[2], 'FAILED'); print(open(sys.argv[1]).read()[-1500:]); sys.exit(0)
d = json.loads(l[-1])
print(sys.argv[2], 'value', d['value'], 'ms/step', d['ms_per_step'], 'step_ev', d['step_ms_hip_events'], 'kern', d['roofline']['kernel'], d['roofline']['kernel_ms'],
      'parity', d['parity_check']['ok'], d['parity_check']['error'], d['config']['plan_slices'], 'setup', d['config']['setup_s'])
PY
done
