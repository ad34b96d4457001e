#!/bin/bash
# Is pass C slowed down by what pass B left behind (write-back of the 600 MB it wrote)?  BE_DBG_C_TWICE=2 runs pass C three times per
# step on the same regions; the kernel trace gives the duration of the 1st / 2nd / 3rd launch of a step.  On the GPU box.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for h in "" "--homo"; do
  d=$R/gpurun_out/prof_twice$h
  rm -rf $d
  BE_DBG_C_TWICE=2 rocprofv3 --kernel-trace --output-format csv -d $d -o p -- python3 $R/bench.py --workload fcn $h --steps 30 --warmup 5 --no-cpu > $d.log 2>&1
  python3 - "$d" "$h" <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_bin_accumulate' in r['Kernel_Name'] or 'k_bin_stream' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
acc, k = {0: [], 1: [], 2: []}, 0
for r in rows:
    if 'k_bin_stream' in r['Kernel_Name']:
        k = 0
        continue
    acc[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    k += 1
for i in range(3):
    v = acc[i][5:]
    print(f'pass C launch {i + 1} of a step {sys.argv[2]}: mean {sum(v) / len(v):.1f} us over {len(v)} steps')
PY
done
