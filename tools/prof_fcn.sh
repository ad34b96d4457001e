#!/bin/bash
# rocprofv3 kernel stats of the C4 bench (hetero + homo); run on the GPU box from the repo root: bash tools/prof_fcn.sh <tag>
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for h in "" "--homo"; do
  d=$R/gpurun_out/prof_fcn_$tag$h
  rm -rf $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 $R/bench.py --workload fcn $h --steps 40 --warmup 10 > $d.log 2>&1
  python3 $R/tools/summarize_prof.py "$d/*/*kernel_stats.csv" "$d/*kernel_stats.csv" 2>/dev/null | grep -v "at::native\|rocclr" | head -12
done
