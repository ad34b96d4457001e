#!/bin/bash
# rocprofv3 kernel stats of the emulated 1-of-8 shard step; usage: bash tools/prof_strong.sh <tag> [bench args]
tag=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
d=$R/gpurun_out/prof_strong_$tag
rm -rf $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 $R/bench.py --emulate-world 8 --steps 100 --warmup 20 --no-cpu "$@" > $d.log 2>&1
python3 $R/tools/summarize_prof.py "$d/p_kernel_stats.csv" | grep -v "at::native\|rocclr" | head -14
grep -o '"ms_per_step": [0-9.]*' $d.log
