#!/bin/bash
# rocprofv3 kernel stats of any python tool: bash tools/prof_any.sh <tag> <script> [args]   (on the GPU box)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
d=$R/gpurun_out/prof_$tag
rm -rf $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 $R/"$@" > $d.log 2>&1
python3 $R/tools/summarize_prof.py "$d/p_kernel_stats.csv" | grep -v "at::native\|rocclr\|rocprim" | head -14
