#!/bin/bash
# rocprofv3 kernel stats of tools/exp_layouts.py (env BE_EXP_* / BE_PLAN_* select the case); on the GPU box: bash tools/prof_layouts.sh <tag>
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
d=$R/gpurun_out/prof_lay_$tag
rm -rf $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 $R/tools/exp_layouts.py > $d.log 2>&1
grep -v amdgpu.ids $d.log | tail -4
python3 $R/tools/summarize_prof.py "$d/*/*kernel_stats.csv" "$d/*kernel_stats.csv" 2>/dev/null | grep -v "at::native\|rocclr" | head -12
