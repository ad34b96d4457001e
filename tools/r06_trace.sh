#!/bin/bash
# timelines of the one-rank-of-8 step, sequential vs pipelined (VERDICT r5 next #1a)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for s in seq ahead; do
  for f in "" "--fcn"; do
    tag=r06_tl_${s}${f/--/_}
    timeout -k 10 200 python3 $R/tools/rank_step_lab.py $f --schedule $s --check > $R/gpurun_out/$tag.plain.log 2>&1 || { tail -5 $R/gpurun_out/$tag.plain.log; exit 1; }
    tail -1 $R/gpurun_out/$tag.plain.log
    rm -rf $R/gpurun_out/$tag
    timeout -k 10 300 rocprofv3 --hip-trace --kernel-trace --output-format csv -d $R/gpurun_out/$tag -o p -- python3 $R/tools/rank_step_lab.py $f --schedule $s > $R/gpurun_out/$tag.log 2>&1 || { tail -5 $R/gpurun_out/$tag.log; exit 1; }
    anchor=k_plan_accumulate; [ -n "$f" ] && anchor=k_bin_stream
    python3 $R/tools/timeline.py $R/gpurun_out/$tag $anchor 100 > $R/gpurun_out/$tag.timeline.txt 2>&1
    cat $R/gpurun_out/$tag.timeline.txt
    find $R/gpurun_out/$tag -name '*.csv' -size +8M -delete
  done
done
