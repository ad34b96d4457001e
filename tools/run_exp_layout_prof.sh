# kernel-level timing of exp_layouts.py cases: env CASES="<name>:<env assignments separated by commas>;..."
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/explay
mkdir -p $O
IFS=';' read -ra CS <<< "$CASES"
for c in "${CS[@]}"; do
  name=${c%%:*}; envs=${c#*:}
  ( IFS=','; for kv in $envs; do export "$kv"; done
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o p -- python3 $R/tools/exp_layouts.py > $O/$name.log 2>&1 )
  echo "## $name"; grep "^N=" $O/$name.log
  python $R/tools/summarize_prof.py $O/$name/p_kernel_stats.csv | grep "k_plan_acc\|k_plan_red\|k_compact\|k_plan_single" | cut -c1-60,80-130
done
