set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/explay
mkdir -p $O
export BE_EXP_HOMO_ONLY=1 BE_EXP_NS=350000
for lay in u16 h8; do
  BE_EXP_LAYOUTS=$lay timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$lay -o p -- python3 $R/tools/exp_layouts.py > $O/$lay.log 2>&1
  echo "$lay rc=$?"; tail -1 $O/$lay.log
  python $R/tools/summarize_prof.py $O/$lay/p_kernel_stats.csv | grep -v "at::native\|rocclr" | head -8
done
