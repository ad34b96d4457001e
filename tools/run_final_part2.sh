# secondary workloads under rocprofv3 (kernel stats) + their bench lines
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final2
mkdir -p $O
run() {  # name, args...
  name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o p -- python3 $R/bench.py "$@" > $O/$name.log 2>&1
  echo "$name rc=$?"
}
run jitc --workload jitc --steps 50 --warmup 10
run fcn_homo --workload fcn --homo --steps 25 --warmup 5
run fcn_hetero --workload fcn --steps 25 --warmup 5
run dense --workload dense --steps 25 --warmup 5
run dense_p50 --workload dense --fire 0.5 --steps 15 --warmup 5
run c2_homo --homo --no-cpu --steps 100 --warmup 20
cd $R
for n in jitc fcn_homo fcn_hetero dense dense_p50 c2_homo; do
  echo "## $n"; python tools/summarize_prof.py $O/$n/p_kernel_stats.csv | grep -v "at::native\|rocclr" | head -7; grep '^{"metric"' $O/$n.log | tail -1 | cut -c1-420
done
