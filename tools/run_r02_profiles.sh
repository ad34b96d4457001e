#!/bin/bash
# Round-2 evidence: (1) rocprofv3 kernel stats of the default bench command (headline + secondary configs),
# (2) HBM traffic counters of the dominant kernels, separate --pmc passes (FETCH_SIZE, WRITE_SIZE) as the guide prescribes.
# usage (GPU box, repo root): bash tools/run_r02_profiles.sh <tag>
tag=${1:-a}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_$tag
rm -rf $O; mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 $R/bench.py > $O/bench_profiled.log 2>&1
echo "stats rc=$?"
python3 $R/tools/summarize_prof.py "$O/stats/p_kernel_stats.csv" | grep -v "at::native\|rocclr\|rocprim" > $O/kernel_stats.txt
( cd $R && timeout -k 10 400 python3 bench.py > $O/bench.log 2>&1 ); echo "bench rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/c2_$c -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu --no-secondary > $O/c2_$c.log 2>&1
  echo "c2 $c rc=$?"
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/c4_$c -o p -- python3 $R/bench.py --workload fcn --steps 10 --warmup 3 > $O/c4_$c.log 2>&1
  echo "c4 $c rc=$?"
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/c4h_$c -o p -- python3 $R/bench.py --workload fcn --homo --steps 10 --warmup 3 > $O/c4h_$c.log 2>&1
  echo "c4 homo $c rc=$?"
done
# the K = 1000 planned regime: per-step times over N and the kernel stats at N = 1M weighted
( cd $R && BE_EXP_NS=100000,200000,350000,500000,1000000,1500000,2500000 timeout -k 10 400 python3 tools/exp_layouts.py 2>&1 | grep -v amdgpu.ids > $O/k1000_sweep.txt ); echo "k1000 sweep rc=$?"
BE_EXP_LAYOUTS=None BE_EXP_NS=1000000 BE_EXP_HETERO_ONLY=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k1000_1m -o p -- python3 $R/tools/exp_layouts.py > $O/k1000_1m.log 2>&1
python3 $R/tools/summarize_prof.py "$O/k1000_1m/p_kernel_stats.csv" | grep -v "at::native\|rocclr\|rocprim" | head -12 > $O/k1000_1m_kernel_stats.txt
python3 $R/tools/summarize_prof.py "$O/c2_*/*counter_collection.csv" | grep -i "plan_acc\|plan_red\|compact\|kernel " > $O/pmc_c2.txt
python3 $R/tools/summarize_prof.py "$O/c4_*/*counter_collection.csv" | grep -i "k_bin\|compact\|kernel " > $O/pmc_c4.txt
python3 $R/tools/summarize_prof.py "$O/c4h_*/*counter_collection.csv" | grep -i "k_bin\|compact\|kernel " > $O/pmc_c4_homo.txt
cat $O/kernel_stats.txt | cut -c1-150 | head -30
cat $O/pmc_c2.txt $O/pmc_c4.txt $O/pmc_c4_homo.txt
cat $O/k1000_sweep.txt $O/k1000_1m_kernel_stats.txt | cut -c1-160
tail -c 400 $O/bench.log
