#!/bin/bash
# Round-3 evidence, one file per configuration so that every roofline fraction of the bench line can be recomputed from
# profiles/ alone: rocprofv3 kernel stats of C2 (headline), C3, C4, C4 homo, C5 and of one rank of the 8-way cut of C2 / C4, each
# as its own bench.py run; HBM traffic counters of C2 and C4 in separate --pmc passes (FETCH_SIZE, WRITE_SIZE) as the guide
# prescribes; the bench line of the default command.
# usage (GPU box, repo root): bash tools/run_r03_profiles.sh <tag>     -> gpurun_out/r03_<tag>/
tag=${1:-a}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_$tag
rm -rf $O; mkdir -p $O
stats() {   # name, bench args...
  local name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s_$name -o p -- python3 $R/bench.py "$@" > $O/s_$name.log 2>&1
  echo "stats $name rc=$?"
  { echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py $*"; python3 $R/tools/summarize_prof.py "$O/s_$name/p_kernel_stats.csv" | grep -v "at::native\|rocclr\|rocprim" | head -14 | cut -c1-150;
    grep -o '"ms_per_step": [0-9.]*' $O/s_$name.log | head -1; } > $O/${name}_kernel_stats.txt
}
stats c2 --no-secondary --no-cpu
stats c2_homo --homo --no-secondary --no-cpu
stats c3 --workload jitc --no-cpu
stats c4 --workload fcn --no-cpu
stats c4_homo --workload fcn --homo --no-cpu
stats c5 --workload dense --no-cpu
stats c2_rank_of_8 --emulate-world 8 --steps 100 --warmup 20 --no-cpu --no-secondary
stats c4_rank_of_8 --emulate-world 8 --workload fcn --steps 100 --warmup 20 --no-cpu --no-secondary
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/c2_$c -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu --no-secondary > $O/c2_$c.log 2>&1
  echo "c2 $c rc=$?"
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/c4_$c -o p -- python3 $R/bench.py --workload fcn --steps 10 --warmup 3 --no-cpu > $O/c4_$c.log 2>&1
  echo "c4 $c rc=$?"
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/c4h_$c -o p -- python3 $R/bench.py --workload fcn --homo --steps 10 --warmup 3 --no-cpu > $O/c4h_$c.log 2>&1
  echo "c4 homo $c rc=$?"
done
{ echo "# separate --pmc passes; FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them (guide: FETCH_SIZE x 2 on gfx950 for wide coalesced reads)";
  echo "# C2 headline"; python3 $R/tools/summarize_prof.py "$O/c2_*/*counter_collection.csv" | grep -i "plan_acc\|plan_red\|compact\|kernel ";
  echo "# C4 hetero (binned route)"; python3 $R/tools/summarize_prof.py "$O/c4_*/*counter_collection.csv" | grep -i "k_bin\|compact\|kernel ";
  echo "# C4 homo"; python3 $R/tools/summarize_prof.py "$O/c4h_*/*counter_collection.csv" | grep -i "k_bin\|compact\|kernel "; } > $O/pmc_c2_c4.txt
( cd $R && timeout -k 10 900 python3 bench.py > $O/bench.log 2>&1 ); echo "bench rc=$?"
grep '^{' $O/bench.log | tail -1 > $O/bench_line.json
cat $O/*_kernel_stats.txt | cut -c1-150
cat $O/pmc_c2_c4.txt
tail -c 600 $O/bench_line.json
