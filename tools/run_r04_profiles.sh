#!/bin/bash
# Round-4 evidence, one file per configuration so that every roofline fraction of the bench line can be recomputed from
# profiles/ alone: rocprofv3 kernel stats of every configuration of the bench line, each as its own bench.py run; HBM traffic
# counters of every HBM-bound configuration in separate --pmc passes (FETCH_SIZE, WRITE_SIZE), as MI355X_MICROARCH.md
# prescribes (never combined with a trace domain); the bench line of the default command.
# usage (GPU box, repo root): bash tools/run_r04_profiles.sh <tag> [stats|pmc|bench ...]     -> gpurun_out/r04_<tag>/
tag=${1:-a}; shift
what=${*:-stats pmc bench}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_$tag
mkdir -p $O
declare -A CFG=(
  [c2]="--no-secondary --no-cpu"
  [c2_homo]="--homo --no-secondary --no-cpu"
  [c2_gather_mirror]="--workload gather_mirror --no-cpu --steps 60 --warmup 10"
  [c3]="--workload jitc --no-cpu"
  [c4]="--workload fcn --no-cpu --steps 60 --warmup 10"
  [c4_homo]="--workload fcn --homo --no-cpu --steps 60 --warmup 10"
  [c5]="--workload dense --no-cpu"
  [c2_rank_of_8]="--emulate-world 8 --steps 100 --warmup 20 --no-cpu --no-secondary"
  [c4_rank_of_8]="--emulate-world 8 --workload fcn --steps 100 --warmup 20 --no-cpu --no-secondary"
)
ORDER=${ONLY:-"c2 c2_homo c2_gather_mirror c3 c4 c4_homo c5 c2_rank_of_8 c4_rank_of_8"}      # ONLY="c4 c4_rank_of_8": a subset
PMC_ORDER=${ONLY:-"c2 c2_homo c2_gather_mirror c4 c4_homo c5 c2_rank_of_8 c4_rank_of_8"}
if [[ $what == *stats* ]]; then
  for name in $ORDER; do
    rm -rf $O/s_$name
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s_$name -o p -- python3 $R/bench.py ${CFG[$name]} > $O/s_$name.log 2>&1
    echo "stats $name rc=$?"
    { echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py ${CFG[$name]}"; python3 $R/tools/summarize_prof.py "$O/s_$name/p_kernel_stats.csv" | grep -v "at::native\|rocclr\|rocprim" | head -14 | cut -c1-150;
      grep -o '"ms_per_step": [0-9.]*' $O/s_$name.log | head -1; } > $O/${name}_kernel_stats.txt
  done
fi
if [[ $what == *pmc* ]]; then
  for name in $PMC_ORDER; do
    [ "$name" = c3 ] && continue
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf $O/p_${name}_$c
      timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/p_${name}_$c -o p -- python3 $R/bench.py ${CFG[$name]} --steps 12 --warmup 3 > $O/p_${name}_$c.log 2>&1
      echo "pmc $name $c rc=$?"
    done
  done
  { echo "# separate --pmc passes per configuration; FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them, mean per launch"
    echo "# (guide: bytes = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024 on gfx950 for wide coalesced reads)"
    for name in $PMC_ORDER; do
      [ "$name" = c3 ] && continue
      echo "# $name: bench.py ${CFG[$name]} --steps 12 --warmup 3"
      python3 $R/tools/summarize_prof.py "$O/p_${name}_FETCH_SIZE/*counter_collection.csv" "$O/p_${name}_WRITE_SIZE/*counter_collection.csv" | grep -i "k_plan\|k_bin\|k_compact\|k_dense\|k_mfma\|k_gather\|kernel " | cut -c1-150
    done; } > $O/pmc_all.txt
  python3 $R/tools/make_traffic_json.py $O > $O/traffic.json
fi
if [[ $what == *bench* ]]; then
  ( cd $R && timeout -k 10 900 python3 bench.py > $O/bench.log 2>&1 ); echo "bench rc=$?"
  grep '^{' $O/bench.log | tail -1 > $O/bench_line.json
fi
cat $O/*_kernel_stats.txt 2>/dev/null | cut -c1-150
cat $O/pmc_all.txt 2>/dev/null
tail -c 400 $O/bench_line.json 2>/dev/null
