#!/bin/bash
# Per-round evidence, one file per configuration of the bench line, so that every roofline fraction can be recomputed from
# profiles/ alone.  rocprofv3 --kernel-trace --stats of each configuration as its own bench.py run; the summary reports, per
# kernel, the MEDIAN over the timed launches (the last <steps> of every kernel; tools/summarize_prof.py --last) next to the
# plain average that rocprofv3's own stats print — warm-up launches pollute the average.  HBM traffic of the
# HBM-bound configurations in separate --pmc passes (FETCH_SIZE, WRITE_SIZE), never combined with a trace domain; SQ counters of
# the JITC walks (c3, c3_gather) in their own pass.
# usage (GPU box, repo root): bash tools/run_profiles.sh <round> <tag> [stats|pmc|sq|bench ...]   -> gpurun_out/<round>_<tag>/
round=${1:-r05}; tag=${2:-a}; shift; shift
what=${*:-stats pmc sq bench}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${round}_$tag
mkdir -p $O
declare -A CFG=(
  [c2]="--no-secondary --no-cpu"
  [c2_homo]="--homo --no-secondary --no-cpu"
  [c2_gather_mirror]="--workload gather_mirror --no-cpu --steps 60 --warmup 10"
  [c3]="--workload jitc --no-cpu"
  [c3_gather]="--workload jitc --jit-gather --no-cpu --steps 20 --warmup 5"
  [c4]="--workload fcn --no-cpu --steps 60 --warmup 10"
  [c4_homo]="--workload fcn --homo --no-cpu --steps 60 --warmup 10"
  [c5]="--workload dense --no-cpu"
  [c2_rank_of_8]="--emulate-world 8 --steps 100 --warmup 20 --no-cpu --no-secondary"
  [c4_rank_of_8]="--emulate-world 8 --workload fcn --steps 100 --warmup 20 --no-cpu --no-secondary"
  [ref_tuner_point]="--n 500000 --conn 0.004 --fire 0.004 --homo --exact-active --no-secondary --no-cpu"
)
declare -A STEPS=( [c2]=200 [c2_homo]=200 [c2_gather_mirror]=60 [c3]=200 [c3_gather]=20 [c4]=60 [c4_homo]=60 [c5]=200 [c2_rank_of_8]=100 [c4_rank_of_8]=100 [ref_tuner_point]=200 )
ORDER=${ONLY:-"c2 c2_homo c2_gather_mirror c3 c3_gather c4 c4_homo c5 c2_rank_of_8 c4_rank_of_8 ref_tuner_point"}
PMC_ORDER=${ONLY:-"c2 c2_homo c2_gather_mirror c4 c4_homo c5 c2_rank_of_8 c4_rank_of_8"}
if [[ $what == *stats* ]]; then
  for name in $ORDER; do
    rm -rf $O/s_$name
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s_$name -o p -- python3 $R/bench.py ${CFG[$name]} > $O/s_$name.log 2>&1
    echo "stats $name rc=$?"
    { echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py ${CFG[$name]}"
      echo "# med_timed = median over the last ${STEPS[$name]} launches of the kernel (the timed steps); avg_us = rocprofv3's average over all launches"
      python3 $R/tools/summarize_prof.py --last ${STEPS[$name]} "$O/s_$name/p_kernel_trace.csv" | grep -v "at::native\|rocclr\|rocprim" | head -14 | cut -c1-170
      grep -o "\"ms_per_step\": *[0-9.]*" $O/s_$name.log | head -1; } > $O/${name}_kernel_stats.txt
    rm -rf $O/s_$name        # (the traces are tens of MB; the summary is what is kept)
  done
fi
if [[ $what == *pmc* ]]; then
  for name in $PMC_ORDER; do
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf $O/p_${name}_$c
      timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/p_${name}_$c -o p -- python3 $R/bench.py ${CFG[$name]} --steps 12 --warmup 3 > $O/p_${name}_$c.log 2>&1
      echo "pmc $name $c rc=$?"
    done
  done
  { echo "# separate --pmc passes per configuration; FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them, mean per launch"
    echo "# (guide: bytes = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024 on gfx950 for wide coalesced reads)"
    for name in $PMC_ORDER; do
      echo "# $name: bench.py ${CFG[$name]} --steps 12 --warmup 3"
      python3 $R/tools/summarize_prof.py "$O/p_${name}_FETCH_SIZE/*counter_collection.csv" "$O/p_${name}_WRITE_SIZE/*counter_collection.csv" | grep -i "k_plan\|k_bin\|k_compact\|k_dense\|k_mfma\|k_gather\|kernel " | cut -c1-150
    done; } > $O/pmc_all.txt
  python3 $R/tools/make_traffic_json.py $O > $O/traffic.json
fi
if [[ $what == *sq* ]]; then
  for name in c3 c3_gather; do
    for c in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAVES GRBM_GUI_ACTIVE"; do
      t=$(echo $c | tr ' ' '_')
      rm -rf $O/q_${name}_$t
      timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/q_${name}_$t -o p -- python3 $R/bench.py ${CFG[$name]} --steps 6 --warmup 2 > $O/q_${name}_$t.log 2>&1
      echo "sq $name $t rc=$?"
    done
  done
  { echo "# SQ counters of the JITC walks, separate --pmc passes (bench.py ... --steps 6 --warmup 2), mean per launch"
    for name in c3 c3_gather; do
      echo "# $name: bench.py ${CFG[$name]}"
      python3 $R/tools/summarize_prof.py "$O/q_${name}_SQ_INSTS_VALU_SQ_ACTIVE_INST_VALU/*counter_collection.csv" "$O/q_${name}_SQ_WAVE_CYCLES_SQ_BUSY_CYCLES/*counter_collection.csv" "$O/q_${name}_SQ_WAVES_GRBM_GUI_ACTIVE/*counter_collection.csv" | grep -i "k_jit\|kernel " | cut -c1-150
    done; } > $O/c3_sq_counters.txt
  python3 $R/tools/make_sq_json.py $O > $O/sq_counters.json
fi
if [[ $what == *bench* ]]; then
  ( cd $R && timeout -k 10 900 python3 bench.py > $O/bench.log 2>&1 ); echo "bench rc=$?"
  grep '^{' $O/bench.log | tail -1 > $O/bench_line.json
fi
cat $O/*_kernel_stats.txt 2>/dev/null | cut -c1-170
cat $O/pmc_all.txt 2>/dev/null
cat $O/c3_sq_counters.txt 2>/dev/null
tail -c 400 $O/bench_line.json 2>/dev/null
