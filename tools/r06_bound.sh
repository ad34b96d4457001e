#!/bin/bash
# what bounds k_plan_accumulate_h8 at C2 (one weight) and k_plan_accumulate_d8<0,1,0> on the 1-of-8 shard — SQ counters
# in separate --pmc passes (8 SQ slots per pass)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_bound
mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_plan_contracts_gpu.py -x -q -k "bad_column or protocol_flag or reproducible" > $O/tests.log 2>&1; echo "binned tests rc=$?"; tail -2 $O/tests.log
cd /tmp && export TMPDIR=/tmp
PA="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
PB="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU"
for pass in A B; do
  eval "CN=\$P$pass"
  timeout -k 10 300 rocprofv3 --pmc $CN --output-format csv -d $O/homo_$pass -o p -- python3 $R/bench.py --homo --steps 10 --warmup 3 --no-cpu --no-secondary > $O/homo_$pass.log 2>&1 || { echo "homo pass $pass failed"; tail -3 $O/homo_$pass.log; }
  timeout -k 10 300 rocprofv3 --pmc $CN --output-format csv -d $O/shard_$pass -o p -- python3 $R/tools/rank_step_lab.py --schedule seq --steps 30 --warmup 5 > $O/shard_$pass.log 2>&1 || { echo "shard pass $pass failed"; tail -3 $O/shard_$pass.log; }
done
cd $R
python3 tools/summarize_prof.py "$O/homo_A/*counter_collection.csv" "$O/homo_B/*counter_collection.csv" 2>/dev/null | grep -i "plan_accumulate\|kernel " > $O/homo_counters.txt
python3 tools/summarize_prof.py "$O/shard_A/*counter_collection.csv" "$O/shard_B/*counter_collection.csv" 2>/dev/null | grep -i "plan_accumulate\|kernel " > $O/shard_counters.txt
cat $O/homo_counters.txt $O/shard_counters.txt
find $O -name '*.csv' -size +4M -delete
# (the dynamic row-ticket experiment of this round — -DBE_PLAN_DYN=<rows> in k_plan_accumulate_d8 — was measured with this script and reverted:
#  profiles/r06_bound_h8_and_shard_d8.txt holds its numbers)
