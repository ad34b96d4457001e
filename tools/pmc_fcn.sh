#!/bin/bash
# SQ counters of the C4 bench's kernels (separate --pmc passes; on the GPU box): bash tools/pmc_fcn.sh <tag> [--homo]
tag=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_IFETCH"; do
  d=$R/gpurun_out/pmc_fcn_${tag}_$i
  rm -rf $d
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -o p -- python3 $R/bench.py --workload fcn "$@" --steps 12 --warmup 3 --no-cpu > $d.log 2>&1
  python3 $R/tools/summarize_prof.py "$d/*counter_collection.csv" "$d/*/*counter_collection.csv" 2>/dev/null | grep "k_bin_"
  i=$((i+1))
done
