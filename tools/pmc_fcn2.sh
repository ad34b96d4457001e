#!/bin/bash
# stall-side SQ counters of the C4 bench's kernels (separate --pmc passes; on the GPU box): bash tools/pmc_fcn2.sh <tag> [--homo]
tag=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for set in "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_BRANCH" \
           "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH_LEVEL SQ_LEVEL_WAVES"; do
  d=$R/gpurun_out/pmc2_fcn_${tag}_$i
  rm -rf $d
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -o p -- python3 $R/bench.py --workload fcn "$@" --steps 12 --warmup 3 --no-cpu > $d.log 2>&1
  python3 $R/tools/summarize_prof.py "$d/*counter_collection.csv" "$d/*/*counter_collection.csv" 2>/dev/null | grep "k_bin_stream\|k_bin_acc"
  i=$((i+1))
done
