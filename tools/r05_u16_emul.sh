#!/bin/bash
# VERDICT r4 task 3 (ii), upper bound by a timing build: pass B reading 16-bit columns (6 B / weighted entry instead of 8)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05_u16_emul.txt; : > $O
export AB_FILE=brainevent_amd/csrc/be_csr_binned.hip
run() {
  for args in "--n-post 10000000 --k 1000 --steps 40" "--n-post 10000000 --k 1000 --steps 40 --homo" "" "--homo"; do
    bash tools/prof_any.sh u16 tools/time_binned.py $args 2>&1 | grep "k_bin_stream<float, \(false\|true\), [0-9]*, false, [12]>\|k_bin_acc" | cut -c1-60,82-140
    grep "ms/step" gpurun_out/prof_u16.log
  done
}
export -f run
bash tools/ab_build.sh "" "-DBE_DBG_U16IDX" -- bash -c run 2>&1 | tee -a $O
