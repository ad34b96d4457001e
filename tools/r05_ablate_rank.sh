#!/bin/bash
# staged ablations of pass B on the post-slice geometry (timing builds: results are garbage above level 0)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export AB_FILE=brainevent_amd/csrc/be_csr_binned.hip
run() { bash tools/prof_any.sh abl tools/time_binned.py "$@" 2>&1 | grep "k_bin_stream<float, false\|k_bin_acc" | grep -v ", 2>" | cut -c1-60,82-140; grep "ms/step" gpurun_out/prof_abl.log; }
export -f run
bash tools/ab_build.sh "" "-DBE_DBG_NOAPPEND" "-DBE_DBG_LEVEL=4" "-DBE_DBG_LEVEL=3" "-DBE_DBG_LEVEL=2" "-DBE_DBG_LEVEL=1" "-DBE_DBG_NOSTORE" -- bash -c run
