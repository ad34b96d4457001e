# usage: tools/ab_c4_prof.sh "<flags A>" "<flags B>" ...  — rocprofv3 kernel times of the C4 bench (hetero, homo) per hipcc flag set of
# be_csr_binned.hip; on the GPU box
set -e
export AB_FILE=brainevent_amd/csrc/be_csr_binned.hip
run() { bash tools/prof_fcn.sh ab 2>&1 | grep "k_bin_stream\|k_bin_acc" | cut -c1-60,82-130; }
export -f run
bash tools/ab_build.sh "$@" -- bash -c run
