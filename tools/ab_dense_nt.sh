# round 4: ring depth of the no-transpose MFMA kernel (W @ S.T) and the other dense variants, tools/bench_dense.py
set -e
export AB_FILE=brainevent_amd/csrc/be_dense.hip
run() { timeout -k 10 300 python3 tools/bench_dense.py 2>&1 | grep -v amdgpu | cut -c1-200; }
export -f run
bash tools/ab_build.sh "" "-DBE_NT_RING=4" "-DBE_NT_RING=12" -- bash -c run
