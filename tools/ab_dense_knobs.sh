# round 4: compile-time knobs of the C5 MFMA kernel re-measured (K-steps of weight rows in flight per thread: BE_MFMA_D, workgroups:
# BE_MFMA_WG_TARGET, columns per workgroup: BE_MFMA_COLS); C5 at 1 % firing and the dense regime (50 %)
set -e
export AB_FILE=brainevent_amd/csrc/be_dense.hip
run() {
  for f in 0.01 0.5; do
    timeout -k 10 300 python3 bench.py --workload dense --fire $f --no-cpu --no-secondary > gpurun_out/ab_dense.log 2>&1 || { tail -3 gpurun_out/ab_dense.log; return 1; }
    echo "  C5 fire $f: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab_dense.log | head -1) $(grep -o '"kernel_ms": [0-9.]*' gpurun_out/ab_dense.log | head -1)"
  done
}
export -f run
bash tools/ab_build.sh "" "-DBE_MFMA_D=2" "-DBE_MFMA_D=2 -DBE_MFMA_WG_TARGET=512" "-DBE_MFMA_D=2 -DBE_MFMA_WG_TARGET=640" "-DBE_MFMA_D=2 -DBE_MFMA_WG_TARGET=384" "-DBE_MFMA_D=2 -DBE_MFMA_WG_TARGET=1024" "-DBE_MFMA_D=2 -DBE_MFMA_COLS=512" -- bash -c run
