# round 4: workgroup target of the JITC scatter re-measured on the final kernel (C3)
set -e
export AB_FILE=brainevent_amd/csrc/be_jitc.hip
run() {
  for i in 1 2; do
    timeout -k 10 300 python3 bench.py --workload jitc --no-cpu --no-secondary > gpurun_out/ab_jitc.log 2>&1 || { tail -3 gpurun_out/ab_jitc.log; return 1; }
    echo "  C3: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab_jitc.log | head -1) $(grep -o '"kernel_ms": [0-9.]*' gpurun_out/ab_jitc.log | head -1)"
  done
}
export -f run
bash tools/ab_build.sh "" "-DBE_JIT_WG_TARGET=128" "-DBE_JIT_WG_TARGET=512" -- bash -c run
