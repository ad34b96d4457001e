# round 4: compile-time knobs of the binned route re-measured on the final kernel — loads in flight per wave of pass B
# (BE_STREAM_U), groups in flight per thread of pass C (BE_BIN_U)
set -e
export AB_FILE=brainevent_amd/csrc/be_csr_binned.hip
run() {
  for a in "--emulate-world 8 --workload fcn --steps 300 --warmup 50" "--workload fcn --steps 100 --warmup 20" "--workload fcn --homo --steps 100 --warmup 20"; do
    timeout -k 10 300 python3 bench.py $a --no-cpu --no-secondary > gpurun_out/ab_knobs.log 2>&1 || { tail -3 gpurun_out/ab_knobs.log; return 1; }
    echo "  $a: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab_knobs.log | head -1)"
  done
}
export -f run
bash tools/ab_build.sh "" "-DBE_STREAM_U=1" "-DBE_STREAM_U=3" "-DBE_BIN_U=2" "-DBE_BIN_U=8" -- bash -c run
