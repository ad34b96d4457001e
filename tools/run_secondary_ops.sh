#!/bin/bash
# the secondary operators' micro-benchmarks in one file: bash tools/run_secondary_ops.sh > gpurun_out/secondary_ops.txt  (GPU box)
echo "# tools/bench_gather_rows.py, bench_gather_wide.py, bench_gather_batched.py, bench_dense.py, bench_dense_f32_batches.py, exp_jitmm.py, bench_fcn_dirs.py, bench_plan_build.py"
for t in bench_gather_rows.py bench_gather_wide.py bench_gather_batched.py bench_dense.py bench_dense_f32_batches.py exp_jitmm.py bench_fcn_dirs.py bench_plan_build.py; do
  timeout -k 10 300 python tools/$t 2>&1 | grep -v amdgpu.ids
done
