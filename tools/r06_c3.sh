#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 600 python -m pytest tests/test_jitc_gpu.py tests/test_float_gpu.py -x -q > gpurun_out/r06_c3_tests.log 2>&1; echo "jitc tests rc=$?"; tail -3 gpurun_out/r06_c3_tests.log
timeout -k 10 600 python -m pytest tests/test_full_size_gpu.py -x -q -k c3 >> gpurun_out/r06_c3_tests.log 2>&1; echo "full-size c3 rc=$?"; tail -2 gpurun_out/r06_c3_tests.log
for i in 1 2; do timeout -k 10 300 python bench.py --workload jitc --steps 100 --warmup 20 --no-cpu --full-line --full-line-file '' 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('C3', d['value'], d['ms_per_step'], d['kernel_ms'], d['step_ms_hip_events'], d.get('parity_check'))"; done | tee gpurun_out/r06_c3_bench.txt
bash tools/prof_any.sh r06_c3 bench.py --workload jitc --steps 60 --warmup 10 --no-cpu | tee -a gpurun_out/r06_c3_bench.txt
