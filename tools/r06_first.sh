#!/bin/bash
# round 6, first GPU pass: new tests (CUBA step, armed-cache eviction, bench line), network unroll sweep, default bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_graph_capture_gpu.py tests/test_bench_line_gpu.py tests/test_jitc_gpu.py -x -q > gpurun_out/r06_first_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r06_first_tests.log
tail -5 gpurun_out/r06_first_tests.log
for u in 1 4 10 25; do
  BENCH_NETWORK_UNROLL=$u timeout -k 10 300 python - > gpurun_out/r06_unroll_$u.log 2>&1 <<PY
import sys, json
sys.argv=['bench.py']
import bench
for name, f in (('C1_coba','coba_2005.py'),('C1_cuba','cuba_2005.py')):
    r = bench.network_sweep(name, f, steps=20000)
    print(name, 'unroll', $u, json.dumps(r['sweep']))
PY
  cat gpurun_out/r06_unroll_$u.log | tail -3
done
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --full-line-file gpurun_out/r06_a_bench_full_line.json > gpurun_out/r06_a_bench_line.json 2> gpurun_out/r06_a_bench.err
echo "bench rc=$? bytes=$(wc -c < gpurun_out/r06_a_bench_line.json)"
tail -3 gpurun_out/r06_a_bench.err
