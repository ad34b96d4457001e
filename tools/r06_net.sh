#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 600 python -m pytest tests/test_graph_capture_gpu.py tests/test_dist_gpu.py -x -q > gpurun_out/r06_net_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r06_net_tests.log
timeout -k 10 300 python - <<'PY' 2>&1 | grep -v amdgpu | tee gpurun_out/r06_net_combined.txt
import sys, json
sys.argv=['bench.py']
import bench
for name, f in (('C1_coba','coba_2005.py'),('C1_cuba','cuba_2005.py')):
    r = bench.network_sweep(name, f, steps=100000)
    print(name, json.dumps(r['sweep']), r['parity_check'])
PY
