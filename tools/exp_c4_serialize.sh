#!/bin/bash
# Experiment (round 4): the C4 step is 11-15 % faster under rocprofv3 --kernel-trace than bare.  Which part of what the
# profiler does to the queue is it?  bare / AMD_SERIALIZE_KERNEL=3 (HIP waits around every launch) / rocprofv3 --kernel-trace
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
B="$R/bench.py --workload fcn --steps 100 --warmup 20 --no-cpu --no-secondary"
get() { grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}' $1 | head -2 | tr '\n' ' '; }
python3 $B > $R/gpurun_out/ser_bare.log 2>&1; echo "bare:                 $(get $R/gpurun_out/ser_bare.log)"
AMD_SERIALIZE_KERNEL=3 python3 $B > $R/gpurun_out/ser_ser3.log 2>&1; echo "AMD_SERIALIZE_KERNEL=3: $(get $R/gpurun_out/ser_ser3.log)"
AMD_SERIALIZE_KERNEL=1 python3 $B > $R/gpurun_out/ser_ser1.log 2>&1; echo "AMD_SERIALIZE_KERNEL=1: $(get $R/gpurun_out/ser_ser1.log)"
rm -rf /tmp/ser_prof; rocprofv3 --kernel-trace --output-format csv -d /tmp/ser_prof -o p -- python3 $B > $R/gpurun_out/ser_prof.log 2>&1; echo "rocprofv3 --kernel-trace: $(get $R/gpurun_out/ser_prof.log)"
python3 $B > $R/gpurun_out/ser_bare2.log 2>&1; echo "bare again:           $(get $R/gpurun_out/ser_bare2.log)"
