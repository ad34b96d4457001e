#!/bin/bash
# usage: bash tools/run_suite_n.sh N [pytest args...] — the gpu test suite N times in fresh processes, one log each under gpurun_out/;
# stops at the first run that was killed or timed out (anything but pytest's 0 / 1)
n=${1:-3}; shift
for i in $(seq 1 $n); do
  timeout -k 10 900 python -m pytest tests -q -m gpu -x "$@" > gpurun_out/suite_run_$i.log 2>&1
  rc=$?
  echo "run $i rc=$rc: $(tail -1 gpurun_out/suite_run_$i.log)"
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
done
