#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for x in "" "--exchange-ahead 1"; do
timeout -k 10 300 python3 bench.py --emulate-world 8 --steps 100 --warmup 20 --no-cpu --no-secondary $x 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('bench $x', d['ms_per_step'], d['rank_breakdown'])"
done
timeout -k 10 200 python3 tools/rank_step_lab.py --pg --schedule ahead_auto --steps 100 --warmup 10 2>&1 | grep "rank 0 of"
