#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for q in 4 8; do
GPU_MAX_HW_QUEUES=$q timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 --no-cpu --full-line-file '' 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); s=d['secondary']; print('GPU_MAX_HW_QUEUES=$q', d['value'], 'C2r', s['C2_rank_of_8']['rank_breakdown'], 'C4r', s['C4_rank_of_8']['rank_breakdown'], 'C1', s['C1_coba']['sweep'])"
done
