#!/bin/bash
# rehearsal of the multi-rank default command on ONE card over gloo (plumbing only: timings of ranks sharing a card mean nothing)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export BENCH_EXTRAS_FORCE=1 BENCH_EXTRAS_ARGS='--neurons 200000 --k 100' BENCH_BACKEND=gloo
echo "## legs complete"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 10 --warmup 3 --neurons 200000 --no-cpu --full-line-file '' 2>gpurun_out/r06_rehearse_a.err | grep "^{" > gpurun_out/r06_rehearse_a.json; echo "rc=${PIPESTATUS[0]} lines=$(wc -l < gpurun_out/r06_rehearse_a.json) bytes=$(wc -c < gpurun_out/r06_rehearse_a.json)"
python3 -c "
import json; d=json.load(open('gpurun_out/r06_rehearse_a.json')); print('headline', d['value'], d['n_gpus'], d['parity_check']['ok'], d['rank_breakdown']); [print(' ', k, v.get('value'), v.get('parity'), v.get('n_gpus')) for k, v in d['secondary'].items()]; print('extras' in d)"
echo "## rank 1 fails in the first leg"
BENCH_EXTRAS_FAIL=C4_strong:1 BENCH_EXTRAS_SECONDS=60 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 10 --warmup 3 --neurons 200000 --no-cpu --full-line-file '' 2>gpurun_out/r06_rehearse_b.err | grep "^{" > gpurun_out/r06_rehearse_b.json; echo "rc=${PIPESTATUS[0]} lines=$(wc -l < gpurun_out/r06_rehearse_b.json)"
python3 -c "
import json; d=json.load(open('gpurun_out/r06_rehearse_b.json')); print('headline', d['value'], d['parity_check']['ok'], 'extras', d.get('extras'))"
