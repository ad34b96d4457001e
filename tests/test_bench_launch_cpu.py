"""`python bench.py --gpus N` launches its own ranks (a child torch.distributed.run, no exec) and prints ONE JSON line.

Rehearsed here on CPU tensors over gloo with BENCH_MOCK_STEP=1: the launcher, the post-slice partition of one global
matrix, the spike exchange, the max / sum reductions and the JSON line are the real code; the scatter itself is replaced by
a torch index_add (no GPU in this container), which the line says (`"mock_step": true`)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, n):
    env = dict(os.environ, BENCH_MOCK_STEP='1', OMP_NUM_THREADS='2')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--neurons', '20000', '--steps', '3', '--warmup', '1',
           '--no-cpu', '--no-secondary'] + extra
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 prints exactly one line
    return json.loads(lines[0])


def test_bench_launches_two_ranks_by_itself():
    d = _run([], 2)
    assert d['n_gpus'] == 2 and d['scaling'] == 'strong' and d['mock_step'] is True
    assert d['steps'] == 3 and d['warmup'] == 1 and d['higher_is_better'] is True and d['unit'] == 'Geff/s'
    pc = d['parity_check']
    assert pc['ok'] and pc['stored_synapses_all_ranks'] == pc['expected_stored_synapses'] == 20000 * 200
    assert 'post-slice x2' in d['config']['parallelism'] and d['config']['n_post_per_gpu'] == 10000
    assert d['value'] > 0 and d['ms_per_step'] > 0


def test_bench_launches_the_fixed_number_workload_over_three_ranks():
    """C4's shape (FixedNumPerPre as a CSR of equal rows) cut three ways: ragged word-aligned slices of the pre population."""
    d = _run(['--workload', 'fcn', '--k', '50', '--exchange', 'bytes'], 3)
    assert d['n_gpus'] == 3 and d['scaling'] == 'strong'
    pc = d['parity_check']
    assert pc['ok'] and pc['stored_synapses_all_ranks'] == pc['expected_stored_synapses'] == 20000 * 50
