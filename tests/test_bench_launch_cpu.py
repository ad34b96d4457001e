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


def _run_extras(env_extra, n=2, expect_rc=0):
    env = dict(os.environ, BENCH_MOCK_STEP='1', OMP_NUM_THREADS='2', BENCH_EXTRAS_FORCE='1',
               BENCH_EXTRAS_ARGS='--neurons 20000 --k 50 --steps 3 --warmup 1 --exchange bytes', **env_extra)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--neurons', '20000', '--steps', '3', '--warmup', '1',
           '--no-cpu', '--exchange', 'bytes', '--full-line-file', '']
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert (r.returncode == 0) == (expect_rc == 0), (r.returncode, r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    assert len(lines[0]) <= 8000
    return json.loads(lines[0])


def test_the_extra_legs_of_a_multi_rank_job_ride_on_the_one_line():
    """The default `--gpus N` job measures C4_strong and C2_weak over the same process group after the headline (rehearsed on CPU
    tensors): both legs arrive compact in `secondary`, with their rank breakdown keys, in ONE line within the driver's 8 KB."""
    d = _run_extras({})
    assert d['n_gpus'] == 2 and d['parity_check']['ok'] and 'extras' not in d
    for leg in ('C4_strong', 'C2_weak'):
        e = d['secondary'][leg]
        assert e['value'] > 0 and e['parity'][1] is True and e['n_gpus'] == 2, e
    assert 'C4_strong' in d['legend']['workloads']


def test_a_leg_that_raises_on_one_rank_costs_the_legs_and_the_status_never_the_headline():
    """Rank 1 fails inside the first extra leg while rank 0 sits in that leg's collectives: rank 1 leaves non-zero without entering
    another collective, the launcher's SIGTERM reaches rank 0 through the wake-up pipe, and rank 0 still prints the complete headline
    with `extras` naming the leg (the advisor's scenario: done.set() used to precede the closing barrier)."""
    d = _run_extras({'BENCH_EXTRAS_FAIL': 'C4_strong:1', 'BENCH_EXTRAS_SECONDS': '120'}, expect_rc=1)
    assert d['n_gpus'] == 2 and d['value'] > 0 and d['parity_check']['ok']
    assert d['extras']['leg'] == 'C4_strong' and d['extras']['incomplete']


def test_a_leg_that_raises_on_rank_zero_prints_the_headline_before_anything_else():
    d = _run_extras({'BENCH_EXTRAS_FAIL': 'C2_weak:0', 'BENCH_EXTRAS_SECONDS': '120'}, expect_rc=1)
    assert d['value'] > 0 and 'C4_strong' in d['secondary'] and 'error' in d['secondary']['C2_weak']
    assert d['extras']['leg'] == 'C2_weak'
