"""`bench.py` end to end on a small problem: the JSON line the driver parses keeps its contract (one line, strict JSON, metric /
value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload,
`roofline` with bound / achieved / peak / unit / frac / traffic, `cpu_baseline` with value / unit / cores / kind / sample), and the
one-rank emulation of a multi-GPU step carries its `rank_breakdown`.  (The full-size line is the driver's BENCH run.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *args], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, f'bench.py must print ONE JSON line, got {len(lines)}'

    def no_const(x):
        raise ValueError(f'non-finite constant {x} in the JSON line')
    return json.loads(lines[0], parse_constant=no_const)


def test_the_line_keeps_the_driver_contract_on_a_small_problem():
    d = _run('--n', '200000', '--steps', '8', '--warmup', '3', '--no-secondary', '--cpu-seconds', '1')
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'parity_check'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 8 and d['warmup'] == 3 and d['higher_is_better'] is True and d['vs_baseline'] is None
    assert d['unit'] == 'Geff/s' and d['dtype'] == 'f32' and d['data'] == 'synthetic' and 'workload' in d['config']
    assert d['value'] > 0 and abs(d['ms_per_step'] - 1e3 * d['config']['mean_active_rows'] * d['config']['n_conn'] / (d['value'] * 1e9)) \
        <= 0.05 * d['ms_per_step']
    roof = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in roof, k
    assert roof['bound'] == 'hbm' and roof['unit'] == 'GB/s' and roof['peak'] == 8000.0 and 0 < roof['frac'] <= 1.0
    cb = d['cpu_baseline']
    assert cb['kind'] == 'port' and cb['cores'] >= 1 and cb['value'] > 0 and 'sample' in cb and cb['unit'] == 'Geff/s'
    assert d['parity_check']['ok'] is True


def test_one_rank_of_eight_reports_where_its_step_goes():
    d = _run('--n', '200000', '--emulate-world', '8', '--steps', '8', '--warmup', '3', '--no-cpu', '--no-secondary')
    assert d['parity_check']['ok'] is True and 'post-slice x8' in d['config']['parallelism']
    rb = d['rank_breakdown']          # compact form: [min, max] over the ranks per quantity (bench.py LEGEND['rank_breakdown'])
    assert rb['schedule'] == 'sequential' and rb['other_schedule'] == 'exchange_ahead_1'
    assert rb['ex'][1] > 0 and rb['sc'][1] > 0 and rb['other'][1] > 0 and rb['step'][0] <= rb['step'][1]
    assert 'rank_breakdown' in d['legend']
    # both schedules once more against an exchange lengthened by 9 us (be_exchange_emulate_latency_us): [us, sequential, pipelined]
    assert rb['emul'][0] == 9 and rb['emul'][1] > 0 and rb['emul'][2] > 0        # (8 steps of a small problem: no ordering asserted)
    d2 = _run('--n', '200000', '--emulate-world', '8', '--steps', '8', '--warmup', '3', '--no-cpu', '--no-secondary', '--exchange-ahead', '1')
    assert d2['parity_check']['ok'] is True and d2['rank_breakdown']['schedule'] == 'exchange_ahead_1'


def test_the_default_line_fits_the_drivers_record_and_every_entry_carries_its_own_check():
    """The default command at the driver's flags: ONE line of at most 8000 bytes (the driver keeps an 8 KB tail of stdout), every
    secondary present with a value, a roofline fraction where one applies and its own parity verdict; the headline's roofline carries
    the read-only ceiling measured in the same run.  (About a minute: the full-size configurations.)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '20', '--warmup', '5', '--cpu-seconds', '2',
                        '--full-line-file', ''], capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1 and len(lines[0].encode()) <= 8000, (len(lines), len(lines[0]) if lines else 0)
    d = json.loads(lines[0])
    assert d['parity_check']['ok'] is True and d['roofline']['read_ceiling_GBps'] > 4000 and 0 < d['roofline']['real_rd'] <= 1.0
    sec = d['secondary']
    for name in ('C2_gather_mirror', 'C3', 'C3_gather', 'C4', 'C4_homo', 'C5', 'C2_homo', 'ref_tuner_point', 'C2_rank_of_8',
                 'C4_rank_of_8', 'C1_coba', 'C1_cuba'):
        assert name in sec and 'error' not in sec[name], (name, sec.get(name))
        assert sec[name]['value'] > 0 and sec[name]['parity'][1] is True, (name, sec[name])
        assert name in d['legend']['workloads']
    for name in ('C1_coba', 'C1_cuba'):
        assert [row[0] for row in sec[name]['sweep']] == [1, 10, 100]
