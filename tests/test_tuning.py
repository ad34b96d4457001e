"""Persisted per-architecture tuning of the scatter routes (brainevent_amd._tuning) — the store contract of the reference's
per-GPU hybrid config (brainevent/_csr/hybrid_config.py:229-295: env override > per-device JSON entry > defaults; saving one
device keeps the others; a corrupt file never breaks anything) and that the route choice follows it."""
import dataclasses
import json

import pytest
import torch


@pytest.fixture
def tuning_env(monkeypatch, tmp_path):
    from brainevent_amd import _tuning as T
    path = tmp_path / 'store' / 'scatter_tuning.json'
    monkeypatch.setenv('BRAINEVENT_AMD_TUNING_FILE', str(path))
    monkeypatch.delenv('BRAINEVENT_AMD_TUNING', raising=False)
    monkeypatch.setattr(T, 'current_device_kind', lambda: 'gfx950')
    T.get_scatter_tuning.cache_clear()
    yield T, path
    monkeypatch.delenv('BRAINEVENT_AMD_TUNING_FILE', raising=False)
    monkeypatch.delenv('BRAINEVENT_AMD_TUNING', raising=False)
    monkeypatch.undo()
    T.get_scatter_tuning.cache_clear()
    T.apply_scatter_tuning()              # back to what this process resolves without the test's store


def test_store_round_trip_and_resolution_order(tuning_env, monkeypatch):
    T, path = tuning_env
    import brainevent_amd._csr as C
    assert T.get_scatter_tuning() == T.DEFAULT_SCATTER_TUNING and not path.exists()
    mine = dataclasses.replace(T.DEFAULT_SCATTER_TUNING, plan_min_segment=5, plan_min_segment_homo=7, binned_task_groups=512)
    rec = [{'n': 1000, 'plan_us': 1.5, 'binned_us': 2.5}]
    assert T.save_scatter_tuning(mine, benchmark_records=rec) == path
    got = T.get_scatter_tuning()
    assert (got.plan_min_segment, got.plan_min_segment_homo, got.binned_task_groups) == (5, 7, 512)
    assert got.benchmark_records == tuple(rec)
    assert (C.PLAN_MIN_SEGMENT, C.PLAN_MIN_SEGMENT_HOMO) == (5, 7)          # applied to the route choice
    # another device's entry is kept, and does not leak into this device's resolution
    other = dataclasses.replace(T.DEFAULT_SCATTER_TUNING, plan_min_nnz=123456)
    T.save_scatter_tuning(other, device_kind='gfx942')
    store = json.loads(path.read_text())
    assert set(store) == {'gfx950', 'gfx942'} and store['gfx942']['plan_min_nnz'] == 123456
    assert T.get_scatter_tuning().plan_min_nnz == T.DEFAULT_SCATTER_TUNING.plan_min_nnz
    # the environment override wins over the store
    monkeypatch.setenv('BRAINEVENT_AMD_TUNING', json.dumps({'plan_min_segment': 3}))
    T.get_scatter_tuning.cache_clear()
    assert T.get_scatter_tuning().plan_min_segment == 3
    assert T.get_scatter_tuning().plan_min_segment_homo == T.DEFAULT_SCATTER_TUNING.plan_min_segment_homo
    monkeypatch.delenv('BRAINEVENT_AMD_TUNING')
    # a device without an entry, a corrupt store, a store of the wrong shape: defaults
    monkeypatch.setattr(T, 'current_device_kind', lambda: 'gfx90a')
    T.get_scatter_tuning.cache_clear()
    assert T.get_scatter_tuning() == T.DEFAULT_SCATTER_TUNING
    monkeypatch.setattr(T, 'current_device_kind', lambda: 'gfx950')
    for junk in ('{ not json', '[1, 2, 3]', '{"gfx950": {"plan_min_segment": "many"}}', '{"gfx950": 7}'):
        path.write_text(junk)
        T.get_scatter_tuning.cache_clear()
        assert T.get_scatter_tuning() == T.DEFAULT_SCATTER_TUNING, junk
    # saving over a corrupt store starts a fresh one
    path.write_text('{ not json')
    T.save_scatter_tuning(mine)
    assert json.loads(path.read_text())['gfx950']['plan_min_segment'] == 5


def test_values_are_validated_and_no_device_means_an_explicit_kind(tuning_env, monkeypatch):
    T, path = tuning_env
    with pytest.raises(ValueError):
        T.save_scatter_tuning(dataclasses.replace(T.DEFAULT_SCATTER_TUNING, plan_min_segment=0))
    with pytest.raises(ValueError):
        T.save_scatter_tuning(dataclasses.replace(T.DEFAULT_SCATTER_TUNING, binned_min_tasks=1 << 40))
    monkeypatch.setattr(T, 'current_device_kind', lambda: None)
    with pytest.raises(RuntimeError):
        T.save_scatter_tuning(T.DEFAULT_SCATTER_TUNING)
    assert not path.exists()
    T.save_scatter_tuning(T.DEFAULT_SCATTER_TUNING, device_kind='gfx950')
    assert path.exists()


def test_route_choice_follows_the_tuning(tuning_env):
    T, _ = tuning_env
    from brainevent_amd._csr import choose_scatter_route, ScatterPlan
    w = torch.ones(7)                         # per-entry f32 weights (only numel and dtype matter to the choice)
    m = k = 1_500_000
    nse = m * 1000                            # 96 slices -> ~10 entries per (row, slice) block
    n_slices = -(-k // ScatterPlan.auto_geometry(m, k, nse, False, ScatterPlan.default_shift(k, False))[1])
    per_block = nse / (m * n_slices)
    T.apply_scatter_tuning(dataclasses.replace(T.DEFAULT_SCATTER_TUNING, plan_min_segment=int(per_block) + 2))
    assert choose_scatter_route(nse, m, k, w) != 'plan'
    T.apply_scatter_tuning(dataclasses.replace(T.DEFAULT_SCATTER_TUNING, plan_min_segment=max(1, int(per_block) - 2)))
    assert choose_scatter_route(nse, m, k, w) == 'plan'
    T.apply_scatter_tuning(dataclasses.replace(T.DEFAULT_SCATTER_TUNING, plan_min_nnz=nse + 1))
    assert choose_scatter_route(nse, m, k, w) == 'direct'


@pytest.mark.gpu
def test_tuner_measures_and_the_library_takes_the_task_size(tuning_env):
    """`tune_scatter_routes` at toy sizes (the measurement itself, not its numbers), the store it writes, and
    `be_binned_set_tuning` reaching pass B: every task size gives the same bits (integer sums)."""
    T, path = tuning_env
    import numpy as np
    import brainevent_amd as be
    from brainevent_amd._csr import BinnedScatter
    out = T.tune_scatter_routes(sizes=(60_000,), conn=200, steps=3, save=True)
    assert path.exists() and len(out.benchmark_records) == 2
    assert all('entries_per_block' in r for r in out.benchmark_records)
    assert json.loads(path.read_text())['gfx950']['benchmark_records']
    rng = np.random.default_rng(3)
    m, k, row = 20_000, 300_000, 64
    idx = torch.tensor(rng.integers(0, k, (m, row)).astype(np.int32), device='cuda')
    w = torch.tensor(rng.random((m, row)).astype(np.float32), device='cuda')
    ptr = torch.arange(0, m * row + 1, row, dtype=torch.int32, device='cuda')
    v = torch.tensor(rng.random(m) < 0.1, device='cuda')
    outs = []
    for groups, tasks in ((1024, 2048), (64, 1), (1 << 20, 1 << 20), (7, 3)):
        T.apply_scatter_tuning(dataclasses.replace(T.DEFAULT_SCATTER_TUNING, binned_task_groups=groups, binned_min_tasks=tasks))
        ws = BinnedScatter(w.reshape(-1), m, k, m * row, indices=idx.reshape(-1))
        outs.append(be.binary_csrmv(w.reshape(-1), idx.reshape(-1), ptr, v, shape=(m, k), transpose=True, workspace=ws))
    for o in outs[1:]:
        assert torch.equal(outs[0], o)
    from oracle import oracle_np as O
    ref = O.binary_csrmv(w.cpu().numpy().reshape(-1).astype(np.float64), idx.cpu().numpy().reshape(-1), ptr.cpu().numpy(),
                         v.cpu().numpy(), (m, k), True)
    np.testing.assert_allclose(outs[0].cpu().numpy(), ref, rtol=1e-5, atol=1e-5)


def test_import_with_a_store_present_does_not_touch_the_gpu(tmp_path):
    """importing the package must not initialise the GPU even when a tuning store exists (torchrun ranks import
    before set_device; fork-after-import; exec-before-GPU re-launches).  The per-device entry is resolved at the first route
    choice instead.  Run in a fresh interpreter so that the import really happens."""
    import os
    import subprocess
    import sys
    store = tmp_path / 'scatter_tuning.json'
    store.write_text(json.dumps({'gfx950': {'plan_min_segment': 5}}))
    code = ("import torch, brainevent_amd, brainevent_amd._csr as C\n"
            "assert not torch.cuda.is_initialized(), 'import initialised the GPU'\n"
            "assert C.PLAN_MIN_SEGMENT == 8, C.PLAN_MIN_SEGMENT\n"
            "from brainevent_amd import _tuning as T\n"
            "T.current_device_kind = lambda: 'gfx950'\n"
            "C.choose_scatter_route(10**9, 10**6, 10**6, torch.ones(3))\n"
            "assert C.PLAN_MIN_SEGMENT == 5, C.PLAN_MIN_SEGMENT\n"
            "print('ok')\n")
    env = dict(os.environ, BRAINEVENT_AMD_TUNING_FILE=str(store))
    env.pop('BRAINEVENT_AMD_TUNING', None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-c', code], cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and 'ok' in r.stdout, r.stdout


def test_a_malformed_override_never_breaks_a_later_resolution(monkeypatch):
    """with a bad BRAINEVENT_AMD_TUNING every later get_scatter_tuning() (e.g. from BinnedScatter's
    push_to_library) used to raise; it must warn once and keep the defaults."""
    from brainevent_amd import _tuning as T
    for junk in ('{ not json', '{"plan_min_segment": "many"}', '[1]', '{"plan_min_segment": 0}'):
        monkeypatch.setenv('BRAINEVENT_AMD_TUNING', junk)
        T.get_scatter_tuning.cache_clear()
        with pytest.warns(UserWarning):
            T._warned.clear()
            assert T.get_scatter_tuning() == T.DEFAULT_SCATTER_TUNING
        assert T.get_scatter_tuning() == T.DEFAULT_SCATTER_TUNING           # and again, without raising
        T.ensure_resolved()
    monkeypatch.delenv('BRAINEVENT_AMD_TUNING')
    T.get_scatter_tuning.cache_clear()
