"""CPU tests: the oracle (numpy + C restatements) is pinned against the golden vectors generated from the
reference's own numpy golden model and against the reference's known-answer tests (tests/golden/)."""
import json
import os

import numpy as np
import pytest

from oracle import oracle_np as O
from oracle import oracle_c

G = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def scalars():
    return json.load(open(os.path.join(G, 'light_rng_scalars.json')))


def test_light_rng_scalars(scalars):
    for x, y in scalars['mix32']:
        assert O.lr_mix32(x) == y
    for x, y in scalars['next']:
        assert O.lr_next(x) == y
    for r, b, y in scalars['bounded']:
        assert O.lr_bounded(r, b) == y
    for s, r, c, l, y in scalars['init']:
        assert O.lr_init(s, r, c, l) == y
    for st, cl, q, s2 in scalars['initial_q']:
        assert O.lr_initial_q(st, cl) == (q, s2)
    for s, r, c, y in scalars['uniform01']:
        assert float(O.lr_uniform01(s, r, c)) == y
    for p, y in scalars['conn_length']:
        assert O.conn_length(p) == y
    for n, y in scalars['default_chunk_size']:
        assert O.default_chunk_size(n) == y


def test_light_rng_hashes_match_the_reference_exact_value_tests(scalars):
    """The reference pins both per-edge hashes with exact values (``brainevent/_numba_random_test.py:58-70`` uniform01,
    exact float32; ``:81-93`` normal01, rtol = atol = 1e-6); the cases are transcribed as data in tests/golden/."""
    t = scalars['uniform01_reference_test']
    assert t['src'] == 'brainevent/_numba_random_test.py:58-70'
    for seed, row, col, expect in t['cases']:
        assert np.float32(O.lr_uniform01(seed, row, col)) == np.float32(expect), (seed, row, col)
    t = scalars['normal01_reference_test']
    assert t['src'] == 'brainevent/_numba_random_test.py:81-93'
    for seed, row, col, expect in t['cases']:
        np.testing.assert_allclose(np.float32(O.lr_normal01(seed, row, col)), np.float32(expect), rtol=1e-6, atol=1e-6)


def test_light_rng_normal01_matches_reference_golden_model(scalars):
    """A grid of ``hash_normal01`` values generated from the reference's own numpy golden model
    (``brainevent/_jit_normal/_test_util.py:10-47``, both Acklam tails included): the restatement is bit-identical."""
    for s, r, c, y in scalars['normal01']:
        assert float(O.lr_normal01(s, r, c)) == y, (s, r, c)
    z = np.array([y for *_, y in scalars['normal01']])
    assert np.isfinite(z).all() and (z < -1.9).any() and (z > 1.9).any()


def test_normal_dense_matches_reference_golden_model():
    """``dense_normal_reference`` (``brainevent/_jit_normal/_test_util.py:50-80``) for every (transpose, corder, mv/mm)."""
    dense = np.load(os.path.join(G, 'jitn_dense.npz'))
    assert len(dense.files) == 16
    for key in dense.files:
        shp, t, c, mm = key.split('_')
        shape = tuple(int(x) for x in shp.split('x'))
        transpose, corder = bool(int(t[1])), bool(int(c[1]))
        Gm = O.jit_generator_matrix('n', np.float32(0.25), np.float32(1.5), 0.2, 123, shape=shape, transpose=transpose,
                                    corder=corder, matrix_mode=mm, dtype=np.float32)
        D = Gm if corder else Gm.T
        assert D.shape == dense[key].shape
        np.testing.assert_array_equal(D, dense[key], err_msg=key)


def test_edges_match_reference_golden_model():
    edges = np.load(os.path.join(G, 'jitc_edges.npz'))
    for key in edges.files:
        parts = dict((p[0], p[1:]) for p in key.replace('st', 'T').split('_'))
        seed, prob, n_rows, n_cols, stride = int(parts['s']), float(parts['p']), int(parts['r']), int(parts['c']), int(parts['T'])
        cs = 1250 if n_cols == 5000 else O.default_chunk_size(n_cols)
        got = np.array(list(O.jit_iter_edges(seed, O.conn_length(prob), n_rows, n_cols, stride, cs)), dtype=np.int32).reshape(-1, 2)
        assert np.array_equal(got, edges[key]), key


def test_uniform_dense_matches_reference_golden_model():
    dense = np.load(os.path.join(G, 'jitu_dense.npz'))
    for key in dense.files:
        shp, t, c, mm = key.split('_')
        shape = tuple(int(x) for x in shp.split('x'))
        transpose, corder = bool(int(t[1])), bool(int(c[1]))
        Gm = O.jit_generator_matrix('u', np.float32(-1.5), np.float32(1.5), 0.2, 123, shape=shape, transpose=transpose,
                                    corder=corder, matrix_mode=mm, dtype=np.float32)
        D = Gm if corder else Gm.T
        assert D.shape == dense[key].shape
        np.testing.assert_array_equal(D, dense[key], err_msg=key)


def _run_kat(k, impl):
    if k['op'] == 'csrmv':
        return impl.binary_csrmv(np.array(k['w'], np.float32), np.array(k['indices'], np.int32), np.array(k['indptr']),
                                 np.array(k['v']), tuple(k['shape']), k['transpose'])
    if k['op'] == 'csrmm':
        return impl.binary_csrmm(np.array(k['w'], np.float32), np.array(k['indices'], np.int32), np.array(k['indptr']),
                                 np.array(k['B']), tuple(k['shape']), k['transpose'])
    if k['op'] == 'densemv':
        return impl.binary_densemv(np.array(k['W'], np.float32), np.array(k['s']), k['transpose'])
    if k['op'] == 'fcnmv':
        return impl.binary_fcnmv(np.array(k['w'], np.float32), np.array(k['indices'], np.int32), np.array(k['s']),
                                 tuple(k['shape']), k['transpose'])
    if k['op'] == 'fcnmm':
        return impl.binary_fcnmm(np.array(k['w'], np.float32), np.array(k['indices'], np.int32), np.array(k['M']),
                                 tuple(k['shape']), k['transpose'])
    if k['op'] == 'float_csrmv':
        return impl.csrmv(np.array(k['w'], np.float32), np.array(k['indices'], np.int32), np.array(k['indptr']),
                          np.array(k['v'], np.float32), tuple(k['shape']), k['transpose'])
    if k['op'] == 'float_csrmm':
        return impl.csrmm(np.array(k['w'], np.float32), np.array(k['indices'], np.int32), np.array(k['indptr']),
                          np.array(k['B'], np.float32), tuple(k['shape']), k['transpose'])
    raise KeyError(k['op'])


def test_known_answer_tests_numpy_oracle():
    for k in json.load(open(os.path.join(G, 'kat.json'))):
        np.testing.assert_allclose(_run_kat(k, O), np.array(k['expect']), rtol=0, atol=0, err_msg=k['src'])


def test_c_oracle_matches_numpy_oracle():
    rng = np.random.default_rng(0)
    for trial in range(6):
        m, k = rng.integers(5, 80, 2)
        lens = rng.integers(0, 12, m)
        ptr = np.concatenate([[0], np.cumsum(lens)])
        idx = rng.integers(0, k, ptr[-1]).astype(np.int32)
        homo = trial % 2 == 0
        w = np.array([0.75], np.float32) if homo else rng.random(ptr[-1]).astype(np.float32)
        for transpose in (True, False):
            n = m if transpose else k
            for v in (rng.random(n) < 0.4, np.where(rng.random(n) < 0.4, 1.5, -0.5).astype(np.float32)):
                a = oracle_c.csrmv_f32(w, idx, ptr, v, (m, k), transpose)
                b = O.binary_csrmv(w, idx, ptr, v, (m, k), transpose)
                np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-6)


def test_c_oracle_known_answers():
    for k in json.load(open(os.path.join(G, 'kat.json'))):
        if k['op'] != 'csrmv':
            continue
        v = np.array(k['v'])
        v = v.astype(np.float32) if v.dtype.kind == 'f' else v
        got = oracle_c.csrmv_f32(np.array(k['w'], np.float32), np.array(k['indices'], np.int32), np.array(k['indptr']), v,
                                 tuple(k['shape']), k['transpose'])
        np.testing.assert_array_equal(got, np.array(k['expect'], np.float32))


def test_c_oracle_jit_and_dense_match_numpy_oracle():
    rng = np.random.default_rng(5)
    for mode, (w0, w1) in {'s': (0.5, 0.0), 'u': (0.1, 0.9), 'n': (0.2, 1.3)}.items():
        for transpose in (False, True):
            for corder in (False, True):
                for stride in (32, 4):
                    shape, prob, seed = (19, 37), 0.25, 11
                    v = rng.random(shape[0] if transpose else shape[1]) < 0.5
                    a = oracle_c.jitmv(mode, w0, w1, prob, v, seed, shape=shape, transpose=transpose, corder=corder, stride=stride)
                    f = O.binary_jitmv if stride == 32 else (lambda *args, **kw: O.binary_jitmm(args[0], args[1], args[2], args[3],
                                                                                               np.asarray(args[4])[:, None], args[5], **kw)[:, 0])
                    b = f(mode, np.float32(w0), np.float32(w1), prob, v, seed, shape=shape, transpose=transpose, corder=corder)
                    tol = 1e-5 if mode == 'n' else 0.0      # logf (libm) vs numpy log differ by ULPs in the tails
                    np.testing.assert_allclose(a, b, rtol=tol, atol=tol)
    W = rng.normal(size=(11, 13)).astype(np.float32)
    for transpose in (True, False):
        s = rng.random(11 if transpose else 13) < 0.5
        np.testing.assert_allclose(oracle_c.densemv_f32(W, s, transpose), O.binary_densemv(W, s, transpose), rtol=1e-6, atol=1e-6)


def test_c_oracle_parallel_variant_equals_serial():
    """The OpenMP + atomics variant timed by bench.py's cpu_baseline (not the reference's algorithm) adds the same numbers."""
    from oracle import oracle_c
    rng = np.random.default_rng(0)
    m, k = 300, 2000
    lens = rng.integers(0, 40, m)
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    idx = rng.integers(0, k, ptr[-1]).astype(np.int32)
    v = rng.random(m) < 0.4
    for w in (np.ones(1, np.float32), rng.integers(1, 8, ptr[-1]).astype(np.float32)):     # integers: order independent
        a = oracle_c.csrmv_t_f32_parallel(w, idx, ptr, v, (m, k), 4)
        b = oracle_c.csrmv_f32(w, idx, ptr, v, (m, k), True)
        np.testing.assert_array_equal(a, b)


def test_fcn_oracle_matches_the_reference_dense_recipe():
    """The reference's forward tests of the ELL ops (``brainevent/_fcn/binary_test.py:202-236``, ``:292-313``) compare
    them with ``dense_from_fixed_conn @ binarised events`` at rtol = atol = 1e-3; the oracle's loops (restating
    ``_fcn/binary.py:156-253``, ``:677-766``) pass the same check."""
    from test_fcn_mm_gpu import fcn_reference_case, fcn_reference_events
    for shape in ((20, 40), (50, 30)):
        for homo in (True, False):
            for replace in (True, False):
                rng = np.random.default_rng(0x5EED)
                w, idx, dense = fcn_reference_case(rng, shape, homo, replace)
                for transpose in (True, False):
                    for as_bool in (True, False):
                        ev = fcn_reference_events(rng, shape[0] if transpose else shape[1], as_bool)
                        b = (ev > 0).astype(np.float64)
                        y = O.binary_fcnmv(w, idx, ev, shape, transpose)
                        np.testing.assert_allclose(y, b @ dense if transpose else dense @ b, rtol=1e-3, atol=1e-3)
                        M = fcn_reference_events(rng, (shape[0] if transpose else shape[1], 10), as_bool)
                        Y = O.binary_fcnmm(w, idx, M, shape, transpose)
                        Bm = (M > 0).astype(np.float64)
                        np.testing.assert_allclose(Y, dense.T @ Bm if transpose else dense @ Bm, rtol=1e-3, atol=1e-3)
