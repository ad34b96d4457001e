"""GPU parity for the JIT-connectivity ops against (i) the golden vectors generated from the reference's own
numpy golden model and (ii) the numpy oracle.  Connectivity is integer work: scalar-weight results are exact
(count * w); uniform / normal sums are compared at rtol = atol = 1e-5 (1e-4 for normal: logf/sqrtf ULPs),
the tolerances the reference's own golden-model tests use (brainevent/_jit_uniform/binary_test.py:102-184)."""
import os

import numpy as np
import pytest
import torch

from test_csr_gpu import spikes_of

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')
COMBOS = [(t, c) for t in (False, True) for c in (False, True)]


@pytest.mark.parametrize('transpose,corder', COMBOS)
@pytest.mark.parametrize('shape', [(13, 17), (20, 30)])
def test_jitumv_matches_reference_dense_golden(be, transpose, corder, shape):
    dense = np.load(os.path.join(G, 'jitu_dense.npz'))
    D = dense[f'{shape[0]}x{shape[1]}_t{int(transpose)}_c{int(corder)}_mv'].astype(np.float64)   # [out_len, in_len]
    rng = np.random.default_rng(3)
    for kind in ('bool', 'float'):
        v = spikes_of(rng, D.shape[1], 0.5, kind)
        got = be.binary_jitumv(np.float32(-1.5), np.float32(1.5), 0.2, v, 123, shape=shape, transpose=transpose, corder=corder)
        act = (v > 0) if kind == 'float' else v
        np.testing.assert_allclose(got, D @ act.astype(np.float64), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('transpose,corder', COMBOS)
def test_jitumm_matches_reference_dense_golden(be, transpose, corder):
    shape = (13, 17)
    dense = np.load(os.path.join(G, 'jitu_dense.npz'))
    D = dense[f'13x17_t{int(transpose)}_c{int(corder)}_mm'].astype(np.float64)
    rng = np.random.default_rng(4)
    B = rng.random((D.shape[1], 5)) < 0.5
    got = be.binary_jitumm(np.float32(-1.5), np.float32(1.5), 0.2, B, 123, shape=shape, transpose=transpose, corder=corder)
    np.testing.assert_allclose(got, D @ B.astype(np.float64), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('transpose,corder', COMBOS)
@pytest.mark.parametrize('shape', [(13, 17), (20, 30)])
def test_jitnmv_matches_reference_dense_golden(be, transpose, corder, shape):
    # dense_normal_reference of the reference's golden model (brainevent/_jit_normal/_test_util.py:50-80), tolerance of its
    # own test (brainevent/_jit_normal/binary_test.py:108-160)
    dense = np.load(os.path.join(G, 'jitn_dense.npz'))
    D = dense[f'{shape[0]}x{shape[1]}_t{int(transpose)}_c{int(corder)}_mv'].astype(np.float64)   # [out_len, in_len]
    rng = np.random.default_rng(5)
    for kind in ('bool', 'float'):
        v = spikes_of(rng, D.shape[1], 0.5, kind)
        got = be.binary_jitnmv(np.float32(0.25), np.float32(1.5), 0.2, v, 123, shape=shape, transpose=transpose, corder=corder)
        act = (v > 0) if kind == 'float' else v
        np.testing.assert_allclose(got, D @ act.astype(np.float64), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('transpose,corder', COMBOS)
def test_jitnmm_matches_reference_dense_golden(be, transpose, corder):
    shape = (13, 17)
    dense = np.load(os.path.join(G, 'jitn_dense.npz'))
    D = dense[f'13x17_t{int(transpose)}_c{int(corder)}_mm'].astype(np.float64)
    rng = np.random.default_rng(6)
    B = rng.random((D.shape[1], 5)) < 0.5
    got = be.binary_jitnmm(np.float32(0.25), np.float32(1.5), 0.2, B, 123, shape=shape, transpose=transpose, corder=corder)
    np.testing.assert_allclose(got, D @ B.astype(np.float64), rtol=1e-5, atol=1e-5)


def test_device_hashes_match_the_reference_exact_value_tests(be):
    """The device's per-edge hashes (``be_jit_edge_weights``) against the values the reference's own tests pin:
    ``brainevent/_numba_random_test.py:58-70`` (uniform01, exact float32) and ``:81-93`` (normal01, rtol = atol = 1e-6),
    plus the grid generated from the reference's numpy golden model (both Acklam tails; logf / sqrtf differ by ULPs
    there: rtol 2e-6)."""
    import json
    from brainevent_amd._jitc import jit_edge_weights
    sc = json.load(open(os.path.join(G, 'light_rng_scalars.json')))
    for key, fam, a, b in (('uniform01_reference_test', 'u', 0.0, 1.0), ('normal01_reference_test', 'n', 0.0, 1.0)):
        cases = sc[key]['cases']
        seeds = sorted({c[0] for c in cases})
        for seed in seeds:
            sub = [c for c in cases if c[0] == seed]
            got = jit_edge_weights(fam, a, b, seed, np.array([c[1] for c in sub], np.int32), np.array([c[2] for c in sub], np.int32))
            want = np.array([c[3] for c in sub], np.float32)
            if fam == 'u':
                np.testing.assert_array_equal(got, want)
            else:
                np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)
    for seed in (0, 42, 123):
        sub = [c for c in sc['normal01'] if c[0] == seed]
        got = jit_edge_weights('n', 0.0, 1.0, seed, np.array([c[1] for c in sub], np.int32), np.array([c[2] for c in sub], np.int32))
        np.testing.assert_allclose(got, np.array([c[3] for c in sub], np.float32), rtol=2e-6, atol=1e-6)
        sub = [c for c in sc['uniform01'] if c[0] == seed]
        got = jit_edge_weights('u', 0.0, 1.0, seed, np.array([c[1] for c in sub], np.int32), np.array([c[2] for c in sub], np.int32))
        np.testing.assert_array_equal(got, np.array([c[3] for c in sub], np.float32))


@pytest.mark.parametrize('transpose,corder', COMBOS)
@pytest.mark.parametrize('family', ['s', 'u', 'n'])
@pytest.mark.parametrize('shape,prob', [((40, 70), 0.1), ((100, 50), 0.3), ((3, 5), 0.5), ((64, 257), 1.0)])
def test_jitmv_matches_oracle(be, oracle, family, transpose, corder, shape, prob):
    rng = np.random.default_rng(shape[0] + int(prob * 10))
    in_len = shape[0] if transpose else shape[1]
    v = spikes_of(rng, in_len, 0.4, 'bool')
    seed = 42
    if family == 's':
        got = be.binary_jitsmv(np.float32(0.5), prob, v, seed, shape=shape, transpose=transpose, corder=corder)
        ref = oracle.binary_jitmv('s', 0.5, 0.0, prob, v, seed, shape=shape, transpose=transpose, corder=corder)
        np.testing.assert_array_equal(got, ref.astype(np.float32))          # integer counts * 0.5: exact
    elif family == 'u':
        got = be.binary_jitumv(np.float32(0.1), np.float32(0.9), prob, v, seed, shape=shape, transpose=transpose, corder=corder)
        ref = oracle.binary_jitmv('u', np.float32(0.1), np.float32(0.9), prob, v, seed, shape=shape, transpose=transpose, corder=corder)
        np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)
    else:
        got = be.binary_jitnmv(np.float32(0.2), np.float32(1.3), prob, v, seed, shape=shape, transpose=transpose, corder=corder)
        ref = oracle.binary_jitmv('n', np.float32(0.2), np.float32(1.3), prob, v, seed, shape=shape, transpose=transpose, corder=corder)
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4)
    assert got.dtype == np.float32 and got.shape == ((shape[1],) if transpose else (shape[0],))


@pytest.mark.parametrize('transpose,corder', COMBOS)
@pytest.mark.parametrize('family', ['s', 'u', 'n'])
def test_jitmm_matches_oracle(be, oracle, family, transpose, corder):
    shape, prob, seed, n = (40, 70), 0.15, 7, 6
    rng = np.random.default_rng(9)
    in_len = shape[0] if transpose else shape[1]
    B = np.stack([spikes_of(rng, in_len, 0.4, 'float') for _ in range(n)], axis=1)
    f = {'s': be.binary_jitsmm, 'u': be.binary_jitumm, 'n': be.binary_jitnmm}[family]
    args = {'s': (np.float32(0.5),), 'u': (np.float32(0.1), np.float32(0.9)), 'n': (np.float32(0.2), np.float32(1.3))}[family]
    got = f(*args, prob, B, seed, shape=shape, transpose=transpose, corder=corder)
    w0, w1 = (args + (0.0,))[:2]
    ref = oracle.binary_jitmm(family, w0, w1, prob, B, seed, shape=shape, transpose=transpose, corder=corder)
    tol = 1e-4 if family == 'n' else 1e-5
    np.testing.assert_allclose(got, ref, rtol=tol, atol=tol)


def test_jit_small_prob_many_chunks(be, oracle):
    # conn_prob 0.1 % on a 5000-wide walk: clen = 2000 as at config C3
    shape, prob, seed = (6, 5000), 0.001, 42
    rng = np.random.default_rng(0)
    v = spikes_of(rng, 5000, 0.5, 'bool')
    got = be.binary_jitsmv(np.float32(1.0), prob, v, seed, shape=shape, transpose=False, corder=True)
    ref = oracle.binary_jitmv('s', 1.0, 0.0, prob, v, seed, shape=shape, transpose=False, corder=True)
    np.testing.assert_array_equal(got, ref.astype(np.float32))
    v2 = np.ones(6, bool)
    got2 = be.binary_jitsmv(np.float32(1.0), prob, v2, seed, shape=shape, transpose=True, corder=False)
    ref2 = oracle.binary_jitmv('s', 1.0, 0.0, prob, v2, seed, shape=shape, transpose=True, corder=False)
    np.testing.assert_array_equal(got2, ref2.astype(np.float32))


def test_jit_scatter_large_walk_pieces(be):
    # a walk long enough to need several LDS pieces per residue class (uniform: 16384 accumulators per piece);
    # gather and scatter over the same generator orientation must agree: M (corder=False rows=inputs) vs its transpose
    n_in, n_out, prob, seed = 64, 3_000_000, 0.0005, 11
    rng = np.random.default_rng(1)
    v = spikes_of(rng, n_in, 0.5, 'bool')
    # scatter: shape (n_in, n_out), transpose=True, corder=False -> out[n_out]
    y_s = be.binary_jitsmv(np.float32(2.0), prob, v, seed, shape=(n_in, n_out), transpose=True, corder=False)
    y_u = be.binary_jitumv(np.float32(0.5), np.float32(1.5), prob, v, seed, shape=(n_in, n_out), transpose=True, corder=False)
    assert y_s.shape == (n_out,) and y_u.shape == (n_out,)
    # every edge carries weight 2 (scalar) and a weight in [0.5, 1.5] (uniform) on the same connectivity
    cnt = y_s / 2.0
    assert np.all(cnt == np.round(cnt)) and cnt.sum() > 0
    assert np.all(y_u >= 0.5 * cnt - 1e-4) and np.all(y_u <= 1.5 * cnt + 1e-4)
    # expected number of edges: active rows * n_out * prob
    exp = v.sum() * n_out * prob
    assert abs(cnt.sum() - exp) < 6 * np.sqrt(exp)


def test_jit_prob_zero_and_one(be):
    v = np.ones(9, bool)
    assert not be.binary_jitsmv(np.float32(1.0), 0.0, v, 1, shape=(4, 9), transpose=False, corder=True).any()
    full = be.binary_jitsmv(np.float32(1.0), 1.0, v, 1, shape=(4, 9), transpose=False, corder=True)
    np.testing.assert_array_equal(full, np.full(4, 9.0, np.float32))      # prob = 1 -> cl = 2 -> every column connected


@pytest.mark.parametrize('family', ['s', 'u', 'n'])
def test_jitc_classes_are_consistent(be, oracle, family):
    shape, prob, seed = (30, 45), 0.2, 5
    rng = np.random.default_rng(2)
    cls_r = {'s': be.JITCScalarR, 'u': be.JITCUniformR, 'n': be.JITCNormalR}[family]
    params = {'s': (np.float32(0.5),), 'u': (np.float32(0.1), np.float32(0.9)), 'n': (np.float32(0.2), np.float32(1.3))}[family]
    w0, w1 = (params + (0.0,))[:2]
    tol = 1e-4 if family == 'n' else 1e-5
    for corder in (False, True):
        M = cls_r((*params, prob, seed), shape=shape, corder=corder)
        v = spikes_of(rng, shape[1], 0.5, 'bool')
        s = spikes_of(rng, shape[0], 0.5, 'bool')
        # M @ v : (shape, transpose=False, corder);   s @ M : (shape, transpose=True, not corder)
        np.testing.assert_allclose(M @ be.BinaryArray(v),
                                   oracle.binary_jitmv(family, w0, w1, prob, v, seed, shape=shape, transpose=False, corder=corder),
                                   rtol=tol, atol=tol)
        np.testing.assert_allclose(be.BinaryArray(s) @ M,
                                   oracle.binary_jitmv(family, w0, w1, prob, s, seed, shape=shape, transpose=True, corder=not corder),
                                   rtol=tol, atol=tol)
        # the two directions see the same matrix: s @ (M @ v) == (s @ M) @ v
        lhs = float(np.dot(s.astype(np.float64), M @ be.BinaryArray(v)))
        rhs = float(np.dot((be.BinaryArray(s) @ M).astype(np.float64), v))
        assert abs(lhs - rhs) <= 1e-3 * max(1.0, abs(lhs))
        # transposed container
        Mt = M.T
        assert Mt.shape == shape[::-1] and Mt.corder == (not corder)
        np.testing.assert_allclose(Mt @ be.BinaryArray(s), be.BinaryArray(s) @ M, rtol=tol, atol=tol)
        np.testing.assert_allclose(be.BinaryArray(v) @ Mt, M @ be.BinaryArray(v), rtol=tol, atol=tol)
    with pytest.raises(ValueError):
        cls_r((*params, 1.5, seed), shape=shape)


def test_jit_f64_and_f16_outputs(be, oracle):
    shape, prob, seed = (25, 40), 0.25, 3
    v = np.ones(40, bool)
    ref = oracle.binary_jitmv('u', 0.25, 0.75, prob, v, seed, shape=shape, transpose=False, corder=True, wdtype=np.float64)
    got64 = be.binary_jitumv(np.float64(0.25), np.float64(0.75), prob, v, seed, shape=shape, transpose=False, corder=True)
    assert got64.dtype == np.float64
    np.testing.assert_allclose(got64, ref, rtol=1e-6, atol=1e-6)
    got16 = be.binary_jitumv(np.float16(0.25), np.float16(0.75), prob, v, seed, shape=shape, transpose=False, corder=True)
    assert got16.dtype == np.float16
    np.testing.assert_allclose(got16.astype(np.float64), ref, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize('family', ['s', 'u', 'n'])
@pytest.mark.parametrize('corder', [True, False])
def test_jitmv_mid_size_matches_c_oracle(be, family, corder):
    """Mid-size parity (many chunks per residue class, several workgroups) against the C restatement of the walk."""
    from oracle import oracle_c
    shape, prob, seed = (3000, 5003), 0.02, 77
    rng = np.random.default_rng(8)
    for transpose in (False, True):
        in_len = shape[0] if transpose else shape[1]
        v = spikes_of(rng, in_len, 0.1, 'bool')
        params = {'s': (0.5, 0.0), 'u': (0.1, 0.9), 'n': (0.2, 1.3)}[family]
        ref = oracle_c.jitmv(family, *params, prob, v, seed, shape=shape, transpose=transpose, corder=corder)
        if family == 's':
            got = be.binary_jitsmv(np.float32(0.5), prob, v, seed, shape=shape, transpose=transpose, corder=corder)
            np.testing.assert_array_equal(got, ref.astype(np.float32))
        elif family == 'u':
            got = be.binary_jitumv(np.float32(0.1), np.float32(0.9), prob, v, seed, shape=shape, transpose=transpose, corder=corder)
            np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)
        else:
            got = be.binary_jitnmv(np.float32(0.2), np.float32(1.3), prob, v, seed, shape=shape, transpose=transpose, corder=corder)
            np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('family', ['s', 'u', 'n'])
@pytest.mark.parametrize('cls_kind', ['R', 'C'])
@pytest.mark.parametrize('corder', [False, True])
def test_jitc_materialisation_matches_ops_and_oracle(be, oracle, family, cls_kind, corder):
    """the materialised matrix (CSR or CSC) is the matrix the on-the-fly ops multiply with."""
    shape, prob, seed = (37, 52), 0.2, 13
    cls = getattr(be, {'s': 'JITCScalar', 'u': 'JITCUniform', 'n': 'JITCNormal'}[family] + cls_kind)
    params = {'s': (np.float32(0.5),), 'u': (np.float32(0.1), np.float32(0.9)), 'n': (np.float32(0.2), np.float32(1.3))}[family]
    w0, w1 = (params + (0.0,))[:2]
    M = cls((*params, prob, seed), shape=shape, corder=corder)
    tol = 1e-4 if family == 'n' else 1e-6
    rng = np.random.default_rng(0)
    for mode in ('mv', 'mm'):
        S = M.tocsr(mode)
        assert S.shape == shape and isinstance(S, be.CSR)
        C2 = M.tocsc(mode)
        assert isinstance(C2, be.CSC) and C2.shape == shape
        D = S.todense()
        np.testing.assert_array_equal(C2.todense(), D)
        np.testing.assert_array_equal(M.materialize(mode).todense(), D)
        view = M.mv if mode == 'mv' else M.mm
        np.testing.assert_array_equal(view.todense(), D)
        assert isinstance(view.tocsr(), be.CSR) and isinstance(view.tocsc(), be.CSC)
    with pytest.raises(ValueError, match='ambiguous'):
        M.todense()
        # oracle: D[out, in] of `M @ v`
        if cls_kind == 'R':
            gshape, transpose = shape, False
        else:
            gshape, transpose = shape[::-1], True
        G = oracle.jit_generator_matrix(family, w0, w1, prob, seed, shape=gshape, transpose=transpose, corder=corder,
                                        matrix_mode=mode, dtype=np.float32)
        Dref = G if corder else G.T
        np.testing.assert_allclose(D, Dref, rtol=tol, atol=tol)
    # the stored matrix and the on-the-fly ops agree in both directions (mv matrix)
    S = M.tocsr('mv')
    v = spikes_of(rng, shape[1], 0.5, 'bool')
    s = spikes_of(rng, shape[0], 0.5, 'bool')
    np.testing.assert_allclose(S @ be.BinaryArray(v), M @ be.BinaryArray(v), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(be.BinaryArray(s) @ S, be.BinaryArray(s) @ M, rtol=1e-5, atol=1e-5)


def test_jitc_materialised_scatter_matches_on_the_fly_mid_size(be):
    n, prob, seed = 20000, 0.01, 42
    M = be.JITCScalarR((np.float32(1.0), prob, seed), shape=(n, n), corder=True)
    S = M.tocsr('mv')
    assert abs(S.nse - n * n * prob) < 6 * np.sqrt(n * n * prob)
    rng = np.random.default_rng(4)
    s = spikes_of(rng, n, 0.02, 'bool')
    np.testing.assert_array_equal(be.BinaryArray(s) @ S, be.BinaryArray(s) @ M)      # integer counts: exact


@pytest.mark.parametrize('cls_name,params', [('JITCScalarR', (1.5,)), ('JITCUniformR', (-1.0, 2.0)), ('JITCNormalC', (0.5, 0.3))])
@pytest.mark.parametrize('world', [1, 2, 3, 8])
def test_scatter_shards_tile_the_output(cls_name, params, world):
    """Multi-GPU partition by walk class: every rank's output is the full result on its own columns and zero
    elsewhere (bitwise), the column sets tile the output, nothing is stored."""
    import ctypes
    import brainevent_amd as be
    from brainevent_amd import _lib
    rng = np.random.default_rng(world)
    n_in, n_out = 3000, 5003
    cls = getattr(be, cls_name)
    is_row = cls_name.endswith('R')
    M = cls((*params, 0.02, 77), shape=(n_in, n_out) if is_row else (n_in, n_out), corder=True if is_row else False)
    # pick the operand order that runs the scatter kernel for this class
    s = rng.random(n_in) < 0.2
    try:
        shards = [M.scatter_shard(world, r) for r in range(world)]
    except ValueError:
        M = cls((*params, 0.02, 77), shape=(n_in, n_out), corder=not M.corder)
        shards = [M.scatter_shard(world, r) for r in range(world)]
    full = be.BinaryArray(s) @ M
    n_cls = _lib.fn('be_jit_scatter_classes', ctypes.c_int, [ctypes.c_int64, ctypes.c_int64, ctypes.c_int])(
        shards[0].shape1, shards[0].out_len, 32)
    assert shards[0].n_classes == n_cls and shards[0].out_len == n_out
    seen = np.zeros(n_out, np.int32)
    total = np.zeros(n_out, full.dtype)
    for sh in shards:
        out = be.BinaryArray(s) @ sh
        cols = sh.owned_columns
        seen[cols] += 1
        mask = np.zeros(n_out, bool); mask[cols] = True
        np.testing.assert_array_equal(out[mask], full[mask])
        assert not out[~mask].any()
        total += out
    assert (seen == 1).all()
    np.testing.assert_array_equal(total, full)


@pytest.mark.parametrize('seed', range(int(__import__('os').environ.get('BE_STRESS_SEEDS', 10))))
def test_jit_randomized_against_oracle(be, oracle, seed):
    """Random shapes / probabilities / seeds / families / orientations, vector and batched, against the numpy oracle."""
    rng = np.random.default_rng(4000 + seed)
    shape = (int(rng.integers(1, 90)), int(rng.integers(1, 400)))
    prob = float(rng.choice([0.01, 0.05, 0.2, 0.5, 1.0]))
    rseed = int(rng.integers(0, 2 ** 31 - 1))
    family = 'sun'[seed % 3]
    transpose, corder = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    in_len = shape[0] if transpose else shape[1]
    args = {'s': (np.float32(0.75),), 'u': (np.float32(-0.3), np.float32(1.1)), 'n': (np.float32(0.1), np.float32(0.7))}[family]
    w0, w1 = (args + (0.0,))[:2]
    tol = 1e-4 if family == 'n' else 1e-5
    fmv = {'s': be.binary_jitsmv, 'u': be.binary_jitumv, 'n': be.binary_jitnmv}[family]
    fmm = {'s': be.binary_jitsmm, 'u': be.binary_jitumm, 'n': be.binary_jitnmm}[family]
    for fire in (0.05, 0.7):
        v = spikes_of(rng, in_len, fire, 'bool')
        got = fmv(*args, prob, v, rseed, shape=shape, transpose=transpose, corder=corder)
        ref = oracle.binary_jitmv(family, w0, w1, prob, v, rseed, shape=shape, transpose=transpose, corder=corder)
        np.testing.assert_allclose(got, ref, rtol=tol, atol=tol * max(1.0, float(np.abs(ref).max())))
    nb = int(rng.integers(1, 40))
    B = np.stack([spikes_of(rng, in_len, 0.3, 'bool') for _ in range(nb)], axis=1)
    got = fmm(*args, prob, B, rseed, shape=shape, transpose=transpose, corder=corder)
    ref = oracle.binary_jitmm(family, w0, w1, prob, B, rseed, shape=shape, transpose=transpose, corder=corder)
    np.testing.assert_allclose(got, ref, rtol=tol, atol=tol * max(1.0, float(np.abs(ref).max())))


@pytest.mark.parametrize('family', ['s', 'u', 'n'])
@pytest.mark.parametrize('cls_kind,corder', [('R', False), ('R', True), ('C', False)])
def test_prepare_serves_both_directions_from_the_stored_matrix(be, oracle, family, cls_kind, corder, monkeypatch):
    """`JITC*.prepare()` (round 4): the drawn connectivity is materialised once and `spk @ M` / `M @ spk` (vectors) then run
    event-driven on the stored matrix — the reference's default object walks the whole matrix per step in one of the two
    directions (brainevent/_jit_scalar/main.py:990-1007).  Same numbers as on the fly: exact for the scalar family, 1e-5
    otherwise; `prepare('mm')` does the same for matrix operands (a different draw)."""
    import brainevent_amd._csr as C
    monkeypatch.setattr(C, 'PLAN_MIN_NNZ', 1000)
    monkeypatch.setattr(C, 'AUTO_MIRROR_MIN_NNZ', 1000)
    name = {'s': 'JITCScalar', 'u': 'JITCUniform', 'n': 'JITCNormal'}[family] + cls_kind
    params = {'s': (np.float32(0.5), 0.04, 21), 'u': (np.float32(0.1), np.float32(0.9), 0.04, 22),
              'n': (np.float32(0.3), np.float32(0.8), 0.04, 23)}[family]
    shape = (1100, 1400)
    rng = np.random.default_rng(51)
    fly = getattr(be, name)(params, shape=shape, corder=corder)
    stored = getattr(be, name)(params, shape=shape, corder=corder).prepare('mv').prepare('mm')
    assert stored.buffers['materialized_mv'] is not None and stored.buffers['materialized_mm'] is not None
    assert not any(k.startswith('materialized') for k in stored.T.buffers)          # a transposed object materialises its own
    sr, sc = rng.random(shape[0]) < 0.05, rng.random(shape[1]) < 0.05
    Br, Bc = rng.random((3, shape[0])) < 0.05, rng.random((shape[1], 3)) < 0.05
    cmp = (np.testing.assert_array_equal if family == 's' else
           (lambda a, b: np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-5)))
    cmp(be.BinaryArray(sr) @ stored, be.BinaryArray(sr) @ fly)
    cmp(stored @ be.BinaryArray(sc), fly @ be.BinaryArray(sc))
    cmp(be.BinaryArray(Br) @ stored, be.BinaryArray(Br) @ fly)
    cmp(stored @ be.BinaryArray(Bc), fly @ be.BinaryArray(Bc))
    # both directions of the vector product are event-driven on the stored matrix: a scatter workspace and a mirror exist
    S = stored.buffers['materialized_mv']
    assert S.buffers.get('scatter_plan') is not None and isinstance(S.buffers.get('mirror'), C.Mirror)
    # refused (with a warning) when the stored form does not fit beside what is resident
    monkeypatch.setattr(C, '_free_device_bytes', lambda: 1 << 10)
    with pytest.warns(UserWarning, match='does not fit'):
        tight = getattr(be, name)(params, shape=shape, corder=corder).prepare()
    assert tight.buffers.get('materialized_mv') is None
    cmp(be.BinaryArray(sr) @ tight, be.BinaryArray(sr) @ fly)


def test_scalar_mm_gather_counts_are_bit_sliced_and_exact(be, oracle):
    """Round 4: the scalar mm gather keeps its per-column counts bit-sliced (a ripple-carry add of the edge's column mask);
    32 batch columns at 50 % firing drive the counts through several carries — still the oracle's integers times the weight,
    for the LDS-mask kernel and (one wide chunk) the global-mask kernel, narrow and wide batches."""
    rng = np.random.default_rng(52)
    for shape, prob, n in (((300, 2000), 0.2, 32), ((64, 900), 0.5, 17), ((500, 700), 0.1, 8), ((200, 40000), 0.01, 32)):
        B = rng.random((shape[1], n)) < 0.5
        got = be.binary_jitsmm(np.float32(0.25), prob, B, 99, shape=shape, transpose=False, corder=True)
        ref = oracle.binary_jitmm('s', 0.25, 0.0, prob, B, 99, shape=shape, transpose=False, corder=True)
        np.testing.assert_array_equal(got, ref.astype(np.float32))


@pytest.mark.parametrize('cls_name,params', [('JITCScalarR', (np.float32(0.5),)), ('JITCScalarC', (np.float32(0.5),)),
                                             ('JITCUniformR', (np.float32(0.1), np.float32(0.9))),
                                             ('JITCNormalC', (np.float32(0.2), np.float32(1.1)))])
@pytest.mark.parametrize('world', [1, 3, 8])
def test_gather_shards_by_output_rows_concatenate_to_the_product(cls_name, params, world):
    """Multi-GPU partition of the GATHER orientation (round 4; DESIGN section 7.4 listed it open): every rank computes its own
    output rows from the full spike vector (`be_binary_jitmv_rows`); the slices tile the output and concatenate to the
    unsharded product bit for bit; the side that runs the scatter kernel refuses (it shards by walk class)."""
    import brainevent_amd as be
    from brainevent_amd._dist import post_slice_bounds
    rng = np.random.default_rng(100 + world)
    shape = (2100, 3301)
    M = getattr(be, cls_name)((*params, 0.03, 91), shape=shape, corder=False)
    for side, ev_len, prod in (('left', shape[0], lambda e, X: e @ X), ('right', shape[1], lambda e, X: X @ e)):
        s = rng.random(ev_len) < 0.15
        ev = be.BinaryArray(s)
        try:
            shards = [M.gather_shard(world, r, side) for r in range(world)]
        except ValueError:            # this side runs the scatter kernel for corder=False: the walk-class partition serves it
            with pytest.raises(ValueError):
                be.JITCGatherShard(M, world, 0, side)
            continue
        full = prod(ev, M)
        parts = [prod(ev, sh) for sh in shards]
        assert [len(p) for p in parts] == [post_slice_bounds(len(full), world, r)[1] - post_slice_bounds(len(full), world, r)[0]
                                           for r in range(world)]
        np.testing.assert_array_equal(np.concatenate(parts), full)
        # packed words (what the exchange delivers) are taken as they are
        import torch
        packed = be.BitPackedBinary.from_packed(be.bitpack(torch.from_numpy(s).cuda(), 0).reshape(-1), ev_len)
        np.testing.assert_array_equal(np.asarray(torch.as_tensor(prod(packed, shards[-1])).cpu()), parts[-1])
        assert packed._value is None
    # at least one side of every class is the gather orientation
    assert any(_ok(M, world, side) for side in ('left', 'right'))


def _ok(M, world, side):
    try:
        M.gather_shard(world, 0, side)
        return True
    except ValueError:
        return False


def test_armed_scatter_workspaces_skip_the_zeroing_launch_and_change_no_bit(be, monkeypatch):
    """Round 5: the scatter orientation keeps its workspace armed (`be_jit_scatter_workspace_arm`: spike counters zeroed once, the
    call's last kernel leaves them at zero), so repeated calls issue no zeroing launch.  Same bits as the same call over a fresh,
    UNARMED workspace full of garbage (which the library zeroes per call); spike vectors of very different activity in sequence (a
    stale counter would lengthen or shorten the next call's list); mv and mm; the cache is bounded and disarms what it evicts."""
    import brainevent_amd._jitc as J
    rng = np.random.default_rng(51)
    dev = torch.device('cuda', 0)
    shape, prob, seed = (3000, 2600), 0.01, 5
    M = be.JITCScalarR((np.float32(0.5), prob, seed), shape=shape, corder=False)
    spikes = [torch.from_numpy(rng.random(shape[0]) < fire).to(dev) for fire in (0.5, 0.001, 0.2, 0.0, 0.9)]
    S = torch.from_numpy(rng.random((5, shape[0])) < 0.1).to(dev)            # a batch of event rows: [batch, n_pre] @ M
    armed = [be.BinaryArray(s) @ M for s in spikes] + [be.BinaryArray(S) @ M, be.BinaryArray(S) @ M]
    assert len(J._armed) >= 1
    with monkeypatch.context() as mp:        # per-call workspaces the library has never seen: it zeroes their counters itself
        mp.setattr(J, '_armed_scatter_workspace', lambda n: torch.full((max(int(n), 256),), 0x5a, dtype=torch.uint8, device=dev))
        fresh = [be.BinaryArray(s) @ M for s in spikes] + [be.BinaryArray(S) @ M]
    for a, b in zip(armed, fresh):
        assert torch.equal(a, b)
    assert torch.equal(armed[-1], armed[-2])
    assert float(armed[3].abs().sum()) == 0.0 and float(armed[4].sum()) > float(armed[2].sum()) > float(armed[1].sum())
    # bounded cache: more distinct shapes than it holds; what it evicts is disarmed (and may be handed out again as fresh memory)
    for i in range(J._ARMED_MAX + 3):
        Mi = be.JITCScalarR((np.float32(1.0), 0.02, 3), shape=(500 + 64 * i, 700), corder=False)
        si = torch.from_numpy(rng.random(500 + 64 * i) < 0.1).to(dev)
        assert torch.equal(be.BinaryArray(si) @ Mi, be.BinaryArray(si) @ Mi)
    assert len(J._armed) <= J._ARMED_MAX
    assert torch.equal(be.BinaryArray(spikes[2]) @ M, armed[2])


@pytest.mark.parametrize('fire,n_rows', [(0.02, 150_000), (1.0, 150_000), (1.0, 300_000)])
def test_scalar_scatter_counts_agree_with_the_materialised_matrix_at_any_firing_rate(be, fire, n_rows):
    """The scalar-weight scatter's per-(class, part) counts against the same product over the materialised CSR, exact: few rows
    active, every row of 150 000 active, every row of 300 000 active at prob 0.5 — columns whose count in ONE part passes 65535
    (what a 16-bit partial-sum format would have to detect; round 6 measured that format — half the bytes through the walk kernel's
    epilogue and the reduce — at +-0 / +1.5 us on C3 and dropped it: both are latency-, not byte-bound)."""
    import torch
    k = 4096
    prob = 0.5 if n_rows == 300_000 else 0.02
    M = be.JITCScalarR((np.float32(1.0), prob, 17), shape=(n_rows, k), corder=True)
    g = torch.Generator(device='cuda'); g.manual_seed(3)
    spk = torch.rand(n_rows, device='cuda', generator=g) < fire
    got = be.BinaryArray(spk) @ M
    csr = M.materialize('mv')
    rows = torch.nonzero(spk).flatten()
    ref = torch.zeros(k, dtype=torch.int64, device='cuda')
    ptr = csr.indptr.long()
    for c0 in range(0, rows.numel(), 20000):
        r = rows[c0:c0 + 20000]
        b, ln = ptr[r], ptr[r + 1] - ptr[r]
        tot = int(ln.sum().item())
        off = torch.repeat_interleave(b - torch.cumsum(ln, 0) + ln, ln) + torch.arange(tot, device='cuda')
        ref += torch.bincount(csr.indices[off].long(), minlength=k)
    assert torch.equal(got.double(), ref.double())
    if n_rows == 300_000:
        assert int(ref.max()) > 2 * 65535
