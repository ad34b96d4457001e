"""GPU parity: batched CSR products (binary_csrmm) and fixed-number connectivity (binary_fcnmv / binary_fcnmm)."""
import numpy as np
import pytest
import torch

from test_csr_gpu import rand_csr, spikes_of, RTOL, ATOL

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('transpose', [True, False])
@pytest.mark.parametrize('homo', [True, False])
@pytest.mark.parametrize('kind', ['bool', 'float'])
def test_csrmm_matches_oracle(be, oracle, transpose, homo, kind):
    rng = np.random.default_rng(21)
    m, k, n = 150, 220, 7
    w, idx, ptr = rand_csr(rng, m, k, rng.integers(0, 30, m), homo=homo)
    rows = m if transpose else k
    B = np.stack([spikes_of(rng, rows, 0.3, kind) for _ in range(n)], axis=1)
    got = be.binary_csrmm(w, idx, ptr, B, shape=(m, k), transpose=transpose)
    ref = oracle.binary_csrmm(w.astype(np.float64), idx, ptr, B, (m, k), transpose)
    assert got.shape == ((k if transpose else m), n)
    np.testing.assert_allclose(got, ref, rtol=RTOL, atol=ATOL)


def test_csrmm_planned_batch(be, oracle, monkeypatch):
    import brainevent_amd._csr as C
    monkeypatch.setattr(C, 'PLAN_MIN_NNZ', 1000)
    rng = np.random.default_rng(22)
    m, k, n = 400, 40000, 5
    w, idx, ptr = rand_csr(rng, m, k, [200] * m)
    csr = be.CSR((w, idx, ptr), shape=(m, k))
    S = np.stack([spikes_of(rng, m, 0.2, 'bool') for _ in range(n)], axis=0)      # [n, m] @ csr -> [n, k]
    got = be.BinaryArray(S) @ csr
    assert isinstance(csr.buffers['scatter_plan'], C.ScatterPlan) and got.shape == (n, k)
    ref = oracle.binary_csrmm(w.astype(np.float64), idx, ptr, S.T, (m, k), True).T
    np.testing.assert_allclose(got, ref, rtol=RTOL, atol=ATOL)
    S2 = np.stack([spikes_of(rng, k, 0.2, 'bool') for _ in range(n)], axis=1)     # csr @ [k, n] -> [m, n]
    got2 = csr @ be.BinaryArray(S2)
    np.testing.assert_allclose(got2, oracle.binary_csrmm(w.astype(np.float64), idx, ptr, S2, (m, k), False), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize('transpose', [True, False])
@pytest.mark.parametrize('homo', [True, False])
@pytest.mark.parametrize('kind', ['bool', 'float'])
@pytest.mark.parametrize('n_conn', [1, 5, 64, 130])
def test_fcnmv_matches_oracle(be, oracle, transpose, homo, kind, n_conn):
    rng = np.random.default_rng(n_conn)
    n_pre, n_post = 211, 333
    idx = rng.integers(0, n_post, (n_pre, n_conn)).astype(np.int32)
    w = np.asarray([0.7], np.float32) if homo else rng.uniform(0.1, 1, (n_pre, n_conn)).astype(np.float32)
    s = spikes_of(rng, n_pre if transpose else n_post, 0.3, kind)
    got = be.binary_fcnmv(w, idx, s, shape=(n_pre, n_post), transpose=transpose)
    ref = oracle.binary_fcnmv(w.astype(np.float64), idx, s, (n_pre, n_post), transpose)
    np.testing.assert_allclose(got, ref, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize('transpose', [True, False])
def test_fcnmm_matches_oracle(be, oracle, transpose):
    rng = np.random.default_rng(5)
    n_pre, n_post, n_conn, n = 90, 120, 17, 6
    idx = rng.integers(0, n_post, (n_pre, n_conn)).astype(np.int32)
    w = rng.uniform(0.1, 1, (n_pre, n_conn)).astype(np.float32)
    M = np.stack([spikes_of(rng, n_pre if transpose else n_post, 0.4, 'float') for _ in range(n)], axis=1)
    got = be.binary_fcnmm(w, idx, M, shape=(n_pre, n_post), transpose=transpose)
    ref = oracle.binary_fcnmm(w.astype(np.float64), idx, M, (n_pre, n_post), transpose)
    np.testing.assert_allclose(got, ref, rtol=RTOL, atol=ATOL)


def test_fcn_docstring_kats(be):
    # brainevent/_fcn/binary.py:124-130 and :646-654
    idx = np.array([[0, 1], [1, 2]], np.int32)
    w = np.array([1.0], np.float32)
    out = be.binary_fcnmv(w, idx, np.array([True, False, True]), shape=(2, 3), transpose=False)
    np.testing.assert_array_equal(out, [1.0, 1.0])
    M = np.array([[True, False], [False, True], [True, True]])
    out = be.binary_fcnmm(w, idx, M, shape=(2, 3), transpose=False)
    np.testing.assert_array_equal(out, [[1, 1], [1, 2]])


def test_fixed_num_classes(be, oracle, monkeypatch):
    import brainevent_amd._csr as C
    monkeypatch.setattr(C, 'PLAN_MIN_NNZ', 1000)
    rng = np.random.default_rng(8)
    n_pre, n_post, n_conn = 300, 40000, 128
    idx = rng.integers(0, n_post, (n_pre, n_conn)).astype(np.int32)
    w = rng.uniform(0.1, 1, (n_pre, n_conn)).astype(np.float32)
    conn = be.FixedNumPerPre((w, idx), shape=(n_pre, n_post))
    s = spikes_of(rng, n_pre, 0.2, 'bool')
    got = be.BinaryArray(s) @ conn                      # favourable: scatter (planned)
    assert isinstance(conn.buffers['scatter_plan'], C.ScatterPlan)
    np.testing.assert_allclose(got, oracle.binary_fcnmv(w.astype(np.float64), idx, s, (n_pre, n_post), True), rtol=RTOL, atol=ATOL)
    s2 = spikes_of(rng, n_post, 0.2, 'bool')
    got2 = conn @ be.BinaryArray(s2)                    # unfavourable: gather
    np.testing.assert_allclose(got2, oracle.binary_fcnmv(w.astype(np.float64), idx, s2, (n_pre, n_post), False), rtol=RTOL, atol=ATOL)
    post = conn.T                                       # FixedNumPerPost, shape (n_post, n_pre)
    assert isinstance(post, be.FixedNumPerPost) and post.shape == (n_post, n_pre)
    np.testing.assert_allclose(post @ be.BinaryArray(s), got, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(be.BinaryArray(s2) @ post, got2, rtol=RTOL, atol=ATOL)
    # dense equivalence
    np.testing.assert_allclose(got, s.astype(np.float32) @ conn.todense(), rtol=1e-4, atol=1e-4)
    # 2-D operands
    S = np.stack([spikes_of(rng, n_pre, 0.2, 'bool') for _ in range(3)], axis=0)
    np.testing.assert_allclose(be.BinaryArray(S) @ conn, S.astype(np.float32) @ conn.todense(), rtol=1e-4, atol=1e-4)
    with pytest.raises(ValueError):
        be.FixedNumPerPre((w, idx + n_post), shape=(n_pre, n_post))


@pytest.mark.parametrize('homo', [True, False])
@pytest.mark.parametrize('wdtype', ['float32', 'float16'])
@pytest.mark.parametrize('nb,spike_kind', [(4, 'bool'), (32, 'bool'), (37, 'float'), (70, 'bool')])
def test_csrmm_gather_fused_over_batch_long_rows(homo, wdtype, nb, spike_kind):
    """binary_csrmm transpose=False with long rows runs the kernel that is fused over the batch (one pass over the
    matrix per 32 columns); rows of very different lengths, empty rows, more than 32 columns."""
    import brainevent_amd as be
    from oracle import oracle_np as O
    rng = np.random.default_rng(nb + homo)
    m, k = 37, 9000
    lens = rng.integers(300, 2500, m); lens[3] = 0; lens[10] = 1; lens[20] = 5000
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    idx = rng.integers(0, k, ptr[-1]).astype(np.int32)
    w = (np.array([0.75]) if homo else rng.random(ptr[-1])).astype(wdtype)
    B = rng.random((k, nb)) < 0.05
    Bv = B if spike_kind == 'bool' else np.where(B, 1.5, -0.5).astype(np.float32)
    got = be.binary_csrmm(w, idx, ptr, Bv, shape=(m, k), transpose=False)
    ref = O.binary_csrmm(w.astype(np.float32), idx, ptr, B, (m, k), False)
    tol = 1e-5 if wdtype == 'float32' else 2e-2
    np.testing.assert_allclose(np.asarray(got, np.float32), ref, rtol=tol, atol=tol * 10)
    # the class route (CSR @ B) gives the same numbers
    got2 = be.CSR((w, idx, ptr), shape=(m, k)) @ be.BinaryArray(Bv)
    np.testing.assert_allclose(np.asarray(got2, np.float32), np.asarray(got, np.float32), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('seed', range(int(__import__('os').environ.get('BE_STRESS_SEEDS', 8))))
def test_fixed_num_randomized(be, oracle, monkeypatch, seed):
    """Random FixedNumPerPre / PerPost matrices through the class surface: all routes (direct, planned d8 / u16, binned),
    both operand orders, vectors and batches, plain / bit-packed / compacted events, against the oracle's dense view."""
    import brainevent_amd._csr as C
    rng = np.random.default_rng(5000 + seed)
    monkeypatch.setattr(C, 'PLAN_MIN_NNZ', int(rng.choice([1, 10 ** 9])))       # plan / binned when they apply, or direct
    n_pre, n_post = int(rng.integers(1, 500)), int(rng.choice([5, 300, 20000, 90000]))
    n_conn = int(rng.choice([1, 7, 64, 300]))
    homo = bool(seed & 1)
    idx = rng.integers(0, n_post, (n_pre, n_conn)).astype(np.int32)
    w = np.array([0.25], np.float32) if homo else rng.uniform(-1, 1, (n_pre, n_conn)).astype(np.float32)
    conn = be.FixedNumPerPre((w, idx), shape=(n_pre, n_post))
    dense = conn.todense().astype(np.float64)
    tol = dict(rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(dense).sum(axis=0).max())))
    for fire in (0.1, 0.9):
        s = spikes_of(rng, n_pre, fire, 'bool')
        ref = s.astype(np.float64) @ dense
        np.testing.assert_allclose(be.BinaryArray(s) @ conn, ref, **tol)
        np.testing.assert_allclose(be.BinaryArray(s).bitpack() @ conn, ref, **tol)
        np.testing.assert_allclose(be.CompactBinary.from_array(s) @ conn, ref, **tol)
        np.testing.assert_allclose(conn.T @ be.BinaryArray(s), ref, **tol)
        s2 = spikes_of(rng, n_post, fire, 'float')
        ref2 = dense @ (s2 > 0).astype(np.float64)
        tol2 = dict(rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(dense).sum(axis=1).max())))
        np.testing.assert_allclose(conn @ be.BinaryArray(s2), ref2, **tol2)
        np.testing.assert_allclose(be.BinaryArray(s2) @ conn.T, ref2, **tol2)
    S = np.stack([spikes_of(rng, n_pre, 0.3, 'bool') for _ in range(int(rng.integers(1, 6)))], axis=0)
    np.testing.assert_allclose(be.BinaryArray(S) @ conn, S.astype(np.float64) @ dense, **tol)


def test_fixed_num_conversions(be):
    """fromdense (uniform, padded, error cases) and tocsr / tocsc (reference ``_fcn/main.py:857-897``, ``:1118-1160``)."""
    rng = np.random.default_rng(3)
    dense = np.zeros((6, 9), np.float32)
    for r in range(6):
        dense[r, rng.choice(9, 3, replace=False)] = rng.uniform(0.5, 1.5, 3)
    pre = be.FixedNumPerPre.fromdense(dense)
    assert pre.shape == (6, 9) and tuple(pre.indices.shape) == (6, 3)
    np.testing.assert_array_equal(pre.todense(), dense)
    np.testing.assert_array_equal(pre.tocsr().todense(), dense)
    np.testing.assert_array_equal(pre.tocsc().todense(), dense)
    assert isinstance(pre.tocsr(), be.CSR) and isinstance(pre.tocsc(), be.CSC)
    dense[0, :] = 0; dense[0, 4] = 2.0                                  # now ragged
    with pytest.raises(ValueError, match='non-uniform'):
        be.FixedNumPerPre.fromdense(dense)
    with pytest.raises(ValueError, match='too small'):
        be.FixedNumPerPre.fromdense(dense, num_conn=2)
    padded = be.FixedNumPerPre.fromdense(dense, num_conn=4)             # zero-weight sentinel at index 0
    np.testing.assert_array_equal(padded.todense(), dense)
    s = rng.random(6) < 0.6
    np.testing.assert_allclose(be.BinaryArray(s) @ padded, s.astype(np.float32) @ dense, rtol=1e-6, atol=1e-6)
    post = be.FixedNumPerPost.fromdense(dense, num_conn=5)
    assert post.shape == (6, 9) and tuple(post.indices.shape) == (9, 5)
    np.testing.assert_array_equal(post.todense(), dense)
    np.testing.assert_array_equal(post.tocsr().todense(), dense)
    np.testing.assert_array_equal(post.tocsc().todense(), dense)
    with pytest.raises(ValueError):
        be.FixedNumPerPre.fromdense(dense[0])


# ---------------------------------------------------------------------------------------------------
# the reference's own forward tests of the ELL ops, restated: brainevent/_fcn/binary_test.py:202-236 (matvec) and :292-313
# (matmat) compare the op with `dense_from_fixed_conn(weights, indices) @ binarised events` (`_mv_reference` /
# `_mm_reference`, :157-190) at rtol = atol = 1e-3, for shapes (20, 40), (50, 30) [CPU] / (400, 200) [GPU],
# n_conn = max(1, int(n * 0.1)), weights [1.5] or N(0, 1), bool events (rand < 0.5) or floats where(raw > 0.4, raw, 0),
# indices drawn with or without replacement.  The dense product below is built independently of the oracle's loops.
# ---------------------------------------------------------------------------------------------------
def fcn_reference_case(rng, shape, homo, replace):
    m, n = shape
    n_conn = max(1, int(n * 0.1))
    if replace:
        idx = rng.integers(0, n, (m, n_conn)).astype(np.int32)
    else:
        idx = np.stack([rng.choice(n, size=n_conn, replace=False) for _ in range(m)]).astype(np.int32)
    w = np.asarray([1.5], np.float32) if homo else rng.normal(0.0, 1.0, idx.shape).astype(np.float32)
    dense = np.zeros(shape, np.float64)
    np.add.at(dense, (np.repeat(np.arange(m), n_conn), idx.reshape(-1)),
              np.broadcast_to(w.astype(np.float64).reshape(-1), (idx.size,)) if homo else w.astype(np.float64).reshape(-1))
    return w, idx, dense


def fcn_reference_events(rng, shape, as_bool):
    if as_bool:
        return rng.random(shape) < 0.5
    raw = rng.random(shape).astype(np.float32)
    return np.where(raw > 0.4, raw, np.float32(0.0))


@pytest.mark.parametrize('homo', [True, False])
@pytest.mark.parametrize('transpose', [True, False])
@pytest.mark.parametrize('as_bool', [True, False])
@pytest.mark.parametrize('shape', [(20, 40), (50, 30), (400, 200)])
@pytest.mark.parametrize('replace', [True, False])
def test_fcn_forward_matches_the_reference_dense_recipe(be, homo, transpose, as_bool, shape, replace):
    rng = np.random.default_rng(0x5EED)
    w, idx, dense = fcn_reference_case(rng, shape, homo, replace)
    m, n = shape
    ev = fcn_reference_events(rng, m if transpose else n, as_bool)
    y = be.binary_fcnmv(w, idx, ev, shape=shape, transpose=transpose)
    b = (ev > 0).astype(np.float64)
    np.testing.assert_allclose(y, b @ dense if transpose else dense @ b, rtol=1e-3, atol=1e-3)
    M = fcn_reference_events(rng, (m if transpose else n, 10), as_bool)
    Y = be.binary_fcnmm(w, idx, M, shape=shape, transpose=transpose)
    Bm = (M > 0).astype(np.float64)
    np.testing.assert_allclose(Y, dense.T @ Bm if transpose else dense @ Bm, rtol=1e-3, atol=1e-3)
