"""Float-operand twins (last clause): ``csrmv`` / ``csrmm`` / ``fcnmv`` / ``fcnmm`` and the ``@`` operator of the
containers with a dense operand, against the oracle's restatement of the reference's CPU loops
(``brainevent/_csr/float.py:153-207``, ``:670-744``; ``brainevent/_fcn/float.py``).  Tolerances: f32 rtol = atol = 1e-5 scaled by
the largest output (the sums are f32 here, f64 in the oracle), f64 1e-10, f16 / bf16 at their own precision."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')


def rand_csr(rng, m, k, lens, dtype=np.float32, homo=False):
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    idx = rng.integers(0, max(k, 1), int(ptr[-1])).astype(np.int32)
    w = np.array([0.75], dtype) if homo else rng.normal(0, 1, int(ptr[-1])).astype(dtype)
    return w, idx, ptr


def close(got, ref, dtype):
    tol = {np.float32: 1e-5, np.float64: 1e-10, np.float16: 4e-3}[dtype]
    scale = max(1.0, float(np.abs(np.asarray(ref, np.float64)).max()) if np.size(ref) else 1.0)
    np.testing.assert_allclose(np.asarray(got, np.float64), np.asarray(ref, np.float64), rtol=tol, atol=tol * scale)


def test_reference_known_answers(be):
    """The two hand-checkable vectors of the reference's own float tests (``_csr/float_test.py:111-120``, ``:134-144``)."""
    for k in json.load(open(os.path.join(G, 'kat.json'))):
        if k['op'] == 'float_csrmv':
            got = be.csrmv(np.array(k['w'], np.float32), np.array(k['indices'], np.int32), np.array(k['indptr'], np.int64),
                           np.array(k['v'], np.float32), shape=tuple(k['shape']), transpose=k['transpose'])
        elif k['op'] == 'float_csrmm':
            got = be.csrmm(np.array(k['w'], np.float32), np.array(k['indices'], np.int32), np.array(k['indptr'], np.int64),
                           np.array(k['B'], np.float32), shape=tuple(k['shape']), transpose=k['transpose'])
        else:
            continue
        np.testing.assert_array_equal(got, np.array(k['expect'], np.float32), err_msg=k['src'])


@pytest.mark.parametrize('dtype', [np.float32, np.float64, np.float16])
@pytest.mark.parametrize('homo', [False, True])
@pytest.mark.parametrize('transpose', [False, True])
@pytest.mark.parametrize('style', ['tiny', 'short', 'medium', 'long', 'ragged'])
def test_csrmv_against_the_oracle(be, oracle, dtype, homo, transpose, style):
    rng = np.random.default_rng(hash((style, homo, transpose)) % 2 ** 31)
    m, k = {'tiny': (7, 5), 'short': (300, 900), 'medium': (200, 5000), 'long': (40, 30000), 'ragged': (257, 4099)}[style]
    lens = {'tiny': rng.integers(0, 4, m), 'short': rng.integers(0, 20, m), 'medium': rng.integers(20, 150, m),
            'long': rng.integers(500, 3000, m),
            'ragged': np.array([0, 1, 33, 0, 200, 64, 0, 7, 512, 90, 0, 31, 63, 65, 128] * 18)[:m]}[style]
    w, idx, ptr = rand_csr(rng, m, k, lens, dtype=dtype, homo=homo)
    v = rng.normal(0, 1, m if transpose else k).astype(dtype)
    v[rng.random(v.size) < 0.2] = 0
    got = be.csrmv(w, idx, ptr, v, shape=(m, k), transpose=transpose)
    assert got.dtype == dtype and got.shape == ((k,) if transpose else (m,))
    close(got, oracle.csrmv(w.astype(np.float64), idx, ptr, v.astype(np.float64), (m, k), transpose), dtype)


@pytest.mark.parametrize('n', [1, 2, 3, 8, 17, 33, 70])
@pytest.mark.parametrize('homo', [False, True])
@pytest.mark.parametrize('transpose', [False, True])
def test_csrmm_against_the_oracle(be, oracle, n, homo, transpose):
    rng = np.random.default_rng(1000 + n)
    m, k = 150, 700
    lens = rng.integers(0, 60, m)
    lens[::11] = 0
    w, idx, ptr = rand_csr(rng, m, k, lens, homo=homo)
    B = rng.normal(0, 1, (m if transpose else k, n)).astype(np.float32)
    got = be.csrmm(w, idx, ptr, B, shape=(m, k), transpose=transpose)
    assert got.shape == ((k if transpose else m), n)
    close(got, oracle.csrmm(w.astype(np.float64), idx, ptr, B.astype(np.float64), (m, k), transpose), np.float32)


def test_bf16_int64_indptr_unaligned_views_and_device_tensors(be, oracle):
    rng = np.random.default_rng(5)
    m, k = 120, 640
    lens = rng.integers(0, 50, m)
    w, idx, ptr = rand_csr(rng, m, k, lens)
    v = rng.normal(0, 1, k).astype(np.float32)
    ref = oracle.csrmv(w.astype(np.float64), idx, ptr, v.astype(np.float64), (m, k), False)
    wd, idd, ptd, vd = (torch.from_numpy(a).cuda() for a in (w, idx, ptr.astype(np.int64), v))
    got = be.csrmv(wd, idd, ptd, vd, shape=(m, k))                      # device tensors in, device tensor out; int64 indptr
    assert isinstance(got, torch.Tensor) and got.dtype == torch.float32
    close(got.cpu().numpy(), ref, np.float32)
    # views that start off a 16-byte boundary (the kernels read aligned groups of four entries)
    pad_w = torch.cat([torch.zeros(1, device='cuda'), wd])[1:]
    pad_i = torch.cat([torch.zeros(1, dtype=torch.int32, device='cuda'), idd])[1:]
    assert pad_w.data_ptr() % 16 != 0 and pad_i.data_ptr() % 16 != 0
    close(be.csrmv(pad_w, pad_i, ptd, vd, shape=(m, k)).cpu().numpy(), ref, np.float32)
    # bf16 weights and operand, both directions
    wb, vb = wd.to(torch.bfloat16), vd.to(torch.bfloat16)
    refb = oracle.csrmv(wb.float().cpu().numpy().astype(np.float64), idx, ptr, vb.float().cpu().numpy().astype(np.float64), (m, k), False)
    gb = be.csrmv(wb, idd, ptd, vb, shape=(m, k))
    assert gb.dtype == torch.bfloat16
    np.testing.assert_allclose(gb.float().cpu().numpy(), refb, rtol=2e-2, atol=2e-2 * max(1.0, float(np.abs(refb).max())))
    um = torch.from_numpy(rng.normal(0, 1, m).astype(np.float32)).cuda().to(torch.bfloat16)
    reft = oracle.csrmv(wb.float().cpu().numpy().astype(np.float64), idx, ptr, um.float().cpu().numpy().astype(np.float64), (m, k), True)
    gt = be.csrmv(wb, idd, ptd, um, shape=(m, k), transpose=True)
    np.testing.assert_allclose(gt.float().cpu().numpy(), reft, rtol=2e-2, atol=2e-2 * max(1.0, float(np.abs(reft).max())))


def test_empty_shapes_and_validation(be):
    w = np.zeros(0, np.float32); idx = np.zeros(0, np.int32)
    out = be.csrmv(w, idx, np.zeros(4, np.int32), np.ones(5, np.float32), shape=(3, 5))
    np.testing.assert_array_equal(out, np.zeros(3, np.float32))                       # empty rows are written, not left as they were
    out = be.csrmv(w, idx, np.zeros(4, np.int32), np.ones(3, np.float32), shape=(3, 5), transpose=True)
    np.testing.assert_array_equal(out, np.zeros(5, np.float32))
    assert be.csrmm(w, idx, np.zeros(1, np.int32), np.zeros((5, 2), np.float32), shape=(0, 5)).shape == (0, 2)
    with pytest.raises(AssertionError, match='Shape mismatch'):
        be.csrmv(np.ones(1, np.float32), np.zeros(1, np.int32), np.array([0, 1], np.int32), np.ones(4, np.float32), shape=(1, 3))
    with pytest.raises(AssertionError, match='floating-point'):
        be.csrmv(np.ones(1, np.int32), np.zeros(1, np.int32), np.array([0, 1], np.int32), np.ones(3, np.float32), shape=(1, 3))
    with pytest.raises(AssertionError, match='2D'):
        be.csrmm(np.ones(1, np.float32), np.zeros(1, np.int32), np.array([0, 1], np.int32), np.ones(3, np.float32), shape=(1, 3))


@pytest.mark.parametrize('homo', [False, True])
@pytest.mark.parametrize('transpose', [False, True])
@pytest.mark.parametrize('n_conn', [1, 5, 40, 300])
def test_fcn_twins_against_the_oracle(be, oracle, homo, transpose, n_conn):
    rng = np.random.default_rng(77 + n_conn)
    rows, cols = 230, 900
    idx = rng.integers(0, cols, (rows, n_conn)).astype(np.int32)
    w = np.array([1.5], np.float32) if homo else rng.normal(0, 1, (rows, n_conn)).astype(np.float32)
    v = rng.normal(0, 1, rows if transpose else cols).astype(np.float32)
    close(be.fcnmv(w, idx, v, shape=(rows, cols), transpose=transpose),
          oracle.fcnmv(w.astype(np.float64), idx, v.astype(np.float64), (rows, cols), transpose), np.float32)
    M = rng.normal(0, 1, (rows if transpose else cols, 6)).astype(np.float32)
    close(be.fcnmm(w, idx, M, shape=(rows, cols), transpose=transpose),
          oracle.fcnmm(w.astype(np.float64), idx, M.astype(np.float64), (rows, cols), transpose), np.float32)


def test_containers_take_dense_operands(be):
    """``csr @ x``, ``x @ csr``, ``csc @ x``, ``x @ csc`` and the FixedNum classes with vectors and matrices == the products of
    the dense view of the same matrix (the recipe of the reference's operator tests, ``_csr/main_test.py:1289-1372``)."""
    rng = np.random.default_rng(9)
    m, k, n = 40, 55, 6
    lens = rng.integers(0, 12, m)
    w, idx, ptr = rand_csr(rng, m, k, lens)
    csr = be.CSR((w, idx, ptr), shape=(m, k))
    D = csr.todense().astype(np.float64)
    xk, xm = rng.normal(0, 1, k).astype(np.float32), rng.normal(0, 1, m).astype(np.float32)
    Xk, Xm = rng.normal(0, 1, (k, n)).astype(np.float32), rng.normal(0, 1, (n, m)).astype(np.float32)
    tol = dict(rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(csr @ xk, D @ xk, **tol)
    np.testing.assert_allclose(xm @ csr, xm @ D, **tol)
    np.testing.assert_allclose(csr @ Xk, D @ Xk, **tol)
    np.testing.assert_allclose(Xm @ csr, Xm @ D, **tol)
    csc = csr.T                                                # CSC of the transpose: shape (k, m)
    Dt = D.T
    np.testing.assert_allclose(csc @ xm, Dt @ xm, **tol)
    np.testing.assert_allclose(xk @ csc, xk @ Dt, **tol)
    np.testing.assert_allclose(csc @ Xm.T, Dt @ Xm.T, **tol)
    np.testing.assert_allclose(Xk.T @ csc, Xk.T @ Dt, **tol)
    for cls in (be.FixedNumPerPre, be.FixedNumPerPost):
        rows, cols, K = 30, 45, 7
        fi = rng.integers(0, cols if cls is be.FixedNumPerPre else rows, (rows if cls is be.FixedNumPerPre else cols, K)).astype(np.int32)
        fw = rng.normal(0, 1, fi.shape).astype(np.float32)
        conn = cls((fw, fi), shape=(rows, cols))
        Df = conn.todense().astype(np.float64)
        xc, xr = rng.normal(0, 1, cols).astype(np.float32), rng.normal(0, 1, rows).astype(np.float32)
        Xc, Xr = rng.normal(0, 1, (cols, n)).astype(np.float32), rng.normal(0, 1, (n, rows)).astype(np.float32)
        np.testing.assert_allclose(conn @ xc, Df @ xc, **tol)
        np.testing.assert_allclose(xr @ conn, xr @ Df, **tol)
        np.testing.assert_allclose(conn @ Xc, Df @ Xc, **tol)
        np.testing.assert_allclose(Xr @ conn, Xr @ Df, **tol)
        conn.build_mirror()                                    # with a mirror the scatter direction gathers over it: same numbers
        np.testing.assert_allclose(conn @ xc, Df @ xc, **tol)
        np.testing.assert_allclose(xr @ conn, xr @ Df, **tol)
        np.testing.assert_allclose(conn @ Xc, Df @ Xc, **tol)
        np.testing.assert_allclose(Xr @ conn, Xr @ Df, **tol)
    # with a mirror the scatter direction (x @ csr, csc @ x) runs as a gather over the mirror's arrays: same numbers
    csr_m = be.CSR((w, idx, ptr), shape=(m, k)).prepare(mirror=True)
    assert csr_m.buffers.get('mirror') is not None
    np.testing.assert_allclose(xm @ csr_m, xm @ D, **tol)
    np.testing.assert_allclose(Xm @ csr_m, Xm @ D, **tol)
    np.testing.assert_allclose(csr_m.T @ xm, Dt @ xm, **tol)
    # device tensors on either side (torch defers to the container's __rmatmul__ like numpy does)
    tk, tm = torch.from_numpy(xk).cuda(), torch.from_numpy(xm).cuda()
    r = csr @ tk
    assert isinstance(r, torch.Tensor)
    np.testing.assert_allclose(r.cpu().numpy(), D @ xk, **tol)
    np.testing.assert_allclose((tm @ csr).cpu().numpy(), xm @ D, **tol)
    # the event path is untouched: a BinaryArray still means "active or not", a plain array means its values
    s = rng.random(k) < 0.3
    np.testing.assert_allclose(csr @ be.BinaryArray(s), D @ s.astype(np.float64), **tol)
    np.testing.assert_allclose(csr @ s.astype(np.float32), D @ s.astype(np.float64), **tol)


# ---------------------------------------------------------------------------------------------------- JIT connectivity
@pytest.mark.parametrize('family', ['s', 'u', 'n'])
@pytest.mark.parametrize('transpose', [False, True])
@pytest.mark.parametrize('corder', [True, False])
def test_jit_float_twins_against_the_oracle(be, oracle, family, transpose, corder):
    """jit{s,u,n}mv / mm with a dense operand == the oracle's generator matrix (the reference's golden walk, pinned bit for bit by
    tests/test_oracle.py) times the operand; the vector and the matrix products see different draws (stride 32 / 4)."""
    rng = np.random.default_rng(31)
    shape, prob, seed = (150, 210), 0.08, 123
    params = {'s': (1.5,), 'u': (-0.5, 1.25), 'n': (0.3, 0.8)}[family]
    w0, w1 = (params[0], 0.0) if family == 's' else params
    in_len = shape[0] if transpose else shape[1]
    v = rng.normal(0, 1, in_len).astype(np.float32)
    v[rng.random(in_len) < 0.3] = 0
    B = rng.normal(0, 1, (in_len, 11)).astype(np.float32)
    fmv = {'s': be.jitsmv, 'u': be.jitumv, 'n': be.jitnmv}[family]
    fmm = {'s': be.jitsmm, 'u': be.jitumm, 'n': be.jitnmm}[family]
    wargs = tuple(np.float32(p) for p in params)
    got = fmv(*wargs, prob, v, seed, shape=shape, transpose=transpose, corder=corder)
    ref = oracle.jitmv(family, w0, w1, prob, v, seed, shape=shape, transpose=transpose, corder=corder)
    assert got.dtype == np.float32 and got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(ref).max())))
    gotm = fmm(*wargs, prob, B, seed, shape=shape, transpose=transpose, corder=corder)
    refm = oracle.jitmm(family, w0, w1, prob, B, seed, shape=shape, transpose=transpose, corder=corder)
    assert gotm.shape == refm.shape
    np.testing.assert_allclose(gotm, refm, rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(refm).max())))


def test_jit_float_scatter_keeps_small_addends_of_a_wide_operand(be, oracle, monkeypatch):
    """the float-operand scatter (corder=False) sums in LDS fixed point at ONE global exponent taken from the
    operand's largest magnitude; an operand spanning 1e-6 ... 1e6 left its small addends a few bits, and outputs fed only by small
    values came out far from the f32-atomic reference, silently.  The fixed-point sums are now taken only when the smallest
    non-zero |x| keeps `JIT_FLOAT_MIN_BITS` bits there — otherwise the float-atomic kernel.  Outputs fed by small values only are
    compared RELATIVELY here (the main sweep's atol is scaled by the largest output and cannot see them)."""
    import brainevent_amd._jitc as J
    rng = np.random.default_rng(33)
    shape, prob, seed = (4000, 1500), 0.004, 9
    in_len = shape[0]
    v = (10.0 ** rng.uniform(-6, -3, in_len)).astype(np.float32)          # small values on nearly every row ...
    v[rng.random(in_len) < 0.3] = 0
    v[[5, 1700, 3100]] = [1e6, -4e5, 7e5]                                  # ... and three rows that set the exponent
    for family, wargs in (('s', (np.float32(1.5),)), ('u', (np.float32(0.5), np.float32(1.25)))):
        f = {'s': be.jitsmv, 'u': be.jitumv}[family]
        w0, w1 = (float(wargs[0]), 0.0) if family == 's' else (float(wargs[0]), float(wargs[1]))
        got = f(*wargs, prob, v, seed, shape=shape, transpose=True, corder=False)
        ref = oracle.jitmv(family, w0, w1, prob, v.astype(np.float64), seed, shape=shape, transpose=True, corder=False)
        small = (np.abs(ref) > 0) & (np.abs(ref) < 1.0)
        assert small.sum() > 1000, 'the case must contain outputs made of small addends only'
        np.testing.assert_allclose(got[small], ref[small], rtol=2e-5)
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-5 * float(np.abs(ref).max()))
    # an operand of one scale still takes the fixed-point sums (bitwise repeatable, unlike float atomics)
    u = rng.uniform(0.5, 2.0, in_len).astype(np.float32)
    a = be.jitsmv(np.float32(1.5), prob, u, seed, shape=shape, transpose=True, corder=False)
    b = be.jitsmv(np.float32(1.5), prob, u, seed, shape=shape, transpose=True, corder=False)
    np.testing.assert_array_equal(a, b)


def test_jit_float_twins_agree_with_the_event_driven_products_on_0_1_operands(be):
    """On an operand of zeros and ones the float twin and the event-driven product are the same sum (the reference tests its
    binary ops against its float ops this way, ``_jit_scalar/binary_test.py:84-118``) — the scalar family exactly."""
    rng = np.random.default_rng(32)
    shape, prob, seed = (400, 300), 0.05, 7
    for transpose in (False, True):
        for corder in (True, False):
            s = rng.random(shape[0] if transpose else shape[1]) < 0.3
            a = be.binary_jitsmv(np.float32(2.0), prob, s, seed, shape=shape, transpose=transpose, corder=corder)
            b = be.jitsmv(np.float32(2.0), prob, s.astype(np.float32), seed, shape=shape, transpose=transpose, corder=corder)
            np.testing.assert_array_equal(a, b)
            S = rng.random((shape[0] if transpose else shape[1], 5)) < 0.3
            am = be.binary_jitsmm(np.float32(2.0), prob, S, seed, shape=shape, transpose=transpose, corder=corder)
            bm = be.jitsmm(np.float32(2.0), prob, S.astype(np.float32), seed, shape=shape, transpose=transpose, corder=corder)
            np.testing.assert_array_equal(am, bm)


def test_jit_containers_take_dense_operands(be):
    rng = np.random.default_rng(33)
    for cls, params in ((be.JITCScalarR, (np.float32(0.7),)), (be.JITCUniformC, (np.float32(-1.0), np.float32(1.0))),
                        (be.JITCNormalR, (np.float32(0.1), np.float32(0.5)))):
        for corder in (True, False):
            M = cls((*params, 0.1, 11), shape=(60, 90), corder=corder)
            Dv, Dm = M.mv.todense().astype(np.float64), M.mm.todense().astype(np.float64)
            x90, x60 = rng.normal(0, 1, 90).astype(np.float32), rng.normal(0, 1, 60).astype(np.float32)
            X90, X60 = rng.normal(0, 1, (90, 4)).astype(np.float32), rng.normal(0, 1, (4, 60)).astype(np.float32)
            tol = dict(rtol=1e-5, atol=1e-4)
            np.testing.assert_allclose(M @ x90, Dv @ x90, **tol)
            np.testing.assert_allclose(x60 @ M, x60 @ Dv, **tol)
            np.testing.assert_allclose(M @ X90, Dm @ X90, **tol)
            np.testing.assert_allclose(X60 @ M, X60 @ Dm, **tol)
    # half-precision weights and the prob = 0 convention (zeros, like the event-driven twins)
    h = be.jitsmv(np.float16(0.5), 0.1, rng.normal(0, 1, 90).astype(np.float16), 3, shape=(60, 90))
    assert h.dtype == np.float16 and h.shape == (60,)
    z = be.jitsmv(np.float32(0.5), 0.0, np.ones(90, np.float32), 3, shape=(60, 90))
    np.testing.assert_array_equal(z, np.zeros(60, np.float32))
