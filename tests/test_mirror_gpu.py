"""SURVEY.md 8 f1 — the event-driven gather direction: the column-block CSR -> CSC kernels (``be_csr_to_csc_*``), the
mirrors of ``CSR`` / ``CSC`` / ``FixedNumPerPre`` / ``FixedNumPerPost`` built from them, and the perm-fused indexed products.

Reference behaviour being matched (read as text): ``brainevent/_misc.py:1516`` (``csr_to_csc_index``; the
``gpu_column_block`` method leaves the order inside a column unspecified, ``_csr/csr_to_csc.cu:26-27``),
``_csr/main.py:1321-1357`` / ``:1647-1654`` / ``:2643-2650`` (``CSR @ ev`` / ``ev @ CSC`` through the cached CSC triple),
``_fcn/main.py:280-326`` (the same for fixed-number connectivity), ``_csr/binary_indexed.py:70`` / ``:615`` (slot ``j`` reads
``data[perm[j]]``).  The checker is the numpy oracle's gather (``oracle/oracle_np.py``) or scipy's conversion.
"""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-5, 1e-5


def rand_csr(rng, m, k, row_lens, dtype=np.float32, homo=False, ptr_dtype=np.int32):
    row_lens = np.asarray(row_lens, dtype=np.int64)
    indptr = np.concatenate([[0], np.cumsum(row_lens)]).astype(ptr_dtype)
    nnz = int(indptr[-1])
    indices = rng.integers(0, k, nnz).astype(np.int32)
    w = np.asarray([1.5], dtype=dtype) if homo else rng.uniform(0.1, 1.0, nnz).astype(dtype)
    return w, indices, indptr


def _columns_as_sets(ptr, rows, extra=None):
    """Per column: the sorted list of (row[, extra]) — the order inside a column is unspecified."""
    out = []
    for c in range(len(ptr) - 1):
        seg = slice(int(ptr[c]), int(ptr[c + 1]))
        out.append(sorted(zip(rows[seg].tolist(), extra[seg].tolist())) if extra is not None else sorted(rows[seg].tolist()))
    return out


@pytest.mark.parametrize('ptr_dtype', [np.int32, np.int64])
@pytest.mark.parametrize('lens', ['ragged', 'short', 'long'])
def test_column_block_conversion_equals_scipy(be, ptr_dtype, lens):
    rng = np.random.default_rng(11)
    m, k = 700, 531
    row_lens = {'ragged': rng.integers(0, 200, m), 'short': rng.integers(0, 4, m), 'long': rng.integers(900, 1100, m)}[lens]
    w, idx, ptr = rand_csr(rng, m, k, row_lens, ptr_dtype=ptr_dtype)
    cptr, crows, perm = be.csr_to_csc_index(ptr, idx, shape=(m, k), method='gpu_column_block')
    assert cptr.dtype == np.int32 and crows.dtype == np.int32 and perm.dtype == np.int32
    ref = sp.csr_matrix((np.arange(1, idx.size + 1, dtype=np.float64), idx, ptr), shape=(m, k))   # duplicates kept apart below
    counts = np.bincount(idx, minlength=k)
    np.testing.assert_array_equal(cptr, np.concatenate([[0], np.cumsum(counts)]))
    # perm is a permutation, and slot j really is entry perm[j]: its column is the slot's column, its row the stored row
    assert np.array_equal(np.sort(perm), np.arange(idx.size))
    col_of_slot = np.repeat(np.arange(k), np.diff(cptr))
    row_of_entry = np.repeat(np.arange(m), np.diff(ptr))
    np.testing.assert_array_equal(idx[perm], col_of_slot)
    np.testing.assert_array_equal(row_of_entry[perm], crows)
    # same structure as the stable (sort-based) method up to the order inside a column
    sptr, srows, sperm = be.csr_to_csc_index(ptr, idx, shape=(m, k), method='coo')
    np.testing.assert_array_equal(sptr, cptr)
    assert _columns_as_sets(cptr, crows, perm) == _columns_as_sets(sptr, srows, sperm)
    del ref
    # without the permutation
    _, r2, p2 = be.csr_to_csc_index(ptr, idx, shape=(m, k), method='gpu_column_block', include_perm=False)
    assert p2 is None and _columns_as_sets(cptr, r2) == _columns_as_sets(cptr, crows)


def test_column_blocks_tile_the_whole_conversion(be):
    """Any partition of the columns gives the blocks of the same CSC arrays; weights move along (2 / 4 / 8-byte elements)."""
    from brainevent_amd._convert import CscBuilder
    rng = np.random.default_rng(12)
    m, k = 400, 1000
    for dtype in (np.float32, np.float64, np.float16):
        w, idx, ptr = rand_csr(rng, m, k, rng.integers(0, 120, m), dtype=dtype)
        b = CscBuilder(torch.tensor(ptr).cuda(), torch.tensor(idx).cuda(), shape=(m, k))
        wd = torch.tensor(w).cuda()
        dense = np.zeros((m, k)); np.add.at(dense, (np.repeat(np.arange(m), np.diff(ptr)), idx), w.astype(np.float64))
        got = np.zeros((m, k))
        edges = [0, 1, 17, 400, 401, 999, 1000]
        for c0, c1 in zip(edges[:-1], edges[1:]):
            rows, wb, perm = b.block(c0, c1, data=wd, perm=True)
            bptr = b.block_indptr(c0, c1).cpu().numpy()
            assert bptr[0] == 0 and bptr[-1] == rows.numel() == wb.numel() == perm.numel()
            cols = np.repeat(np.arange(c0, c1), np.diff(bptr))
            np.add.at(got, (rows.cpu().numpy(), cols), wb.double().cpu().numpy())
            np.testing.assert_array_equal(wb.cpu().numpy(), w[perm.cpu().numpy()])
        np.testing.assert_allclose(got, dense, rtol=1e-14, atol=0)     # (duplicates of one cell are summed in slot order)
    assert b.max_col_count == int(np.bincount(idx, minlength=k).max())


def test_fixed_number_rows_convert_through_the_implicit_indptr(be):
    from brainevent_amd._convert import CscBuilder
    rng = np.random.default_rng(13)
    n_pre, n_post, K = 300, 450, 37
    idx = rng.integers(0, n_post, (n_pre, K)).astype(np.int32)
    b = CscBuilder(None, torch.tensor(idx).cuda(), shape=(n_pre, n_post), row_len=K)
    rows, _, perm = b.block(0, n_post, perm=True)
    sptr, srows, sperm = be.fixed_conn_num_csc_structure(idx, shape=(n_pre, n_post))
    np.testing.assert_array_equal(b.offsets().cpu().numpy(), sptr)
    assert _columns_as_sets(sptr, rows.cpu().numpy(), perm.cpu().numpy()) == _columns_as_sets(sptr, srows, sperm)
    with pytest.raises(ValueError):        # a column id outside the shape is reported, not dropped silently
        CscBuilder(None, torch.tensor(idx).cuda(), shape=(n_pre, 10), row_len=K)


@pytest.mark.parametrize('homo', [True, False])
@pytest.mark.parametrize('route', ['plan', 'plan_released', 'plan_blocked', 'binned', 'direct'])
def test_mirror_routes_equal_the_gather(be, oracle, monkeypatch, route, homo):
    """Every way a mirror can be held — planned with / without its raw arrays, planned from column blocks that are resident
    one at a time, binned, direct — gives the gather product; a weight update in place is followed on the next call."""
    import brainevent_amd._csr as C
    monkeypatch.setattr(C, 'PLAN_MIN_NNZ', 1000)
    monkeypatch.setattr(C, 'AUTO_MIRROR_MIN_NNZ', None)
    rng = np.random.default_rng(21)
    if route == 'binned':
        # the transpose has 2000 rows of ~50 entries over 200000 outputs: fewer than 8 per (row, slice) -> binned
        m, k = 200_000, 2000
        w, idx, ptr = rand_csr(rng, m, k, rng.integers(0, 2, m), homo=homo)
    else:
        m, k = 2500, 3000
        w, idx, ptr = rand_csr(rng, m, k, rng.integers(100, 300, m), homo=homo)
    if route == 'direct':
        monkeypatch.setattr(C, 'PLAN_MIN_NNZ', 10 ** 9)
    if route == 'plan_blocked':
        real = C._free_device_bytes
        monkeypatch.setattr(C, '_free_device_bytes', lambda: 1 << 20)      # nothing fits whole: column blocks
    data = torch.tensor(w, device='cuda')
    csr = be.CSR((data, torch.tensor(idx, device='cuda'), torch.tensor(ptr, device='cuda')), shape=(m, k))
    mr = csr.build_mirror(keep_raw={'plan': True, 'plan_released': False}.get(route))
    if route == 'plan_blocked':
        monkeypatch.setattr(C, '_free_device_bytes', real)
    if route.startswith('plan'):
        assert isinstance(mr.plan, C.ScatterPlan) and mr.released == (route != 'plan')
    elif route == 'binned':
        assert isinstance(mr.plan, C.BinnedScatter) and not mr.released
    else:
        assert mr.plan is None and not mr.released
    v = rng.random(k) < 0.05
    ev = be.BinaryArray(torch.tensor(v, device='cuda'))
    ref = oracle.binary_csrmv(w.astype(np.float64), idx, ptr, v, (m, k), False)
    got = (csr @ ev).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=RTOL, atol=ATOL)
    if homo:
        np.testing.assert_array_equal(got, ref.astype(np.float32))          # counts x one weight: exact
    B = rng.random((k, 5)) < 0.05
    gotB = (csr @ be.BinaryArray(torch.tensor(B, device='cuda'))).cpu().numpy()
    np.testing.assert_allclose(gotB, oracle.binary_csrmm(w.astype(np.float64), idx, ptr, B, (m, k), False), rtol=RTOL, atol=ATOL)
    csc = be.CSC((data, torch.tensor(idx, device='cuda'), torch.tensor(ptr, device='cuda')), shape=(k, m))
    csc.build_mirror(keep_raw={'plan': True, 'plan_released': False}.get(route))
    np.testing.assert_allclose((ev @ csc).cpu().numpy(), ref, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose((be.BinaryArray(torch.tensor(B.T.copy(), device='cuda')) @ csc).cpu().numpy(), gotB.T, rtol=RTOL, atol=ATOL)
    if not homo:
        data.mul_(-0.5).add_(2.0)
        ref2 = oracle.binary_csrmv((w * np.float32(-0.5) + np.float32(2.0)).astype(np.float64), idx, ptr, v, (m, k), False)
        np.testing.assert_allclose((csr @ ev).cpu().numpy(), ref2, rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose((ev @ csc).cpu().numpy(), ref2, rtol=RTOL, atol=ATOL)


def test_mirror_with_perm_refreshes_by_gather_copy(be, oracle, monkeypatch):
    import brainevent_amd._csr as C
    monkeypatch.setattr(C, 'PLAN_MIN_NNZ', 1000)
    rng = np.random.default_rng(22)
    m, k = 900, 700
    w, idx, ptr = rand_csr(rng, m, k, rng.integers(150, 250, m))
    data = torch.tensor(w, device='cuda')
    csr = be.CSR((data, torch.tensor(idx, device='cuda'), torch.tensor(ptr, device='cuda')), shape=(m, k)).prepare(mirror=True)
    mr = csr.buffers['mirror']
    assert mr.perm is not None and torch.equal(mr.data, data[mr.perm.long()])
    buf = mr.data.data_ptr()
    data.add_(1.0)
    v = rng.random(k) < 0.1
    got = (csr @ be.BinaryArray(torch.tensor(v, device='cuda'))).cpu().numpy()
    assert csr.buffers['mirror'] is mr and mr.data.data_ptr() == buf          # same object, same buffer: a captured graph stays valid
    np.testing.assert_allclose(got, oracle.binary_csrmv((w + np.float32(1)).astype(np.float64), idx, ptr, v, (m, k), False),
                               rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize('homo', [True, False])
def test_fixed_number_mirrors(be, oracle, monkeypatch, homo):
    """`FixedNumPerPre @ spk` / `spk @ FixedNumPerPost` (vectors and matrices) through the mirror == the gather kernel."""
    import brainevent_amd._csr as C
    monkeypatch.setattr(C, 'PLAN_MIN_NNZ', 1000)
    monkeypatch.setattr(C, 'AUTO_MIRROR_MIN_NNZ', None)
    rng = np.random.default_rng(23)
    n_pre, n_post, K = 1500, 2200, 120
    idx = rng.integers(0, n_post, (n_pre, K)).astype(np.int32)
    w = np.array([0.75], np.float32) if homo else rng.uniform(0.1, 1, (n_pre, K)).astype(np.float32)
    s = rng.random(n_post) < 0.08
    S = rng.random((n_post, 3)) < 0.08
    for cls, shape, ev_v, ev_m in ((be.FixedNumPerPre, (n_pre, n_post), lambda M: M @ be.BinaryArray(s), lambda M: M @ be.BinaryArray(S)),
                                   (be.FixedNumPerPost, (n_post, n_pre), lambda M: be.BinaryArray(s) @ M,
                                    lambda M: (be.BinaryArray(S.T.copy()) @ M).T)):
        plain = cls((w, idx), shape=shape)
        want_v, want_m = ev_v(plain), ev_m(plain)            # gather kernel (no mirror: automatic build disabled above)
        assert plain.buffers.get('mirror') is None
        mirrored = cls((w, idx), shape=shape).prepare(mirror=True)
        assert isinstance(mirrored.buffers['mirror'], C.Mirror) and mirrored.buffers['mirror'].shape == (n_post, n_pre)
        np.testing.assert_allclose(ev_v(mirrored), want_v, rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(ev_m(mirrored), want_m, rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(want_v, oracle.binary_fcnmv(np.asarray(w, np.float64), idx, s, (n_pre, n_post), False), rtol=RTOL, atol=ATOL)
        with pytest.raises(AssertionError):          # the shape contract is checked in front of the mirror too
            mirrored._binary_matvec(np.zeros(7, bool), cls is be.FixedNumPerPost)
    # in-place weight update
    if not homo:
        wd = torch.tensor(w, device='cuda')
        M = be.FixedNumPerPre((wd, torch.tensor(idx, device='cuda')), shape=(n_pre, n_post)).prepare(mirror=True)
        sv = be.BinaryArray(torch.tensor(s, device='cuda'))
        a = (M @ sv).cpu().numpy()
        wd.mul_(3.0)
        b = (M @ sv).cpu().numpy()
        np.testing.assert_allclose(b, 3.0 * a, rtol=1e-5, atol=1e-5)


def test_mirror_is_built_on_first_use_when_it_pays(be, oracle, monkeypatch):
    """The reference builds its CSC triple on the first `CSR @ events` (``_csr/main.py:1321-1357``); here a matrix above
    ``AUTO_MIRROR_MIN_NNZ`` entries does, unless the mirror would not fit beside it (then: the gather kernel + a warning)."""
    import brainevent_amd._csr as C
    monkeypatch.setattr(C, 'PLAN_MIN_NNZ', 1000)
    rng = np.random.default_rng(24)
    m, k = 800, 900
    w, idx, ptr = rand_csr(rng, m, k, rng.integers(50, 100, m))
    v = rng.random(k) < 0.1
    ref = oracle.binary_csrmv(w.astype(np.float64), idx, ptr, v, (m, k), False)
    small = be.CSR((w, idx, ptr), shape=(m, k))
    np.testing.assert_allclose(small @ be.BinaryArray(v), ref, rtol=RTOL, atol=ATOL)
    assert small.buffers.get('mirror') is None                       # below the default threshold: the gather kernel
    monkeypatch.setattr(C, 'AUTO_MIRROR_MIN_NNZ', 1000)
    auto = be.CSR((w, idx, ptr), shape=(m, k))
    np.testing.assert_allclose(auto @ be.BinaryArray(v), ref, rtol=RTOL, atol=ATOL)
    assert isinstance(auto.buffers['mirror'], C.Mirror)
    fcn = be.FixedNumPerPre((w[:m * 50].reshape(m, 50), idx[:m * 50].reshape(m, 50)), shape=(m, k))
    fcn @ be.BinaryArray(v)
    assert isinstance(fcn.buffers['mirror'], C.Mirror)
    monkeypatch.setattr(C, '_free_device_bytes', lambda: 1 << 16)
    tight = be.CSR((w, idx, ptr), shape=(m, k))
    with pytest.warns(UserWarning, match='does not fit'):
        np.testing.assert_allclose(tight @ be.BinaryArray(v), ref, rtol=RTOL, atol=ATOL)
    assert tight.buffers['mirror'] is None
    np.testing.assert_allclose(tight @ be.BinaryArray(v), ref, rtol=RTOL, atol=ATOL)      # refused once, not asked again


@pytest.mark.parametrize('dtype', [np.float32, np.float64, np.float16])
@pytest.mark.parametrize('kind', ['bool', 'float'])
def test_perm_fused_direct_products(be, oracle, dtype, kind):
    """workspace=None: slot j reads data[perm[j]] inside the kernel (``be_binary_csrmm_{t,nt}_indexed``), both directions,
    vectors and batches, int32 and int64 permutations."""
    rng = np.random.default_rng(31)
    m, k = 260, 190
    w, idx, ptr = rand_csr(rng, m, k, rng.integers(0, 70, m), dtype=dtype)
    cptr, crows, perm = be.csr_to_csc_index(ptr, idx, shape=(m, k), method='gpu_column_block')
    tol = {np.float32: 1e-5, np.float64: 1e-10, np.float16: 2e-3}[dtype]
    sk, sm = rng.random(k) < 0.3, rng.random(m) < 0.3
    if kind == 'float':
        sk, sm = np.where(sk, 1.5, -1.0).astype(np.float32), np.where(sm, 0.5, 0.0).astype(np.float32)
    w64 = w.astype(np.float64)
    for p in (perm, perm.astype(np.int64)):
        got = be.binary_csrmv_indexed(w, crows, cptr, p, sk, shape=(k, m), transpose=True)
        assert got.dtype == dtype
        np.testing.assert_allclose(got.astype(np.float64), oracle.binary_csrmv(w64, idx, ptr, sk, (m, k), False), rtol=tol, atol=tol)
        got = be.binary_csrmv_indexed(w, crows, cptr, p, sm, shape=(k, m), transpose=False)
        np.testing.assert_allclose(got.astype(np.float64), oracle.binary_csrmv(w64, idx, ptr, sm, (m, k), True), rtol=tol, atol=tol)
    Bk, Bm = rng.random((k, 6)) < 0.3, rng.random((m, 6)) < 0.3
    got = be.binary_csrmm_indexed(w, crows, cptr, perm, Bk, shape=(k, m), transpose=True)
    np.testing.assert_allclose(got.astype(np.float64), oracle.binary_csrmm(w64, idx, ptr, Bk, (m, k), False), rtol=tol, atol=tol)
    got = be.binary_csrmm_indexed(w, crows, cptr, perm, Bm, shape=(k, m), transpose=False)
    np.testing.assert_allclose(got.astype(np.float64), oracle.binary_csrmm(w64, idx, ptr, Bm, (m, k), True), rtol=tol, atol=tol)


@pytest.mark.parametrize('route', ['plan', 'binned'])
def test_indexed_workspace_is_keyed_on_data_and_perm(be, oracle, monkeypatch, route):
    """A workspace from ``indexed_workspace`` holds the permuted weights (embedded / cached): the product performs no
    per-call gather, and follows an in-place update of the canonical weights."""
    import brainevent_amd._csr as C
    import brainevent_amd._convert as V
    monkeypatch.setattr(C, 'PLAN_MIN_NNZ', 1000)
    rng = np.random.default_rng(32)
    if route == 'plan':
        m, k = 1200, 1500
        w, idx, ptr = rand_csr(rng, m, k, rng.integers(100, 200, m))
    else:
        m, k = 200_000, 2000
        w, idx, ptr = rand_csr(rng, m, k, rng.integers(0, 2, m))
    dw, di, dp = torch.tensor(w, device='cuda'), torch.tensor(idx, device='cuda'), torch.tensor(ptr, device='cuda')
    cptr, crows, perm = be.csr_to_csc_index(dp, di, shape=(m, k), method='gpu_column_block')
    ws = be.indexed_workspace(dw, crows, cptr, perm, shape=(k, m))
    assert isinstance(ws, C.ScatterPlan if route == 'plan' else C.BinnedScatter)
    calls = []
    real = V.gather_by_perm
    monkeypatch.setattr(V, 'gather_by_perm', lambda *a, **kw: (calls.append(1), real(*a, **kw))[1])
    v = rng.random(k) < 0.05
    dv = torch.tensor(v, device='cuda')
    for _ in range(3):
        got = be.binary_csrmv_indexed(dw, crows, cptr, perm, dv, shape=(k, m), transpose=True, workspace=ws)
    assert calls == []                                               # three steps, no gather pass
    np.testing.assert_allclose(got.cpu().numpy(), oracle.binary_csrmv(w.astype(np.float64), idx, ptr, v, (m, k), False), rtol=RTOL, atol=ATOL)
    dw.mul_(2.0)
    got2 = be.binary_csrmv_indexed(dw, crows, cptr, perm, dv, shape=(k, m), transpose=True, workspace=ws)
    got3 = be.binary_csrmv_indexed(dw, crows, cptr, perm, dv, shape=(k, m), transpose=True, workspace=ws)
    assert calls == [1]                                              # one refresh for the update, none after
    np.testing.assert_allclose(got2.cpu().numpy(), 2.0 * got.cpu().numpy(), rtol=1e-5, atol=1e-5)
    assert torch.equal(got2, got3)


def test_compacted_and_packed_events_through_the_mirrors(be, oracle, monkeypatch):
    """The mirror turns the gather product into a scatter, so the event encodings the scatter kernels consume natively go straight
    in: a CompactBinary built on the device hands over its id list (no compaction launch), a BitPackedBinary its words — same bits
    as the plain BinaryArray operand, for CSR, CSC and the fixed-number containers."""
    import brainevent_amd._csr as C
    monkeypatch.setattr(C, 'PLAN_MIN_NNZ', 1000)
    rng = np.random.default_rng(61)
    m, k = 1500, 1800
    w, idx, ptr = rand_csr(rng, m, k, rng.integers(80, 160, m))
    v = rng.random(k) < 0.07
    dv = torch.tensor(v, device='cuda')
    plain, compact = be.BinaryArray(dv), be.CompactBinary.from_array(dv)
    packed = be.BitPackedBinary.from_packed(be.bitpack(dv, 0).reshape(-1), k)
    csr = be.CSR((torch.tensor(w, device='cuda'), torch.tensor(idx, device='cuda'), torch.tensor(ptr, device='cuda')), shape=(m, k)).prepare(mirror=True)
    csc = be.CSC((torch.tensor(w, device='cuda'), torch.tensor(idx, device='cuda'), torch.tensor(ptr, device='cuda')), shape=(k, m)).prepare(mirror=True)
    want = csr @ plain
    np.testing.assert_allclose(want.cpu().numpy(), oracle.binary_csrmv(w.astype(np.float64), idx, ptr, v, (m, k), False), rtol=RTOL, atol=ATOL)
    for ev in (compact, packed):
        assert torch.equal(csr @ ev, want) and torch.equal(ev @ csc, want)
    assert packed._value is None
    K = 64
    fidx = torch.tensor(rng.integers(0, k, (m, K)).astype(np.int32), device='cuda')
    fw = torch.tensor(rng.random((m, K)).astype(np.float32), device='cuda')
    fcn = be.FixedNumPerPre((fw, fidx), shape=(m, k)).prepare(mirror=True)
    fwant = fcn @ plain
    for ev in (compact, packed):
        assert torch.equal(fcn @ ev, fwant)


@pytest.mark.parametrize('scale', [1e6, 1e-6])
def test_binned_statistics_follow_weights_rewritten_through_a_raw_pointer(be, oracle, monkeypatch, scale):
    """a binned mirror / indexed workspace that takes its column statistics from binned steps
    (`STATS_BY_STEPS_MIN_NNZ`, the large-matrix path; lowered here) cached them under the weight tensor's stamp — but
    `Mirror.refreshed` and `_fresh_indexed_workspace` rewrite that tensor through a raw pointer (`gather_by_perm(out=...)`), which
    torch's version counter does not see: the refresh kept the old exponent whatever the new weights were.  Weights grown by 1e6
    wrapped the int64 sums; shrunk by 1e-6 they lost the accuracy gate.  `refresh_weights` now voids the cache."""
    import brainevent_amd._csr as C
    monkeypatch.setattr(C, 'PLAN_MIN_NNZ', 1000)
    monkeypatch.setattr(C.BinnedScatter, 'STATS_BY_STEPS_MIN_NNZ', 1000)
    rng = np.random.default_rng(31)
    m, k = 20000, 400_000
    lens = rng.integers(20, 40, m)
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    idx = rng.integers(0, k, ptr[-1]).astype(np.int32)
    w = rng.uniform(0.1, 1.0, ptr[-1]).astype(np.float32)
    data = torch.tensor(w, device='cuda')
    csr = be.CSR((data, torch.tensor(idx, device='cuda'), torch.tensor(ptr, device='cuda')), shape=(m, k)).prepare(mirror=True)
    mr = csr.buffers['mirror']
    assert isinstance(mr.plan, C.BinnedScatter) and mr.perm is not None and mr.plan._rows is not None
    v = rng.random(k) < 0.05
    ev = be.BinaryArray(torch.tensor(v, device='cuda'))
    np.testing.assert_allclose((csr @ ev).cpu().numpy(), oracle.binary_csrmv(w.astype(np.float64), idx, ptr, v, (m, k), False),
                               rtol=RTOL, atol=ATOL)
    e0 = mr.plan.scale_exp
    data.mul_(scale)                                             # in place: the mirror follows by a gather-copy into its own buffer
    got = (csr @ ev).cpu().numpy()
    assert csr.buffers['mirror'] is mr
    ref = oracle.binary_csrmv((w * np.float32(scale)).astype(np.float64), idx, ptr, v, (m, k), False)
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5 * float(np.abs(ref).max()))
    assert (mr.plan.scale_exp < e0) if scale > 1 else (mr.plan.scale_exp >= e0)       # the exponent moved with the weights
    # the perm-fused indexed product over a binned workspace takes the same refresh path
    t_idx, t_ptr, perm = mr.indices, mr.indptr, mr.perm
    ws = C.indexed_workspace(data, t_idx, t_ptr, perm, shape=(k, m), route='binned')
    a = be.binary_csrmv_indexed(data, t_idx, t_ptr, perm, ev.value, shape=(k, m), workspace=ws, transpose=True)
    np.testing.assert_allclose(a.cpu().numpy(), ref, rtol=1e-5, atol=1e-5 * float(np.abs(ref).max()))
    data.mul_(1.0 / scale)
    b = be.binary_csrmv_indexed(data, t_idx, t_ptr, perm, ev.value, shape=(k, m), workspace=ws, transpose=True)
    ref0 = oracle.binary_csrmv((w * np.float32(scale) * np.float32(1.0 / scale)).astype(np.float64), idx, ptr, v, (m, k), False)
    np.testing.assert_allclose(b.cpu().numpy(), ref0, rtol=1e-5, atol=1e-5 * float(np.abs(ref0).max()))


@pytest.mark.parametrize('homo', [True, False])
def test_an_automatic_mirror_is_cross_checked_once_against_the_gather_kernel(be, oracle, monkeypatch, homo):
    """a mirror the container builds by itself must not be able to change a result silently.  Its first plain
    event vector is also evaluated by the streaming gather kernel; agreement keeps the mirror (and the check is not repeated), a
    disagreement — provoked here by corrupting the mirror's weights / structure behind its back — returns the gather kernel's result,
    drops the mirror for good and warns.  A mirror the caller asked for (`build_mirror`, `prepare(mirror=True)`) is not checked."""
    import warnings
    import brainevent_amd._csr as C
    monkeypatch.setattr(C, 'PLAN_MIN_NNZ', 1000)
    monkeypatch.setattr(C, 'AUTO_MIRROR_MIN_NNZ', 1000)
    rng = np.random.default_rng(41)
    m, k = 1200, 1500
    w, idx, ptr = rand_csr(rng, m, k, rng.integers(60, 120, m))
    if homo:
        w = np.array([0.5], np.float32)
    v = torch.tensor(rng.random(k) < 0.1, device='cuda')
    ref = oracle.binary_csrmv(np.asarray(w, np.float64), idx, ptr, v.cpu().numpy(), (m, k), False)

    def container():
        return be.CSR((torch.tensor(w, device='cuda'), torch.tensor(idx, device='cuda'), torch.tensor(ptr, device='cuda')), shape=(m, k))
    # healthy: checked once, kept
    good = container()
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        out = good @ be.BinaryArray(v)
        mr = good.buffers['mirror']
        assert isinstance(mr, C.Mirror) and mr.check is None           # consumed by the first product
        out2 = good @ be.BinaryArray(v)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=RTOL, atol=ATOL)
    assert torch.equal(out, out2)
    # corrupted behind its back: the first product notices, answers with the gather kernel and drops the mirror
    broken = container()
    mr = broken._fresh_mirror(auto=True)
    assert mr.check is not None
    if homo:
        mr.indices[: mr.indices.numel() // 3] = 0                       # a third of the mirror's entries delivered to output 0
    else:
        mr.data.mul_(3.0)                                               # the round-4 signature: an addend delivered three times
    if mr.plan is not None and hasattr(mr.plan, 'refresh_weights') and not homo:
        mr.plan.refresh_weights(mr.data, mr.indices, mr.indptr)
    elif mr.plan is not None and homo:
        mr.plan = None                                                  # (the plan embeds the structure: serve from the raw arrays)
    with pytest.warns(RuntimeWarning, match='disagreed with the gather kernel'):
        got = broken @ be.BinaryArray(v)
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=RTOL, atol=ATOL)
    assert broken.buffers['mirror'] is None
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        np.testing.assert_allclose((broken @ be.BinaryArray(v)).cpu().numpy(), ref, rtol=RTOL, atol=ATOL)      # gather kernel, no rebuild
    # a mirror the caller asked for carries no check
    asked = container().prepare(mirror=True)
    assert asked.buffers['mirror'].check is None
