"""GPU parity tests for the CSR hot path: HIP kernels (through the C ABI) vs the numpy oracle.

Tolerance: fp32 accumulated currents within rtol=1e-5 (atol 1e-5 absorbs cancellation), the
north-star bound; homogeneous integer-count cases are compared exactly.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-5, 1e-5


def rand_csr(rng, m, k, row_lens, dtype=np.float32, homo=False):
    row_lens = np.asarray(row_lens, dtype=np.int64)
    indptr = np.concatenate([[0], np.cumsum(row_lens)]).astype(np.int32)
    nnz = int(indptr[-1])
    indices = rng.integers(0, k, nnz).astype(np.int32)
    w = np.asarray([1.5], dtype=dtype) if homo else rng.uniform(0.1, 1.0, nnz).astype(dtype)
    return w, indices, indptr


def spikes_of(rng, n, p, kind):
    s = rng.random(n) < p
    if kind == 'bool':
        return s
    if kind == 'u8':
        return s.astype(np.uint8)
    return np.where(s, rng.uniform(0.5, 2.0, n), -rng.uniform(0.0, 1.0, n)).astype(np.float32)


TOL = {np.float32: (1e-5, 1e-5), np.float64: (1e-10, 1e-10), np.float16: (2e-3, 2e-3)}


@pytest.mark.parametrize('transpose', [True, False])
@pytest.mark.parametrize('homo', [True, False])
@pytest.mark.parametrize('kind', ['bool', 'u8', 'float'])
@pytest.mark.parametrize('dtype', [np.float32, np.float64, np.float16])
def test_csrmv_direct_random(be, oracle, transpose, homo, kind, dtype):
    rng = np.random.default_rng(7)
    m, k = 300, 257
    w, idx, ptr = rand_csr(rng, m, k, rng.integers(0, 40, m), dtype=dtype, homo=homo)
    v = spikes_of(rng, m if transpose else k, 0.3, kind)
    got = be.binary_csrmv(w, idx, ptr, v, shape=(m, k), transpose=transpose)
    # oracle accumulates in f64 for the f16 case so the comparison isolates the kernel's f32 accumulate
    ref = oracle.binary_csrmv(w.astype(np.float64), idx, ptr, v, (m, k), transpose)
    rtol, atol = TOL[dtype]
    assert got.dtype == dtype and got.shape == ((k,) if transpose else (m,))
    np.testing.assert_allclose(got.astype(np.float64), ref, rtol=rtol, atol=atol)


def test_csrmv_bf16_torch(be, oracle):
    rng = np.random.default_rng(3)
    m, k = 128, 200
    w, idx, ptr = rand_csr(rng, m, k, rng.integers(1, 30, m))
    v = spikes_of(rng, m, 0.5, 'bool')
    wt = torch.tensor(w, device='cuda').to(torch.bfloat16)
    got = be.binary_csrmv(wt, torch.tensor(idx, device='cuda'), torch.tensor(ptr, device='cuda'),
                          torch.tensor(v, device='cuda'), shape=(m, k), transpose=True)
    assert isinstance(got, torch.Tensor) and got.dtype == torch.bfloat16
    ref = oracle.binary_csrmv(wt.float().cpu().numpy().astype(np.float64), idx, ptr, v, (m, k), True)
    np.testing.assert_allclose(got.float().cpu().numpy(), ref, rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize('transpose', [True, False])
def test_csrmv_ragged_and_empty_rows(be, oracle, transpose):
    # row-length sweep of the reference's GPU regression test (brainevent/_csr/binary_test.py:1442-1476)
    lens = [0, 1, 33, 0, 200, 64, 0, 7, 512, 90, 0, 31, 63, 65, 128]
    rng = np.random.default_rng(11)
    m, k = len(lens), 300
    w, idx, ptr = rand_csr(rng, m, k, lens, dtype=np.float64)
    v = np.ones(m if transpose else k, dtype=bool)
    got = be.binary_csrmv(w, idx, ptr, v, shape=(m, k), transpose=transpose)
    ref = oracle.binary_csrmv(w, idx, ptr, v, (m, k), transpose)
    np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-10)
    if not transpose:   # empty rows are written, not left uninitialised
        assert got[0] == 0 and got[3] == 0 and got[6] == 0 and got[10] == 0


@pytest.mark.parametrize('nnz_per_row', [4, 15, 16, 64, 511, 512, 1024])
def test_csrmv_gather_tiers(be, oracle, nnz_per_row):
    # tier-straddling sweep (brainevent/_csr/binary_test.py:1405-1437)
    rng = np.random.default_rng(nnz_per_row)
    m, k = 96, 2048
    w, idx, ptr = rand_csr(rng, m, k, [nnz_per_row] * m, dtype=np.float64)
    v = spikes_of(rng, k, 0.4, 'bool')
    got = be.binary_csrmv(w, idx, ptr, v, shape=(m, k), transpose=False)
    np.testing.assert_allclose(got, oracle.binary_csrmv(w, idx, ptr, v, (m, k), False), rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize('homo', [True, False])
def test_csrmv_gather_array_end_and_tiny_matrices(be, oracle, homo):
    # the lanes-per-row gather reads 16-byte pieces: the last rows end at every alignment against the arrays' end, matrices
    # hold fewer entries than one piece, and what lies behind the arrays in memory is poison (an active column with a huge
    # weight) that must never reach a result
    rng = np.random.default_rng(77)
    k = 300
    dev = torch.device('cuda', 0)
    v = spikes_of(rng, k, 0.5, 'bool')
    v[k - 1] = True
    cases = [[n] for n in range(0, 10)] + [[0, 1, 0], [1, 1, 1], [2, 0, 1], [3], [5, 0, 0, 2]]
    cases += [list(rng.integers(0, 40, 70)) + [t] for t in range(1, 9)]
    for lens in cases:
        m = len(lens)
        w, idx, ptr = rand_csr(rng, m, k, lens, homo=homo)
        nnz = int(ptr[-1])
        idx_buf = torch.full((nnz + 64,), k - 1, dtype=torch.int32, device=dev)
        idx_buf[:nnz] = torch.from_numpy(idx).to(dev)
        if homo:
            w_t = torch.from_numpy(w).to(dev)
        else:
            w_buf = torch.full((nnz + 64,), 1e30, dtype=torch.float32, device=dev)
            w_buf[:nnz] = torch.from_numpy(w).to(dev)
            w_t = w_buf[:nnz]
        got = be.binary_csrmv(w_t, idx_buf[:nnz], torch.from_numpy(ptr).to(dev), torch.from_numpy(v).to(dev), shape=(m, k),
                              transpose=False).cpu().numpy()
        np.testing.assert_allclose(got, oracle.binary_csrmv(w, idx, ptr, v, (m, k), False), rtol=RTOL, atol=ATOL, err_msg=str(lens))


def test_csrmv_int64_indptr(be, oracle):
    rng = np.random.default_rng(5)
    m, k = 64, 100
    w, idx, ptr = rand_csr(rng, m, k, rng.integers(0, 20, m))
    v = spikes_of(rng, m, 0.5, 'bool')
    for tr, vv in ((True, v), (False, spikes_of(rng, k, 0.5, 'bool'))):
        got = be.binary_csrmv(w, idx, ptr.astype(np.int64), vv, shape=(m, k), transpose=tr)
        np.testing.assert_allclose(got, oracle.binary_csrmv(w, idx, ptr, vv, (m, k), tr), rtol=RTOL, atol=ATOL)


def test_csrmv_empty(be):
    w = np.ones(1, np.float32)
    out = be.binary_csrmv(w, np.zeros(0, np.int32), np.zeros(4, np.int32), np.ones(3, bool), shape=(3, 5), transpose=True)
    assert out.shape == (5,) and not out.any()
    out = be.binary_csrmv(w, np.zeros(0, np.int32), np.zeros(4, np.int32), np.ones(5, bool), shape=(3, 5), transpose=False)
    assert out.shape == (3,) and not out.any()


# ---------------------------------------------------------------------------------------------------
# planned (post-sliced, LDS-accumulating) scatter route
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('homo', [True, False])
@pytest.mark.parametrize('kind', ['bool', 'float'])
@pytest.mark.parametrize('shift,parts', [(4, 1), (6, 3), (10, 2), (14, 4)])
def test_csrmv_plan_matches_oracle(be, oracle, homo, kind, shift, parts):
    from brainevent_amd._csr import ScatterPlan, _plan_call
    from brainevent_amd import _array as A
    if homo and shift == 14:
        shift = 15
    rng = np.random.default_rng(100 + shift)
    m, k = 500, 3000
    lens = rng.integers(0, 300, m)
    lens[::7] = 0
    w, idx, ptr = rand_csr(rng, m, k, lens, homo=homo)
    if not homo:
        w = rng.normal(0, 1, w.shape).astype(np.float32)   # mixed signs
    plan = ScatterPlan.build(w, idx, torch.tensor(ptr), shape=(m, k), slice_shift=shift)
    assert plan.blob.numel() % 128 == 0 and plan.nbytes() > 0
    v = spikes_of(rng, m, 0.2, kind)
    spikes, sd = A.spikes_to_device(v)
    out = torch.empty(k, dtype=torch.float32, device='cuda')
    _plan_call(plan, A.to_device(w), spikes, sd, out, parts=parts)
    ref = oracle.binary_csrmv(w.astype(np.float64), idx, ptr, v, (m, k), True)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=RTOL, atol=ATOL)
    # order independent accumulation: bitwise identical on repeat and for another part count
    out2 = torch.empty_like(out)
    _plan_call(plan, A.to_device(w), spikes, sd, out2, parts=max(1, parts - 1) if parts > 1 else 5)
    assert torch.equal(out, out2)


def test_csrmv_plan_homo_counts_exact(be, oracle):
    from brainevent_amd._csr import ScatterPlan, _plan_call
    from brainevent_amd import _array as A
    rng = np.random.default_rng(42)
    m, k = 2000, 70000
    w, idx, ptr = rand_csr(rng, m, k, [400] * m, homo=True)
    w[:] = 1.0
    plan = ScatterPlan.build(w, idx, torch.tensor(ptr), shape=(m, k))
    assert plan.slice_shift == 15 and plan.layout == ScatterPlan.LAYOUT_U16 and plan.n_slices == 12   # 400 per row / 32 per block
    v = spikes_of(rng, m, 0.1, 'bool')
    spikes, sd = A.spikes_to_device(v)
    out = torch.empty(k, dtype=torch.float32, device='cuda')
    _plan_call(plan, A.to_device(w), spikes, sd, out)
    ref = np.zeros(k, np.int64)
    row_of = np.repeat(np.arange(m), 400)
    np.add.at(ref, idx[v[row_of]], 1)
    assert np.array_equal(out.cpu().numpy().astype(np.int64), ref)


def test_csr_class_uses_plan_and_matches(be, oracle, monkeypatch):
    import brainevent_amd._csr as C
    monkeypatch.setattr(C, 'PLAN_MIN_NNZ', 1000)
    rng = np.random.default_rng(9)
    m, k = 600, 40000
    w, idx, ptr = rand_csr(rng, m, k, [300] * m)
    csr = be.CSR((w, idx, ptr), shape=(m, k))
    v = spikes_of(rng, m, 0.1, 'bool')
    got = be.BinaryArray(v) @ csr
    assert isinstance(csr.buffers['scatter_plan'], C.ScatterPlan)
    ref = oracle.binary_csrmv(w.astype(np.float64), idx, ptr, v, (m, k), True)
    np.testing.assert_allclose(got, ref, rtol=RTOL, atol=ATOL)
    # gather direction and CSC mirrors
    v2 = spikes_of(rng, k, 0.1, 'bool')
    np.testing.assert_allclose(csr @ be.BinaryArray(v2), oracle.binary_csrmv(w.astype(np.float64), idx, ptr, v2, (m, k), False),
                               rtol=RTOL, atol=ATOL)
    csc = csr.T   # shape (k, m): csc @ v (len m) scatters, v2 @ csc gathers
    np.testing.assert_allclose(csc @ be.BinaryArray(v), ref, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(be.BinaryArray(v2) @ csc, oracle.binary_csrmv(w.astype(np.float64), idx, ptr, v2, (m, k), False),
                               rtol=RTOL, atol=ATOL)


def test_coba_4k_plumbing_config(be, oracle):
    """BASELINE.json configs[0] (COBA_2005: 4000 neurons = 3200 E + 800 I, 80 synapses per neuron, weights 0.6 / 6.7,
    examples/COBA_2005.py:38-58): both projections through CSR + `@` for a few time steps against the oracle."""
    rng = np.random.default_rng(2005)
    n, n_exc, n_conn = 4000, 3200, 80
    def proj(n_pre, w):
        indptr = (np.arange(n_pre + 1) * n_conn).astype(np.int32)
        indices = rng.integers(0, n, n_pre * n_conn).astype(np.int32)
        return be.CSR((np.asarray([w], np.float32), indices, indptr), shape=(n_pre, n)), indices, indptr
    exc, ei, ep = proj(n_exc, 0.6)
    inh, ii, ip = proj(n - n_exc, 6.7)
    for step in range(5):
        spk = rng.random(n) < 0.05
        ge = be.BinaryArray(spk[:n_exc]) @ exc
        gi = be.BinaryArray(spk[n_exc:]) @ inh
        np.testing.assert_allclose(ge, oracle.binary_csrmv(np.asarray([0.6], np.float32), ei, ep, spk[:n_exc], (n_exc, n), True),
                                   rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(gi, oracle.binary_csrmv(np.asarray([6.7], np.float32), ii, ip, spk[n_exc:], (n - n_exc, n), True),
                                   rtol=1e-5, atol=1e-5)


# ---------------------------------------------------------------------------------------------------
# binned route (no plan): LDS counting sort into per-slice bins, then LDS accumulate
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('homo', [True, False])
@pytest.mark.parametrize('kind', ['bool', 'float'])
@pytest.mark.parametrize('shift,lens', [(6, 'ragged'), (10, 'ragged'), (14, 'fixed'), (8, 'long')])
def test_csrmv_binned_matches_oracle(be, oracle, homo, kind, shift, lens):
    from brainevent_amd._csr import BinnedScatter, _binned_call
    from brainevent_amd import _array as A
    rng = np.random.default_rng(7 + shift)
    m, k = 3000, 50000
    if lens == 'ragged':
        row_lens = rng.integers(0, 200, m); row_lens[::5] = 0
    elif lens == 'fixed':
        row_lens = np.full(m, 100)
    else:
        row_lens = rng.integers(0, 50, m); row_lens[[3, 1500]] = 40000      # rows longer than one LDS batch (carry)
    w, idx, ptr = rand_csr(rng, m, k, row_lens, homo=homo)
    if not homo:
        w = rng.normal(0, 1, w.shape).astype(np.float32)
    v = spikes_of(rng, m, 0.3, kind)
    v[[3, 1500]] = 1 if kind != 'float' else 1.0
    wd, idd, ptd = A.to_device(w), A.to_device(idx), A.to_device(ptr)
    spikes, sd = A.spikes_to_device(v)
    ws = BinnedScatter(wd, m, k, idx.size, max_active_fraction=0.6, slice_shift=shift)
    out = torch.empty(k, dtype=torch.float32, device='cuda')
    _binned_call(ws, wd, idd, ptd, -1, spikes, sd, out)
    ref = oracle.binary_csrmv(w.astype(np.float64), idx, ptr, v, (m, k), True)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=RTOL, atol=ATOL)


def test_csrmv_binned_overflow_falls_back_correctly(be, oracle):
    """Bins sized far too small: the overflowing runs go through global atomics; the result must still be right."""
    from brainevent_amd._csr import BinnedScatter, _binned_call
    from brainevent_amd import _array as A
    rng = np.random.default_rng(3)
    m, k = 4000, 40000
    w, idx, ptr = rand_csr(rng, m, k, np.full(m, 150))
    v = spikes_of(rng, m, 0.5, 'bool')
    wd, idd, ptd = A.to_device(w), A.to_device(idx), A.to_device(ptr)
    spikes, sd = A.spikes_to_device(v)
    ws = BinnedScatter(wd, m, k, idx.size, max_active_fraction=0.001, slice_shift=12)    # ~1/500 of what is needed
    out = torch.empty(k, dtype=torch.float32, device='cuda')
    _binned_call(ws, wd, idd, ptd, -1, spikes, sd, out)
    ref = oracle.binary_csrmv(w.astype(np.float64), idx, ptr, v, (m, k), True)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-5, atol=1e-4)


def test_fixed_num_uses_binned_route_when_rows_are_sparse_per_slice(be, oracle, monkeypatch):
    import brainevent_amd._csr as C
    monkeypatch.setattr(C, 'PLAN_MIN_NNZ', 1000)
    rng = np.random.default_rng(12)
    n_pre, n_post, K = 2000, 200000, 20            # 20 entries per row over 13 slices: < 8 per (row, slice)
    idx = rng.integers(0, n_post, (n_pre, K)).astype(np.int32)
    w = rng.uniform(0.1, 1, (n_pre, K)).astype(np.float32)
    conn = be.FixedNumPerPre((w, idx), shape=(n_pre, n_post))
    s = spikes_of(rng, n_pre, 0.05, 'bool')
    got = be.BinaryArray(s) @ conn
    assert isinstance(conn.buffers['scatter_plan'], C.BinnedScatter)
    np.testing.assert_allclose(got, oracle.binary_fcnmv(w.astype(np.float64), idx, s, (n_pre, n_post), True), rtol=RTOL, atol=ATOL)


def test_mirror_makes_gather_direction_event_driven(be, oracle, monkeypatch):
    """with the transposed mirror built, `CSR @ spk` / `spk @ CSC` scatter over the active columns."""
    import brainevent_amd._csr as C
    monkeypatch.setattr(C, 'PLAN_MIN_NNZ', 1000)
    rng = np.random.default_rng(31)
    m, k = 3000, 2500
    lens = rng.integers(100, 400, m)
    w, idx, ptr = rand_csr(rng, m, k, lens)
    csr = be.CSR((w, idx, ptr), shape=(m, k)).prepare(mirror=True)
    assert isinstance(csr.buffers['mirror'].plan, (C.ScatterPlan, C.BinnedScatter))
    v = spikes_of(rng, k, 0.05, 'bool')
    ref = oracle.binary_csrmv(w.astype(np.float64), idx, ptr, v, (m, k), False)
    np.testing.assert_allclose(csr @ be.BinaryArray(v), ref, rtol=RTOL, atol=ATOL)
    B = np.stack([spikes_of(rng, k, 0.05, 'bool') for _ in range(3)], axis=1)
    np.testing.assert_allclose(csr @ be.BinaryArray(B), oracle.binary_csrmm(w.astype(np.float64), idx, ptr, B, (m, k), False),
                               rtol=RTOL, atol=ATOL)
    csc = be.CSC((w, idx, ptr), shape=(k, m)).prepare(mirror=True)       # same arrays seen as CSC of the transpose
    np.testing.assert_allclose(be.BinaryArray(v) @ csc, ref, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(be.BinaryArray(B.T) @ csc, oracle.binary_csrmm(w.astype(np.float64), idx, ptr, B, (m, k), False).T,
                               rtol=RTOL, atol=ATOL)
    # the mirror is the exact transpose
    mr = csr.buffers['mirror']
    d = np.zeros((k, m)); np.add.at(d, (np.repeat(np.arange(k), np.diff(mr.indptr.cpu().numpy())), mr.indices.cpu().numpy()), mr.data.cpu().numpy())
    np.testing.assert_allclose(d.T, csr.todense(), rtol=1e-6)


# ---------------------------------------------------------------------------------------------------
# the reference's seeded-numpy operator tests (brainevent/_csr/main_test.py:1289-1358): same generator,
# same seeds, same shapes, same dense-equivalence assertion (atol 1e-5)
# ---------------------------------------------------------------------------------------------------
def _rand_dense(rng, m, n, p=0.4):
    return ((rng.random((m, n)) < p) * rng.random((m, n))).astype(np.float32)


def test_reference_seeded_operator_cases(be):
    rng = np.random.default_rng(3)                                   # main_test.py:1296-1307  csr @ ev
    csr = be.CSR.fromdense(_rand_dense(rng, 5, 7))
    ev = rng.random(7) > 0.5
    got = csr @ be.BinaryArray(ev)
    assert got.shape == (5,)
    np.testing.assert_allclose(got, csr.todense() @ ev.astype(np.float32), atol=1e-5)

    rng = np.random.default_rng(4)                                   # :1310-1320  ev @ csr
    csr = be.CSR.fromdense(_rand_dense(rng, 5, 7))
    ev = rng.random(5) > 0.5
    got = be.BinaryArray(ev) @ csr
    assert got.shape == (7,)
    np.testing.assert_allclose(got, ev.astype(np.float32) @ csr.todense(), atol=1e-5)

    rng = np.random.default_rng(5)                                   # :1323-1334  ev @ csc
    csc = be.CSC.fromdense(_rand_dense(rng, 6, 4))
    ev = rng.random(6) > 0.5
    got = be.BinaryArray(ev) @ csc
    assert got.shape == (4,)
    np.testing.assert_allclose(got, ev.astype(np.float32) @ csc.todense(), atol=1e-5)

    rng = np.random.default_rng(6)                                   # :1337-1347  csc @ ev
    csc = be.CSC.fromdense(_rand_dense(rng, 6, 4))
    ev = rng.random(4) > 0.5
    got = csc @ be.BinaryArray(ev)
    assert got.shape == (6,)
    np.testing.assert_allclose(got, csc.todense() @ ev.astype(np.float32), atol=1e-5)

    rng = np.random.default_rng(7)                                   # :1350-1358  with_data round trip
    csr = be.CSR.fromdense(_rand_dense(rng, 5, 7))
    ev = rng.random(7) > 0.5
    got = csr.with_data(csr.data) @ be.BinaryArray(ev)
    np.testing.assert_allclose(got, csr.todense() @ ev.astype(np.float32), atol=1e-5)


@pytest.mark.parametrize('homo', [True, False])
@pytest.mark.parametrize('float_events', [True, False])
def test_reference_binary_operator_matrix(be, homo, float_events):
    """Test_CSR_BinaryOperator-style sweep (brainevent/_csr/main_test.py:83-140): homo / hetero weights x bool /
    float-as-bool events, all four operator forms against the dense product."""
    rng = np.random.default_rng(11)
    m, k = 20, 40
    dense = _rand_dense(rng, m, k, p=0.1)
    if homo:
        dense = (dense != 0) * np.float32(1.5)
    csr = be.CSR.fromdense(dense)
    if homo:
        csr = be.CSR((np.asarray([1.5], np.float32), csr.indices.cpu().numpy(), csr.indptr.cpu().numpy()), shape=(m, k))
    x = rng.random(m) < 0.5
    y = rng.random(k) < 0.5
    xe = x.astype(np.float32) if float_events else x
    ye = y.astype(np.float32) if float_events else y
    np.testing.assert_allclose(be.BinaryArray(xe) @ csr, x.astype(np.float32) @ dense, atol=1e-5)
    np.testing.assert_allclose(csr @ be.BinaryArray(ye), dense @ y.astype(np.float32), atol=1e-5)
    csc = csr.T
    np.testing.assert_allclose(be.BinaryArray(ye) @ csc, y.astype(np.float32) @ dense.T, atol=1e-5)
    np.testing.assert_allclose(csc @ be.BinaryArray(xe), dense.T @ x.astype(np.float32), atol=1e-5)
    X = rng.random((3, m)) < 0.5
    np.testing.assert_allclose(be.BinaryArray(X) @ csr, X.astype(np.float32) @ dense, atol=1e-5)
    Y = rng.random((k, 3)) < 0.5
    np.testing.assert_allclose(csr @ be.BinaryArray(Y), dense @ Y.astype(np.float32), atol=1e-5)


def test_plan_refuses_weights_it_cannot_resolve_and_class_falls_back(be, oracle, monkeypatch):
    import brainevent_amd._csr as C
    monkeypatch.setattr(C, 'PLAN_MIN_NNZ', 1000)
    rng = np.random.default_rng(17)
    m, k = 400, 40000
    w, idx, ptr = rand_csr(rng, m, k, [200] * m)
    v = spikes_of(rng, m, 0.2, 'bool')
    # (a) extreme dynamic range: one weight 1e30, the rest ~1e-12 -> fixed point cannot hold both
    w_dr = (w * 1e-12).astype(np.float32); w_dr[5] = 1e30
    with pytest.raises(be.MathError):
        C.ScatterPlan.build(w_dr, idx, torch.tensor(ptr), shape=(m, k))
    csr = be.CSR((w_dr, idx, ptr), shape=(m, k))
    got = be.BinaryArray(v) @ csr                      # falls back to the direct route (float atomics)
    assert csr.buffers['scatter_plan'] is None
    ref = oracle.binary_csrmv(w_dr.astype(np.float64), idx, ptr, v, (m, k), True)
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-20)
    # (b) inf weights
    w_inf = w.copy(); w_inf[7] = np.inf
    csr = be.CSR((w_inf, idx, ptr), shape=(m, k))
    got = be.BinaryArray(np.ones(m, bool)) @ csr
    assert csr.buffers['scatter_plan'] is None and np.isinf(got[idx[7]])


@pytest.mark.parametrize('width', [None, 300, 511])
@pytest.mark.parametrize('homo', [True, False])
def test_scatter_plan_preserves_structure_bit_exactly(be, homo, width):
    """Decode the planned layout (seg table + 128-byte aligned blocks) back to (row, column, weight) triples on the
    host: it must be exactly the CSR's multiset — integer structure and weight bit patterns."""
    from brainevent_amd._csr import ScatterPlan
    rng = np.random.default_rng(99)
    m, k, shift = 300, 5000, 9
    lens = rng.integers(0, 120, m); lens[::11] = 0
    w, idx, ptr = rand_csr(rng, m, k, lens, homo=homo)
    plan = ScatterPlan.build(w, idx, torch.tensor(ptr), shape=(m, k), slice_shift=shift, slice_width=width or (1 << shift),
                             layout='u16')
    assert plan.layout == ScatterPlan.LAYOUT_U16
    seg = plan.seg.cpu().numpy().view(np.uint32).reshape(m, plan.n_slices, 2)
    blob = plan.blob.cpu().numpy()
    S = 1 << shift                       # pad marker / accumulator capacity
    Wd = plan.slice_width                # columns per slice
    assert Wd == (width or S) and plan.n_slices == -(-k // Wd)
    got = []
    for r in range(m):
        for s in range(plan.n_slices):
            start, ng = int(seg[r, s, 0]), int(seg[r, s, 1])
            if ng == 0:
                continue
            assert (start * 128) % 128 == 0
            base = start * 128
            if homo:
                cols = blob[base:base + ng * 16].view(np.uint16)
                ws = np.zeros(cols.size, np.uint32)
            else:
                ws = blob[base:base + ng * 16].view(np.uint32)
                cols = blob[base + ng * 16:base + ng * 24].view(np.uint16)
            for c, wb in zip(cols, ws):
                if c == S:                       # pad entry
                    assert wb == 0
                    continue
                assert c < Wd
                got.append((r, s * Wd + int(c), int(wb)))
    wbits = np.zeros(idx.size, np.uint32) if homo else w.view(np.uint32)
    rows = np.repeat(np.arange(m), np.diff(ptr))
    ref = sorted(zip(rows.tolist(), idx.tolist(), wbits.tolist()))
    assert sorted(got) == ref
    # and the product through this layout matches the oracle
    from oracle import oracle_np as O
    v = np.random.default_rng(5).random(m) < 0.3
    got_y = be.binary_csrmv(w, idx, ptr, v, shape=(m, k), transpose=True, workspace=plan)
    np.testing.assert_allclose(got_y, O.binary_csrmv(w, idx, ptr, v, (m, k), True), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('width,k', [(None, 5000), (300, 5000), (511, 5000), (16384, 70000), (None, 70000), (20000, 70000)])
def test_scatter_plan_d8_layout_decodes_bit_exactly(be, width, k):
    """The sorted uint8-delta layout: prefix sums of the deltas from the block's base column give back exactly the CSR's
    (row, column, weight bits) multiset; escapes / pads carry weight 0; gaps above 255 are bridged."""
    from brainevent_amd._csr import ScatterPlan
    from oracle import oracle_np as O
    rng = np.random.default_rng(123)
    m = 200
    shift = 9 if k == 5000 else 14
    lens = rng.integers(0, 150, m); lens[::13] = 0; lens[5] = 3000
    w, idx, ptr = rand_csr(rng, m, k, lens, homo=False)
    idx[ptr[7]:ptr[8]] = idx[ptr[7]]                       # a row that hits one column many times (delta 0)
    plan = ScatterPlan.build(w, idx, torch.tensor(ptr), shape=(m, k), slice_shift=shift, slice_width=width, layout='d8')
    assert plan.layout == ScatterPlan.LAYOUT_D8
    seg = plan.seg.cpu().numpy().view(np.uint32).reshape(m, plan.n_slices, 2)
    blob = plan.blob.cpu().numpy()
    Wd = plan.slice_width
    got, n_escape = [], 0
    for r in range(m):
        for s in range(plan.n_slices):
            start, y = int(seg[r, s, 0]), int(seg[r, s, 1])
            ng, base = y & 0xffff, y >> 16
            if ng == 0:
                continue
            b0 = start * 128
            ws = blob[b0:b0 + ng * 16].view(np.uint32)
            ds = blob[b0 + ng * 16:b0 + ng * 20]
            cols = base + np.cumsum(ds.astype(np.int64))
            assert cols.max() < Wd
            for c, wb, d in zip(cols, ws, ds):
                if wb == 0 and (w.view(np.uint32) != 0).all():      # escape or pad
                    n_escape += int(d == 255)
                    continue
                got.append((r, s * Wd + int(c), int(wb)))
    rows = np.repeat(np.arange(m), np.diff(ptr))
    ref = sorted(zip(rows.tolist(), idx.tolist(), w.view(np.uint32).tolist()))
    assert sorted(got) == ref
    if k == 70000:
        assert n_escape > 0                                     # sparse rows over wide slices need escapes
    for fire in (0.3, 1.0):
        v = rng.random(m) < fire
        got_y = be.binary_csrmv(w, idx, ptr, v, shape=(m, k), transpose=True, workspace=plan)
        np.testing.assert_allclose(got_y, O.binary_csrmv(w, idx, ptr, v, (m, k), True), rtol=1e-5, atol=1e-5)
    # same numbers, bit for bit, as the uint16 layout (integer sums do not depend on the entry order)
    plan16 = ScatterPlan.build(w, idx, torch.tensor(ptr), shape=(m, k), slice_shift=shift,
                               slice_width=None if width is None else min(width, 1 << shift), layout='u16')
    v = rng.random(m) < 0.5
    np.testing.assert_array_equal(be.binary_csrmv(w, idx, ptr, v, shape=(m, k), transpose=True, workspace=plan),
                                  be.binary_csrmv(w, idx, ptr, v, shape=(m, k), transpose=True, workspace=plan16))


@pytest.mark.parametrize('width,k', [(None, 5000), (300, 5000), (511, 5000), (32768, 150000), (None, 150000), (40000, 150000)])
def test_scatter_plan_h8_layout_decodes_bit_exactly(be, width, k):
    """The homogeneous-weight delta layout: one byte per entry, code c < 255 = advance c columns and count, 255 = advance
    255 columns without counting (escape / tail pad).  Decoding gives back exactly the CSR's (row, column) multiset."""
    from brainevent_amd._csr import ScatterPlan
    from oracle import oracle_np as O
    rng = np.random.default_rng(321)
    m = 200
    shift = 9 if k == 5000 else 15
    lens = rng.integers(0, 150, m); lens[::13] = 0; lens[5] = 3000
    w, idx, ptr = rand_csr(rng, m, k, lens, homo=True)
    idx[ptr[7]:ptr[8]] = idx[ptr[7]]                       # one column many times (code 0)
    idx[ptr[9]:ptr[9] + 2] = [0, 255]                      # a gap of exactly 255: escape + code 0
    plan = ScatterPlan.build(w, idx, torch.tensor(ptr), shape=(m, k), slice_shift=shift, slice_width=width, layout='h8')
    assert plan.layout == ScatterPlan.LAYOUT_H8
    seg = plan.seg.cpu().numpy().view(np.uint32).reshape(m, plan.n_slices, 2)
    blob = plan.blob.cpu().numpy()
    Wd = plan.slice_width
    got, n_escape = [], 0
    for r in range(m):
        for s in range(plan.n_slices):
            start, y = int(seg[r, s, 0]), int(seg[r, s, 1])
            ng, base = y & 0xffff, y >> 16
            if ng == 0:
                continue
            codes = blob[start * 128:start * 128 + ng * 8]
            cols = base + np.cumsum(codes.astype(np.int64))
            real = codes != 255
            assert cols[real].max() < Wd and codes[0] == 0
            n_escape += int((~real).sum())
            got += [(r, s * Wd + int(c)) for c in cols[real]]
    rows = np.repeat(np.arange(m), np.diff(ptr))
    assert sorted(got) == sorted(zip(rows.tolist(), idx.tolist()))
    assert n_escape > 0
    plan16 = ScatterPlan.build(w, idx, torch.tensor(ptr), shape=(m, k), slice_shift=shift,
                               slice_width=None if width is None else min(width, 1 << shift), layout='u16')
    assert plan16.layout == ScatterPlan.LAYOUT_U16
    for fire in (0.3, 1.0):
        v = rng.random(m) < fire
        got_y = be.binary_csrmv(w, idx, ptr, v, shape=(m, k), transpose=True, workspace=plan)
        np.testing.assert_allclose(got_y, O.binary_csrmv(w, idx, ptr, v, (m, k), True), rtol=1e-6, atol=1e-6)
        np.testing.assert_array_equal(got_y, be.binary_csrmv(w, idx, ptr, v, shape=(m, k), transpose=True, workspace=plan16))


@pytest.mark.parametrize('seed', range(int(__import__('os').environ.get('BE_STRESS_SEEDS', 8))))
def test_h8_layout_randomized(be, oracle, seed):
    """Random shapes / row-length styles / widths through the h8 layout: counts are integers, so every result equals the
    uint16 layout's bit for bit and the oracle's to float rounding; vectors, batches and the part split included."""
    from brainevent_amd._csr import ScatterPlan, _plan_call
    from brainevent_amd import _array as A
    rng = np.random.default_rng(7000 + seed)
    m = int(rng.integers(1, 900))
    k = int(rng.choice([37, 4000, 41000, 300000]))
    shift = int(rng.choice([6, 11, 15]))
    width = None if rng.random() < 0.5 else int(rng.integers(16, min(k, 40000 if shift == 15 else 1 << shift) + 1))
    style = seed % 4
    if style == 0:
        lens = rng.integers(0, 60, m)
    elif style == 1:
        lens = rng.integers(0, 3000, m)
    elif style == 2:
        lens = np.where(rng.random(m) < 0.2, rng.integers(1000, 16384, m), rng.integers(0, 5, m))
    else:
        lens = rng.integers(0, 400, m)
    if k // max(width or (1 << shift), 1) > 1000:
        shift, width = 15, None
    dtype = np.float16 if seed % 5 == 4 else np.float32
    w, idx, ptr = rand_csr(rng, m, k, lens, dtype=dtype, homo=True)
    if style == 3 and idx.size:
        idx[:] = (idx // 1000 * 1000 + idx % 3).clip(0, k - 1)
    if seed % 3 == 1:                                         # canonical rows (ascending columns) skip the build's sort
        for r in range(0, m, 2 if seed % 2 else 1):
            idx[ptr[r]:ptr[r + 1]].sort()
    plan = ScatterPlan.build(w, idx, torch.tensor(ptr), shape=(m, k), slice_shift=shift, slice_width=width, layout='h8')
    assert plan.layout == ScatterPlan.LAYOUT_H8
    w16 = None if width is None else min(width, 1 << shift)
    plan16 = ScatterPlan.build(w, idx, torch.tensor(ptr), shape=(m, k), slice_shift=shift, slice_width=w16, layout='u16')
    tol = 1e-6 if dtype == np.float32 else 2e-3
    for fire in (0.05, 0.5, 1.0):
        v = rng.random(m) < fire
        got = be.binary_csrmv(w, idx, ptr, v, shape=(m, k), transpose=True, workspace=plan)
        ref = oracle.binary_csrmv(w.astype(np.float32), idx, ptr, v, (m, k), True)
        with np.errstate(over='ignore'):
            ref = ref.astype(dtype).astype(np.float32)       # f16 outputs overflow to inf above 65504, as the product's do
        fin = np.isfinite(ref)
        np.testing.assert_allclose(np.asarray(got, np.float32), ref, rtol=tol, atol=tol * max(1.0, float(np.abs(ref[fin]).max()) if fin.any() else 1.0))
        np.testing.assert_array_equal(got, be.binary_csrmv(w, idx, ptr, v, shape=(m, k), transpose=True, workspace=plan16))
        sp, sd = A.spikes_to_device(v)
        out_p = torch.empty(k, dtype=A.to_device(w).dtype, device='cuda')
        _plan_call(plan, A.to_device(w), sp, sd, out_p, parts=3)          # the multi-launch path, three parts
        np.testing.assert_array_equal(out_p.cpu().numpy(), np.asarray(got))
    B = rng.random((m, 5)) < 0.3
    got = be.binary_csrmm(w, idx, ptr, B, shape=(m, k), transpose=True, workspace=plan)
    np.testing.assert_array_equal(got, be.binary_csrmm(w, idx, ptr, B, shape=(m, k), transpose=True, workspace=plan16))


@pytest.mark.parametrize('k', [900, 30000, 120000])
def test_d8_quarter_wave_kernel_equals_full_wave(be, oracle, k):
    """The d8 step kernel has three decoders — a wave per block, and 16 or 8 lanes per block for plans of short blocks
    (``block_hint`` <= 43 / <= 18) — over the same blob.  Same bits from all on rows of 0 ... 5000 entries (blocks from empty
    to many passes: the sub-wave decoders finish long blocks in their serial tail), vectors and batches; the hint only
    selects the kernel."""
    from brainevent_amd._csr import ScatterPlan
    rng = np.random.default_rng(k)
    m = 700
    lens = np.where(rng.random(m) < 0.1, rng.integers(200, 5000, m), rng.integers(0, 40, m))
    lens[::17] = 0
    w, idx, ptr = rand_csr(rng, m, k, lens)
    w = rng.normal(0, 1, w.shape).astype(np.float32)
    plan = ScatterPlan.build(w, idx, torch.tensor(ptr), shape=(m, k), layout='d8')
    # deterministic comparison on fixed vectors
    vs = [np.random.default_rng(s).random(m) < f for s, f in ((1, 0.05), (2, 0.6), (3, 1.1), (4, -1.0))]
    B = np.stack([np.random.default_rng(9).random(m) < 0.3 for _ in range(3)], axis=1)
    ref = None
    for hint in (1, 18, 19, 43, 44, 100000):          # 8 lanes, 16 lanes, a wave per block
        plan.block_hint_override = hint
        got = [np.asarray(be.binary_csrmv(w, idx, ptr, v, shape=(m, k), transpose=True, workspace=plan)) for v in vs]
        got.append(np.asarray(be.binary_csrmm(w, idx, ptr, B, shape=(m, k), transpose=True, workspace=plan)))
        if ref is None:
            ref = got
            for v, g in zip(vs, got):
                np.testing.assert_allclose(g, oracle.binary_csrmv(w, idx, ptr, v, (m, k), True), rtol=1e-5, atol=1e-5)
        else:
            for a, b in zip(ref, got):
                np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize('k', [50000, 200000])
def test_counted_sub_wave_decoders_equal_full_wave(be, oracle, k):
    """Counted entries (one shared weight, uint16 layout): 4, 8, 16 or 32 lanes per block for plans of short blocks, a wave
    per block otherwise — the hint only selects the decoder; rows of 0 ... 5000 entries, vectors and batches."""
    from brainevent_amd._csr import ScatterPlan
    rng = np.random.default_rng(k + 1)
    m = 700
    lens = np.where(rng.random(m) < 0.1, rng.integers(200, 5000, m), rng.integers(0, 40, m))
    lens[::17] = 0
    w, idx, ptr = rand_csr(rng, m, k, lens, homo=True)
    plan = ScatterPlan.build(w, idx, torch.tensor(ptr), shape=(m, k), layout='u16')
    assert plan.n_slices > 1
    vs = [np.random.default_rng(s).random(m) < f for s, f in ((1, 0.05), (2, 0.6), (3, 1.1), (4, -1.0))]
    B = np.stack([np.random.default_rng(9).random(m) < 0.3 for _ in range(3)], axis=1)
    ref = None
    for hint in (1, 18, 19, 43, 44, 96, 97, 210, 211, 100000):
        plan.block_hint_override = hint
        got = [np.asarray(be.binary_csrmv(w, idx, ptr, v, shape=(m, k), transpose=True, workspace=plan)) for v in vs]
        got.append(np.asarray(be.binary_csrmm(w, idx, ptr, B, shape=(m, k), transpose=True, workspace=plan)))
        if ref is None:
            ref = got
            for v, g in zip(vs, got):
                np.testing.assert_allclose(g, oracle.binary_csrmv(w, idx, ptr, v, (m, k), True), rtol=1e-6, atol=1e-6)
        else:
            for a, b in zip(ref, got):
                np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize('k,width', [(50000, None), (200000, None), (60000, 1000)])
def test_weighted_sub_wave_decoders_equal_full_wave(be, oracle, k, width):
    """Weighted entries in the uint16 layout: 4, 8, 16 or 32 lanes per block for plans of short blocks (long blocks finish in
    the serial tail), a wave per block otherwise.  With 60 slices (``width`` 1000) and a hint <= 64 the step also runs on the
    pre-gathered segment table (``k_gather_seg``).  Fixed-point sums: every variant returns the same bits."""
    from brainevent_amd._csr import ScatterPlan
    rng = np.random.default_rng(k + 2)
    m = 900
    lens = np.where(rng.random(m) < 0.1, rng.integers(200, 5000, m), rng.integers(0, 40, m))
    lens[::17] = 0
    w, idx, ptr = rand_csr(rng, m, k, lens)
    w = rng.normal(0, 1, w.shape).astype(np.float32)
    plan = ScatterPlan.build(w, idx, torch.tensor(ptr), shape=(m, k), layout='u16', slice_width=width)
    assert plan.n_slices > 1 and (width is None or plan.n_slices >= 40)
    vs = [np.random.default_rng(s).random(m) < f for s, f in ((1, 0.05), (2, 0.6), (3, 1.1), (4, -1.0))]
    vs.append(np.where(vs[1], 2.5, 0.0).astype(np.float32))          # float spikes: compacted list, not the fused step
    B = np.stack([np.random.default_rng(9).random(m) < 0.3 for _ in range(3)], axis=1)
    ref = None
    for hint in (100000, 1, 7, 8, 18, 19, 43, 44, 64, 65, 96, 97):
        plan.block_hint_override = hint
        got = [np.asarray(be.binary_csrmv(w, idx, ptr, v, shape=(m, k), transpose=True, workspace=plan)) for v in vs]
        got.append(np.asarray(be.binary_csrmm(w, idx, ptr, B, shape=(m, k), transpose=True, workspace=plan)))
        if ref is None:
            ref = got
            for v, g in zip(vs, got):
                np.testing.assert_allclose(g, oracle.binary_csrmv(w, idx, ptr, v > 0, (m, k), True), rtol=1e-5, atol=1e-5)
        else:
            for a, b in zip(ref, got):
                np.testing.assert_array_equal(a, b)


def test_pre_gathered_segment_table_counted_and_packed(be, oracle):
    """>= 40 slices of short blocks: the active rows' segment entries are gathered into the workspace first (FUSED = 3).
    Counted entries, every spike encoding (bool, float, bit-packed events), part counts that do and do not divide the list."""
    from brainevent_amd._csr import ScatterPlan, _plan_call
    from brainevent_amd import _array as A
    rng = np.random.default_rng(77)
    m, k = 3000, 48000
    lens = rng.integers(0, 300, m)
    w, idx, ptr = rand_csr(rng, m, k, lens, homo=True)
    plan = ScatterPlan.build(w, idx, torch.tensor(ptr), shape=(m, k), layout='u16', slice_width=800)
    assert plan.n_slices == 60 and plan.block_hint <= 64
    for fire in (0.0, 0.01, 0.3, 1.0):
        v = rng.random(m) < fire
        ref = oracle.binary_csrmv(w, idx, ptr, v, (m, k), True)
        np.testing.assert_array_equal(be.binary_csrmv(w, idx, ptr, v, shape=(m, k), transpose=True, workspace=plan), ref)
        np.testing.assert_array_equal(be.binary_csrmv(w, idx, ptr, np.where(v, 0.5, -1.0).astype(np.float32), shape=(m, k),
                                                      transpose=True, workspace=plan), ref)
        for parts in (1, 3, 4, 7):
            sp, sd = A.spikes_to_device(v)
            out = torch.empty(k, dtype=torch.float32, device='cuda')
            _plan_call(plan, A.to_device(w), sp, sd, out, parts=parts)
            np.testing.assert_array_equal(out.cpu().numpy(), ref)
        csr = be.CSR((w, idx, ptr), shape=(m, k))
        csr.buffers['scatter_plan'] = plan
        got = be.BitPackedBinary(v) @ csr
        np.testing.assert_array_equal(got.cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got), ref)


def test_d8_layout_falls_back_when_it_does_not_apply(be):
    from brainevent_amd._csr import ScatterPlan
    rng = np.random.default_rng(4)
    m, k = 3, 40000
    lens = [20000, 10, 0]                                    # a row longer than the LDS sort
    w, idx, ptr = rand_csr(rng, m, k, lens, homo=False)
    plan = ScatterPlan.build(w, idx, torch.tensor(ptr), shape=(m, k))
    assert plan.layout == ScatterPlan.LAYOUT_U16
    with pytest.raises(ValueError):
        ScatterPlan.build(w, idx, torch.tensor(ptr), shape=(m, k), layout='d8')
    wh, idxh, ptrh = rand_csr(rng, 50, 3000, [40] * 50, homo=True)
    assert ScatterPlan.build(wh, idxh, torch.tensor(ptrh), shape=(50, 3000)).layout == ScatterPlan.LAYOUT_U16    # auto: too small for h8
    assert ScatterPlan.build(wh, idxh, torch.tensor(ptrh), shape=(50, 3000), layout='h8').layout == ScatterPlan.LAYOUT_H8
    for wrong, args in (('d8', (wh, idxh, ptrh, (50, 3000))), ('h8', (w, idx, ptr, (m, k)))):
        with pytest.raises(ValueError):
            ScatterPlan.build(args[0], args[1], torch.tensor(args[2]), shape=args[3], layout=wrong)
    wl, idxl, ptrl = rand_csr(rng, m, k, lens, homo=True)           # the long row again, one weight
    assert ScatterPlan.build(wl, idxl, torch.tensor(ptrl), shape=(m, k)).layout == ScatterPlan.LAYOUT_U16
    v = rng.random(m) < 0.7
    from oracle import oracle_np as O
    np.testing.assert_allclose(be.binary_csrmv(w, idx, ptr, v, shape=(m, k), transpose=True, workspace=plan),
                               O.binary_csrmv(w, idx, ptr, v, (m, k), True), rtol=1e-5, atol=1e-5)


def test_balanced_slice_width():
    from brainevent_amd._csr import ScatterPlan
    assert ScatterPlan.balanced_width(1_000_000, 14) == 15625          # 64 slices x 4 parts = 256 workgroups
    assert ScatterPlan.balanced_width_cap(1_000_000, 20000) == 19608   # d8: 51 slices x 5 parts = 255 workgroups
    assert ScatterPlan.balanced_width(1_000_000, 15) == 31250          # 32 slices x 8 parts
    for k, shift in ((5, 4), (16, 4), (17, 4), (4000, 12), (125_000, 14), (10_000_000, 14), (70001, 9), (1 << 24, 14)):
        wd = ScatterPlan.balanced_width(k, shift)
        n = -(-k // wd)
        assert 1 <= wd <= (1 << shift) and n * wd >= k
        assert n <= max(256, (-(-k // (1 << shift)) + 7) // 8 * 8)


def test_fixed_point_gate_accepts_wide_uniform_weights(be, oracle):
    """U[0,1) weights with a few values far below the fixed-point resolution (what 1e10 samples contain at C2 scale): the global-minimum test alone would
    refuse the plan; the per-column maxima accept it, and the result is within tolerance."""
    from brainevent_amd._csr import ScatterPlan, BinnedScatter
    rng = np.random.default_rng(23)
    m, k = 2000, 3000
    w, idx, ptr = rand_csr(rng, m, k, [100] * m)
    w[rng.integers(0, w.size, 50)] = np.float32(2.0 ** -38)      # < 2^(16 - scale_exp) = 2^-35 for 2000 rows
    plan = ScatterPlan.build(w, idx, torch.tensor(ptr), shape=(m, k))
    v = spikes_of(rng, m, 0.3, 'bool')
    ref = oracle.binary_csrmv(w, idx, ptr, v, (m, k), True)
    got = be.binary_csrmv(w, idx, ptr, v, shape=(m, k), transpose=True, workspace=plan)
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6)
    wd, idd = torch.from_numpy(w).cuda(), torch.from_numpy(idx).cuda()
    with pytest.raises(be.MathError):            # without the structure only the global test is available
        BinnedScatter(wd, m, k, idx.size)
    ws = BinnedScatter(wd, m, k, idx.size, indices=idd)
    got = be.binary_csrmv(wd, idd, torch.from_numpy(ptr).cuda(), torch.from_numpy(v).cuda(), shape=(m, k), transpose=True,
                          workspace=ws)
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-5, atol=1e-6)
    # a column made only of unresolvable weights is still refused
    w2 = w.copy(); col = idx[0]; w2[idx == col] = np.float32(2.0 ** -45)
    with pytest.raises(be.MathError):
        ScatterPlan.build(w2, idx, torch.tensor(ptr), shape=(m, k))


@pytest.mark.parametrize('seed', range(int(__import__('os').environ.get('BE_STRESS_SEEDS', 12))))
def test_d8_layout_randomized_against_oracle_and_u16(be, oracle, seed):
    """Random shapes through the d8 layout: long blocks (> 256 entries: the carried prefix of the tail chunks), gaps far
    above 255 (chains of escapes), duplicates, empty rows, odd slice widths, batches, f16 weights.  Checked against the
    oracle and, bit for bit, against the u16 layout."""
    from brainevent_amd._csr import ScatterPlan
    rng = np.random.default_rng(1000 + seed)
    m = int(rng.integers(1, 400))
    k = int(rng.choice([7, 300, 5000, 40000, 250000]))
    shift = int(rng.integers(4, 15))
    cap = 1 << shift
    if -(-k // cap) > 1024:
        shift = max(shift, int(np.ceil(np.log2(k / 1024))) + 1)
        cap = 1 << shift
    width = int(rng.integers(max(1, -(-k // 1024)), cap + 1)) if rng.random() < 0.7 else None
    style = seed % 4
    if style == 0:
        lens = rng.integers(0, 60, m)
    elif style == 1:
        lens = rng.integers(0, 3000, m)                       # long blocks when there are few slices
    elif style == 2:
        lens = np.where(rng.random(m) < 0.2, rng.integers(1000, 16384, m), rng.integers(0, 5, m))
    else:
        lens = rng.integers(0, 400, m)
    dtype = np.float16 if seed % 5 == 4 else np.float32
    w, idx, ptr = rand_csr(rng, m, k, lens, dtype=dtype)
    if style == 3 and idx.size:                               # clustered columns: many zero / tiny deltas and huge gaps
        idx[:] = (idx // 1000 * 1000 + idx % 3).clip(0, k - 1)
    if seed % 3 != 0:                                         # canonical rows (ascending columns) skip the build's sort:
        for r in range(m):                                    # all of them, or every other one
            if seed % 3 == 1 or r % 2 == 0:
                idx[ptr[r]:ptr[r + 1]].sort()
    plan = ScatterPlan.build(w, idx, torch.tensor(ptr), shape=(m, k), slice_shift=shift, slice_width=width, layout='d8')
    assert plan.layout == ScatterPlan.LAYOUT_D8
    w16 = None if width is None else min(width, cap)
    plan16 = ScatterPlan.build(w, idx, torch.tensor(ptr), shape=(m, k), slice_shift=shift, slice_width=w16, layout='u16')
    # the speed hint picks the decoder (lanes per block, serial tails for longer blocks, pre-gathered segment table with
    # >= 40 slices): any value has to give the same bits
    for p_ in (plan, plan16):
        h = [None, 1, 10, 30, 60, 100000][int(rng.integers(0, 6))]
        if h is not None:
            p_.block_hint_override = h
    tol = 1e-5 if dtype == np.float32 else 2e-2
    for fire in (0.05, 0.5, 1.0):
        v = rng.random(m) < fire
        got = be.binary_csrmv(w, idx, ptr, v, shape=(m, k), transpose=True, workspace=plan)
        ref = oracle.binary_csrmv(w.astype(np.float32), idx, ptr, v, (m, k), True)
        np.testing.assert_allclose(np.asarray(got, np.float32), ref, rtol=tol, atol=tol * max(1.0, float(np.abs(ref).max())))
        np.testing.assert_array_equal(got, be.binary_csrmv(w, idx, ptr, v, shape=(m, k), transpose=True, workspace=plan16))
    B = rng.random((m, 5)) < 0.3
    got = be.binary_csrmm(w, idx, ptr, B, shape=(m, k), transpose=True, workspace=plan)
    np.testing.assert_array_equal(got, be.binary_csrmm(w, idx, ptr, B, shape=(m, k), transpose=True, workspace=plan16))


def test_fixed_point_exponent_bounds_columns_not_rows(be, oracle):
    """Rows that list the same column many times: a column receives far more addends than there are rows, so the
    exponent has to come from the largest column sum (rows x max|w| would let the 64-bit sums wrap)."""
    from brainevent_amd._csr import ScatterPlan, BinnedScatter
    rng = np.random.default_rng(3)
    m, k = 20, 4
    w, idx, ptr = rand_csr(rng, m, k, [4000] * m)
    v = np.ones(m, bool)
    ref = oracle.binary_csrmv(w, idx, ptr, v, (m, k), True)            # ~ 11000 per output, 20000 addends each
    for layout in ('d8', 'u16'):
        plan = ScatterPlan.build(w, idx, torch.tensor(ptr), shape=(m, k), layout=layout)
        np.testing.assert_allclose(be.binary_csrmv(w, idx, ptr, v, shape=(m, k), transpose=True, workspace=plan), ref,
                                   rtol=1e-5)
    wd, idd = torch.from_numpy(w).cuda(), torch.from_numpy(idx).cuda()
    for ind in (idd, None):
        ws = BinnedScatter(wd, m, k, idx.size, indices=ind, max_active_fraction=1.0)
        got = be.binary_csrmv(wd, idd, torch.from_numpy(ptr).cuda(), torch.from_numpy(v).cuda(), shape=(m, k), transpose=True,
                              workspace=ws)
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-5)


@pytest.mark.parametrize('seed', range(int(__import__('os').environ.get('BE_STRESS_SEEDS', 8))))
def test_binned_and_gather_routes_randomized(be, oracle, seed):
    """Random shapes through the binned scatter route (with and without room in the bins) and the gather kernels
    (sub-wave, wave-per-row, fused-over-batch), duplicates and empty rows included."""
    from brainevent_amd._csr import BinnedScatter
    rng = np.random.default_rng(2000 + seed)
    m = int(rng.integers(1, 600))
    k = int(rng.choice([9, 700, 33000, 200000]))
    lens = [rng.integers(0, 8, m), rng.integers(0, 300, m), rng.integers(200, 3000, m),
            np.where(rng.random(m) < 0.1, rng.integers(2000, 9000, m), rng.integers(0, 3, m))][seed % 4]
    homo = bool(seed & 1)
    w, idx, ptr = rand_csr(rng, m, k, lens, homo=homo)
    wd, idd, ptd = torch.from_numpy(w).cuda(), torch.from_numpy(idx).cuda(), torch.from_numpy(ptr).cuda()
    if idx.size:
        shift = int(rng.integers(6, 15)) if k > 64 else 4
        while not BinnedScatter.serves(k, shift, homo):        # more bins than pass B's LDS blocks hold: wider slices
            shift += 1
        for frac in (1.0, 0.002):
            ws = BinnedScatter(wd, m, k, idx.size, indices=idd, max_active_fraction=frac, slice_shift=shift)
            for fire in (0.1, 1.0):
                v = rng.random(m) < fire
                got = be.binary_csrmv(wd, idd, ptd, torch.from_numpy(v).cuda(), shape=(m, k), transpose=True, workspace=ws)
                ref = oracle.binary_csrmv(w, idx, ptr, v, (m, k), True)
                np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(ref).max())))
    for fire in (0.05, 0.6):
        s = rng.random(k) < fire
        ref = oracle.binary_csrmv(w, idx, ptr, s, (m, k), False)
        got = be.binary_csrmv(w, idx, ptr, s, shape=(m, k), transpose=False)
        np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(ref).max())))
    B = rng.random((k, int(rng.integers(2, 40)))) < 0.1
    ref = oracle.binary_csrmm(w, idx, ptr, B, (m, k), False)
    got = be.binary_csrmm(w, idx, ptr, B, shape=(m, k), transpose=False)
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(ref).max())))


@pytest.mark.parametrize('seed', range(int(__import__('os').environ.get('BE_STRESS_SEEDS', 10))))
def test_class_surface_randomized_including_square_shapes(be, monkeypatch, seed):
    """CSR / CSC / JITC / dense through the ``@`` operator, all four operand orders, vectors and batches — with batch
    sizes deliberately equal to the matrix dimensions now and then (orientation must not be guessed from shapes).
    The checker is the dense view of the same matrix."""
    import brainevent_amd._csr as C
    rng = np.random.default_rng(6000 + seed)
    monkeypatch.setattr(C, 'PLAN_MIN_NNZ', int(rng.choice([1, 10 ** 9])))
    m, k = int(rng.integers(1, 120)), int(rng.integers(1, 120))
    if seed % 3 == 0:
        k = m                                                  # square matrix
    nb = int(rng.choice([m, k, 1, 7]))                         # batch size colliding with a dimension
    lens = rng.integers(0, 30, m)
    w, idx, ptr = rand_csr(rng, m, k, lens, homo=bool(seed & 1))
    csr = be.CSR((w, idx, ptr), shape=(m, k))
    if seed % 4 == 0:
        csr.prepare(mirror=True)
    dense = csr.todense().astype(np.float64)
    assert dense.shape == (m, k)
    tol = dict(rtol=1e-5, atol=1e-4)
    sm, sk = rng.random(m) < 0.4, rng.random(k) < 0.4
    Sm_b, Sk_b = rng.random((nb, m)) < 0.4, rng.random((nb, k)) < 0.4          # batch-major events for ``S @ M``
    Sm_c, Sk_c = rng.random((m, nb)) < 0.4, rng.random((k, nb)) < 0.4          # column-major events for ``M @ S``
    for M, D in ((csr, dense), (csr.T, dense.T)):
        rows, cols = D.shape
        v_r, v_c = (sm, sk) if M is csr else (sk, sm)          # events matching M's rows / columns
        np.testing.assert_allclose(be.BinaryArray(v_r) @ M, v_r.astype(np.float64) @ D, **tol)
        np.testing.assert_allclose(M @ be.BinaryArray(v_c), D @ v_c.astype(np.float64), **tol)
        B_r = (Sm_b if M is csr else Sk_b)
        B_c = (Sk_c if M is csr else Sm_c)
        got = be.BinaryArray(B_r) @ M
        assert got.shape == (nb, cols)
        np.testing.assert_allclose(got, B_r.astype(np.float64) @ D, **tol)
        got = M @ be.BinaryArray(B_c)
        assert got.shape == (rows, nb)
        np.testing.assert_allclose(got, D @ B_c.astype(np.float64), **tol)
    # dense right / left operands
    W = rng.standard_normal((m, k)).astype(np.float32)
    np.testing.assert_allclose(be.BinaryArray(Sm_b) @ W, Sm_b.astype(np.float64) @ W, **tol)
    np.testing.assert_allclose(W @ be.BinaryArray(Sk_c), W.astype(np.float64) @ Sk_c, **tol)
    # JIT connectivity, both class kinds: compare with the materialised matrix
    for cls in (be.JITCUniformR, be.JITCUniformC):
        J = cls((np.float32(-0.5), np.float32(1.0), 0.2, 100 + seed), shape=(m, k), corder=bool(seed & 2))
        Jd = J.tocsr('mv').todense().astype(np.float64)
        np.testing.assert_allclose(be.BinaryArray(sm) @ J, sm.astype(np.float64) @ Jd, **tol)
        np.testing.assert_allclose(J @ be.BinaryArray(sk), Jd @ sk.astype(np.float64), **tol)
        Jm = J.tocsr('mm').todense().astype(np.float64)
        got = be.BinaryArray(Sm_b) @ J
        assert got.shape == (nb, k)
        np.testing.assert_allclose(got, Sm_b.astype(np.float64) @ Jm, **tol)
        got = J @ be.BinaryArray(Sk_c)
        assert got.shape == (m, nb)
        np.testing.assert_allclose(got, Jm @ Sk_c.astype(np.float64), **tol)


@pytest.mark.parametrize('seed', range(int(__import__('os').environ.get('BE_STRESS_SEEDS', 10))))
def test_csr_type_zoo_randomized(be, oracle, monkeypatch, seed):
    """Random combinations of structure dtypes (int32 / int64 indptr, int64 indices that are cast), weight dtypes, spike
    dtypes and containers (numpy / device tensors), routes (plan / direct), both directions, vector and batch."""
    import brainevent_amd._csr as C
    rng = np.random.default_rng(7000 + seed)
    monkeypatch.setattr(C, 'PLAN_MIN_NNZ', int(rng.choice([1, 10 ** 9])))
    m, k = int(rng.integers(1, 200)), int(rng.integers(1, 3000))
    lens = rng.integers(0, 50, m)
    homo = bool(rng.integers(0, 2))
    w, idx, ptr = rand_csr(rng, m, k, lens, homo=homo)
    wdt = [np.float32, np.float64, np.float16][int(rng.integers(0, 3))]
    w = w.astype(wdt)
    ptr = ptr.astype([np.int32, np.int64][int(rng.integers(0, 2))])
    idx_in = idx.astype([np.int32, np.int64][int(rng.integers(0, 2))])
    on_device = bool(rng.integers(0, 2))
    conv = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()) if on_device else (lambda a: a)
    csr = be.CSR((conv(w), conv(idx_in), conv(ptr)), shape=(m, k))
    assert csr.indices.dtype == torch.int32
    # no weight dtype is left to the global-atomic route once a matrix is worth a workspace (>= PLAN_MIN_NNZ entries and at
    # least PLAN_MIN_SEGMENT_NO_BINNED entries per (row, slice) block, or the binned route applies)
    nse = int(idx.size)
    if nse >= C.PLAN_MIN_NNZ and m > 0 and k > 0:
        route = C.choose_scatter_route(nse, m, k, csr.data)
        n_sl = -(-k // C.ScatterPlan.auto_geometry(m, k, nse * (2 if (wdt == np.float64 and not homo) else 1), homo,
                                                    C.ScatterPlan.default_shift(k, homo))[1])
        per_block = nse * (2 if (wdt == np.float64 and not homo) else 1) / (m * n_sl)
        if per_block >= C.PLAN_MIN_SEGMENT_NO_BINNED or C.BinnedScatter.applicable(csr.data, k):
            assert route != 'direct', (wdt, homo, per_block)
            csr.prepare()
            assert csr.buffers['scatter_plan'] is not None
    tol = {np.float32: 1e-5, np.float64: 1e-10, np.float16: 2e-2}[wdt]
    sdt = [np.bool_, np.uint8, np.int8, np.int32, np.float32, np.float64, np.float16][int(rng.integers(0, 7))]

    def spikes(n, fire, shape_tail=()):
        a = rng.random((n,) + shape_tail) < fire
        if np.issubdtype(sdt, np.floating):
            return np.where(a, rng.uniform(0.5, 2.0, a.shape), rng.uniform(-1.0, 0.0, a.shape)).astype(sdt), a
        return a.astype(sdt), a

    def out_np(x):
        return x.double().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x, np.float64)

    w64 = w.astype(np.float64)
    for fire in (0.1, 0.8):
        sv, sa = spikes(m, fire)
        ref = oracle.binary_csrmv(w64, idx, ptr, sa, (m, k), True)
        got = be.BinaryArray(conv(sv)) @ csr
        assert isinstance(got, torch.Tensor) == on_device
        np.testing.assert_allclose(out_np(got), ref, rtol=tol, atol=tol * max(1.0, float(np.abs(ref).max())))
        sv, sa = spikes(k, fire)
        ref = oracle.binary_csrmv(w64, idx, ptr, sa, (m, k), False)
        np.testing.assert_allclose(out_np(csr @ be.BinaryArray(conv(sv))), ref, rtol=tol, atol=tol * max(1.0, float(np.abs(ref).max())))
    nb = int(rng.integers(1, 6))
    Sv, Sa = spikes(nb, 0.3, (m,))
    ref = np.stack([oracle.binary_csrmv(w64, idx, ptr, Sa[b], (m, k), True) for b in range(nb)])
    np.testing.assert_allclose(out_np(be.BinaryArray(conv(Sv)) @ csr), ref, rtol=tol, atol=tol * max(1.0, float(np.abs(ref).max())))


@pytest.mark.parametrize('layout', ['d8', 'u16', 'homo', 'homo_u16'])
@pytest.mark.parametrize('m', [1, 700, 4096, 4097, 20000])
def test_single_launch_planned_step(be, oracle, layout, m):
    """Small matrices (one output slice, one part) run compaction + accumulate + output conversion in one single-workgroup
    launch (k_plan_single): spike vectors shorter, equal and longer than the 4096-spike compaction pass, all layouts, bool
    and float spikes, f32 and f16 weights, no / all / some rows active."""
    from brainevent_amd._csr import ScatterPlan
    rng = np.random.default_rng(m)
    k = 3000
    lens = rng.integers(0, 12, m)
    if m >= 700:
        lens[5] = 900                                     # one long row (several 64-lane chunks of a block)
    w, idx, ptr = rand_csr(rng, m, k, lens, homo=layout.startswith('homo'))
    for wdt in (np.float32, np.float16):
        ww = w.astype(wdt)
        plan = ScatterPlan.build(ww, idx, torch.tensor(ptr), shape=(m, k), layout={'homo': 'h8', 'homo_u16': 'u16'}.get(layout, layout))
        assert plan.n_slices == 1 and plan.default_parts() == 1
        assert plan.layout == {'d8': 1, 'u16': 0, 'homo': 2, 'homo_u16': 0}[layout]
        tol = 1e-5 if wdt == np.float32 else 2e-2
        for v in (np.zeros(m, bool), np.ones(m, bool), rng.random(m) < 0.1, np.where(rng.random(m) < 0.5, 1.5, -2.0).astype(np.float32)):
            ref = oracle.binary_csrmv(ww.astype(np.float32), idx, ptr, v, (m, k), True)
            got = be.binary_csrmv(ww, idx, ptr, v, shape=(m, k), transpose=True, workspace=plan)
            np.testing.assert_allclose(np.asarray(got, np.float32), ref, rtol=tol, atol=tol * max(1.0, float(np.abs(ref).max())))
            # same numbers as the three-kernel path (two parts force it): integer sums do not depend on the split
            got3 = torch.empty(k, dtype=torch.float32 if wdt == np.float32 else torch.float16, device='cuda')
            from brainevent_amd._csr import _plan_call
            from brainevent_amd import _array as A
            sp, sd = A.spikes_to_device(v)
            _plan_call(plan, A.to_device(ww), sp, sd, got3.reshape(1, -1), parts=2)
            np.testing.assert_array_equal(got3.float().cpu().numpy(), np.asarray(got, np.float32))


@pytest.mark.parametrize('k', [1_300_000, 5_000_000])
def test_gather_with_more_columns_than_lds_holds_bits_for(be, oracle, k):
    """Gather over more than 1.2M input columns: LDS holds a coarse bitmap (one bit per 2 / 8 columns) and only entries whose
    group fired test the exact bitmap in global memory.  Short, medium and long rows (every kernel tier), sparse and dense
    firing, a batch, bit-packed events."""
    rng = np.random.default_rng(k)
    for m, lens in ((3000, rng.integers(0, 30, 3000)), (800, rng.integers(0, 400, 800)), (60, rng.integers(500, 3000, 60))):
        lens[::11] = 0
        for homo in (False, True):
            w, idx, ptr = rand_csr(rng, m, k, lens, homo=homo)
            idx[: min(idx.size, 50)] = k - 1 - np.arange(min(idx.size, 50))          # the last columns: the tail of the bitmaps
            for fire in (0.0, 0.01, 0.5):
                v = rng.random(k) < fire
                got = be.binary_csrmv(w, idx, ptr, v, shape=(m, k), transpose=False)
                np.testing.assert_allclose(got, oracle.binary_csrmv(w, idx, ptr, v, (m, k), False), rtol=1e-5, atol=1e-5)
            csr = be.CSR((w, idx, ptr), shape=(m, k))
            got = csr @ be.BitPackedBinary(v)
            got = got.cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
            np.testing.assert_allclose(got, oracle.binary_csrmv(w, idx, ptr, v, (m, k), False), rtol=1e-5, atol=1e-5)
            B = rng.random((k, 3)) < 0.02
            gotm = be.binary_csrmm(w, idx, ptr, B, shape=(m, k), transpose=False)
            for c in range(3):
                np.testing.assert_allclose(np.asarray(gotm)[:, c], oracle.binary_csrmv(w, idx, ptr, B[:, c], (m, k), False),
                                           rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('avg', [30, 80, 150])
def test_batched_gather_fused_over_the_batch_with_lanes_per_row(be, oracle, avg):
    """binary_csrmm, transpose=False, rows of tens to ~200 entries and enough columns: one pass over the matrix fused over the
    batch with 8 / 16 / 32 lanes per row (k_csrmm_nt_fused_vec).  Rows from empty to several passes long (the wave finishes
    those before the group's fold), more than 32 columns (two passes), counted and weighted entries, float spikes."""
    rng = np.random.default_rng(avg)
    m, k = 2500, 40000
    lens = rng.integers(0, 2 * avg, m)
    lens[rng.random(m) < 0.03] = rng.integers(4 * avg, 3000, int((rng.random(m) < 0.03).sum()) or 1)[0]
    lens[::13] = 0
    for homo in (False, True):
        w, idx, ptr = rand_csr(rng, m, k, lens, homo=homo)
        for nb in (8, 32, 40):
            if (lens.mean() * nb) < 768:
                continue
            B = rng.random((k, nb)) < 0.03
            got = np.asarray(be.binary_csrmm(w, idx, ptr, B, shape=(m, k), transpose=False))
            assert got.shape == (m, nb)
            for c in range(nb):
                np.testing.assert_allclose(got[:, c], oracle.binary_csrmv(w, idx, ptr, B[:, c], (m, k), False), rtol=1e-5, atol=1e-5)
            gotf = np.asarray(be.binary_csrmm(w, idx, ptr, np.where(B, 1.5, -2.0).astype(np.float32), shape=(m, k), transpose=False))
            np.testing.assert_array_equal(gotf, got)
