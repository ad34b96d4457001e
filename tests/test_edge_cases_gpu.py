"""GPU edge cases across the operators: empty operands, all-zero / all-one spikes, NaN and negative float spikes,
integer spike dtypes, large batches, torch-tensor operands on the device, dtype preservation."""
import numpy as np
import pytest
import torch

from test_csr_gpu import rand_csr, spikes_of

pytestmark = pytest.mark.gpu


def test_float_spikes_threshold_semantics(be, oracle):
    # active iff > 0: zeros, negatives and NaN are inactive (NaN > 0 is False), +inf is active
    rng = np.random.default_rng(0)
    m, k = 64, 80
    w, idx, ptr = rand_csr(rng, m, k, rng.integers(1, 10, m))
    v = rng.normal(0, 1, m).astype(np.float32)
    v[0], v[1], v[2], v[3] = np.nan, np.inf, -np.inf, 0.0
    got = be.binary_csrmv(w, idx, ptr, v, shape=(m, k), transpose=True)
    act = np.nan_to_num(v, nan=-1.0) > 0
    np.testing.assert_allclose(got, oracle.binary_csrmv(w, idx, ptr, act, (m, k), True), rtol=1e-5, atol=1e-5)
    W = rng.normal(0, 1, (m, k)).astype(np.float32)
    np.testing.assert_allclose(be.binary_densemv(W, v, transpose=True), oracle.binary_densemv(W.astype(np.float64), act, True), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize('dtype', [np.int8, np.uint8, np.int32, np.int64, np.float64, np.float16])
def test_spike_dtypes_are_normalised(be, oracle, dtype):
    rng = np.random.default_rng(1)
    m, k = 50, 60
    w, idx, ptr = rand_csr(rng, m, k, rng.integers(0, 8, m))
    s = (rng.random(m) < 0.5)
    v = (s * 3).astype(dtype)
    np.testing.assert_allclose(be.binary_csrmv(w, idx, ptr, v, shape=(m, k), transpose=True),
                               oracle.binary_csrmv(w, idx, ptr, s, (m, k), True), rtol=1e-5, atol=1e-5)
    W = rng.normal(0, 1, (m, k)).astype(np.float32)
    np.testing.assert_allclose(be.BinaryArray(v) @ W, oracle.binary_densemv(W.astype(np.float64), s, True), rtol=1e-5, atol=1e-4)


def test_no_spikes_and_all_spikes(be, oracle):
    rng = np.random.default_rng(2)
    m, k = 300, 70000
    w, idx, ptr = rand_csr(rng, m, k, np.full(m, 300))
    import brainevent_amd._csr as C
    old = C.PLAN_MIN_NNZ
    C.PLAN_MIN_NNZ = 1
    try:
        csr = be.CSR((w, idx, ptr), shape=(m, k))
        z = be.BinaryArray(np.zeros(m, bool)) @ csr
        assert z.shape == (k,) and not z.any()
        o = be.BinaryArray(np.ones(m, bool)) @ csr
        np.testing.assert_allclose(o, oracle.binary_csrmv(w.astype(np.float64), idx, ptr, np.ones(m, bool), (m, k), True), rtol=1e-5, atol=1e-5)
        z2 = be.BinaryArray(np.zeros(m, bool)) @ csr          # the re-armed counter survives an empty call
        assert not z2.any()
    finally:
        C.PLAN_MIN_NNZ = old
    assert not be.binary_jitsmv(np.float32(1.0), 0.1, np.zeros(k, bool), 3, shape=(m, k), transpose=False, corder=True).any()
    assert not be.binary_jitsmv(np.float32(1.0), 0.1, np.zeros(m, bool), 3, shape=(m, k), transpose=True, corder=False).any()


def test_empty_shapes(be):
    w1 = np.ones(1, np.float32)
    e_i, e_p = np.zeros(0, np.int32), np.zeros(1, np.int32)
    assert be.binary_csrmv(w1, e_i, e_p, np.zeros(0, bool), shape=(0, 5), transpose=True).shape == (5,)
    assert be.binary_csrmv(w1, e_i, np.zeros(4, np.int32), np.zeros(0, bool), shape=(3, 0), transpose=False).shape == (3,)
    assert be.binary_densemv(np.zeros((0, 4), np.float32), np.zeros(0, bool), transpose=True).tolist() == [0, 0, 0, 0]
    assert be.binary_densemm(np.zeros((3, 4), np.float32), np.zeros((4, 0), bool), transpose=False).shape == (3, 0)
    assert be.binary_fcnmv(w1, np.zeros((0, 3), np.int32), np.zeros(0, bool), shape=(0, 7), transpose=True).shape == (7,)
    assert be.binary_jitsmv(np.float32(1.0), 0.5, np.zeros(0, bool), 1, shape=(4, 0), transpose=False, corder=True).tolist() == [0, 0, 0, 0]


def test_large_batch_and_torch_operands(be, oracle):
    rng = np.random.default_rng(3)
    m, k, n = 90, 120, 70          # n > 64: several batch passes for dense (32 per pass)
    w, idx, ptr = rand_csr(rng, m, k, rng.integers(0, 20, m))
    B = rng.random((m, n)) < 0.3
    wt, it, pt, Bt = (torch.tensor(x, device='cuda') for x in (w, idx, ptr, B))
    got = be.binary_csrmm(wt, it, pt, Bt, shape=(m, k), transpose=True)
    assert isinstance(got, torch.Tensor) and got.is_cuda and tuple(got.shape) == (k, n)
    np.testing.assert_allclose(got.cpu().numpy(), oracle.binary_csrmm(w.astype(np.float64), idx, ptr, B, (m, k), True), rtol=1e-5, atol=1e-5)
    W = rng.normal(0, 1, (m, k)).astype(np.float32)
    gd = be.binary_densemm(torch.tensor(W, device='cuda'), Bt, transpose=True)
    np.testing.assert_allclose(gd.cpu().numpy(), oracle.binary_densemm(W.astype(np.float64), B, True), rtol=1e-5, atol=1e-4)
    Wh = torch.tensor(W, device='cuda').half()
    gh = be.BinaryArray(Bt.T.contiguous()) @ Wh          # [n, m] @ [m, k]: MFMA path (70 batch rows = 3 passes)
    assert gh.dtype == torch.float16 and tuple(gh.shape) == (n, k)
    np.testing.assert_allclose(gh.float().cpu().numpy(), oracle.binary_densemm(Wh.float().cpu().numpy().astype(np.float64), B, True).T,
                               rtol=2e-3, atol=2e-2)


def test_backend_keyword_contract(be):
    w, idx, ptr = np.ones(2, np.float32), np.array([0, 1], np.int32), np.array([0, 2], np.int32)
    v = np.array([True])
    assert be.binary_csrmv(w, idx, ptr, v, shape=(1, 2), transpose=True, backend='hip').tolist() == [1.0, 1.0]
    with pytest.raises(be.KernelFallbackExhaustedError):
        be.binary_csrmv(w, idx, ptr, v, shape=(1, 2), transpose=True, backend='numba')
    with pytest.raises(AssertionError):
        be.binary_csrmv(w, idx, ptr, np.array([True, False]), shape=(1, 2), transpose=True)       # shape mismatch
    with pytest.raises(AssertionError):
        be.binary_csrmv(np.array([1, 2]), idx, ptr, v, shape=(1, 2), transpose=True)               # integer weights
    with pytest.raises(ValueError):
        be.CSR((w, np.array([0, 5], np.int32), ptr), shape=(1, 2))                                 # column out of range
    with pytest.raises(ValueError):
        be.CSR((w, idx, np.array([0, 3], np.int32)), shape=(1, 2))                                 # indptr[-1] != nse


def test_operator_benchmark_harness():
    """``OpKernel.benchmark`` (reference ``XLACustomKernel.benchmark``, ``_op/main.py:1237``): per-call timing records for
    every registered backend over the operator's own data generator."""
    import brainevent_amd as be
    recs = be.binary_csrmv_p.benchmark(platform='gpu', n_warmup=1, n_runs=2, n_batch_per_run=2)
    assert len(recs) == 2 * 2 * 2 * 2 * 2
    assert all(r['success'] and r['backend'] == 'hip' and r['mean_ms'] > 0 and r['min_ms'] <= r['mean_ms'] for r in recs)
    assert recs[0]['name'].startswith('1000x1000,p=1%,NT,homo')
    bad = be.binary_csrmv_p.benchmark(platform='gpu', n_warmup=0, n_runs=1, backends=['nope'])
    assert all((not r['success']) and 'nope' in r['error'] for r in bad)


def test_index_structure_conversions_match_scipy():
    """csr_to_coo / coo_to_csc / coo2csr / csr_to_csc / csc_to_csr (reference ``brainevent/_misc.py:871-1085``, ``:1516-1700``)
    against scipy, including the reference's docstring example; the permutation reorders CSR data into CSC order."""
    import scipy.sparse as sp
    import brainevent_amd as be
    ptr, idx, perm = be.csr_to_csc_index(np.array([0, 2, 3, 5]), np.array([0, 2, 1, 0, 3]), shape=(3, 4))
    np.testing.assert_array_equal(ptr, [0, 2, 3, 4, 5])
    np.testing.assert_array_equal(idx, [0, 2, 1, 0, 2])
    np.testing.assert_array_equal(perm, [0, 3, 2, 1, 4])
    rng = np.random.default_rng(0)
    for m, k in ((1, 1), (17, 5), (40, 300), (300, 40)):
        lens = rng.integers(0, 12, m)
        indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        indices = np.concatenate([np.sort(rng.choice(k, min(l, k), replace=False)) for l in lens] + [np.zeros(0, int)]).astype(np.int32)
        indptr = np.concatenate([[0], np.cumsum([min(l, k) for l in lens])]).astype(np.int32)
        data = rng.random(indices.size).astype(np.float32)
        ref = sp.csr_matrix((data, indices, indptr), shape=(m, k)).tocsc()
        cptr, cidx, perm = be.csr_to_csc_index(indptr, indices, shape=(m, k))
        assert cidx.dtype == np.int32
        np.testing.assert_array_equal(cptr, ref.indptr)
        np.testing.assert_array_equal(cidx, ref.indices)
        np.testing.assert_array_equal(data[perm], ref.data)
        assert be.csr_to_csc_index(indptr, indices, shape=(m, k), include_perm=False)[2] is None
        rows, cols = be.csr_to_coo_index(indptr, indices)
        np.testing.assert_array_equal(rows, np.repeat(np.arange(m), np.diff(indptr)))
        p2, i2, _ = be.coo_to_csc_index(rows, cols, shape=(m, k))
        np.testing.assert_array_equal(p2, ref.indptr); np.testing.assert_array_equal(i2, ref.indices)
        shuffle = rng.permutation(indices.size)
        p3, i3, pos = be.coo2csr(rows[shuffle], cols[shuffle], shape=(m, k))
        np.testing.assert_array_equal(p3, indptr)
        np.testing.assert_array_equal(np.sort(i3), np.sort(indices))
        np.testing.assert_array_equal(cols[shuffle][pos], i3)
        rptr, ridx, rperm = be.csc_to_csr_index(cptr, cidx, shape=(m, k))          # round trip
        np.testing.assert_array_equal(rptr, indptr); np.testing.assert_array_equal(ridx, indices)
        np.testing.assert_array_equal(ref.data[rperm], data)
    t = be.csr_to_csc_index(torch.tensor([0, 1, 2], device='cuda'), torch.tensor([1, 0], dtype=torch.int32, device='cuda'), shape=(2, 2))
    assert all(isinstance(x, torch.Tensor) and x.is_cuda for x in t)
    with pytest.raises(ValueError):
        be.csr_to_csc_index(np.array([0, 1]), np.array([0]), shape=(1, 1), method='bogus')


def test_fixed_conn_num_csc_helpers_match_scipy():
    """fixed_conn_num_csr_indptr / _csc_structure / _to_csc (reference ``_misc.py:1135-1320``) against scipy's CSR -> CSC,
    and the CSC mirror drives the same products as the fixed-number container."""
    import scipy.sparse as sp
    import brainevent_amd as be
    rng = np.random.default_rng(8)
    n_pre, n_post, K = 60, 45, 7
    indices = rng.integers(0, n_post, (n_pre, K)).astype(np.int32)
    w = rng.uniform(0.1, 1.0, (n_pre, K)).astype(np.float32)
    ptr = be.fixed_conn_num_csr_indptr(indices)
    assert ptr.dtype == np.int32 and np.array_equal(ptr, np.arange(n_pre + 1) * K)
    cptr, crows, perm = be.fixed_conn_num_csc_structure(indices, shape=(n_pre, n_post))
    assert crows.dtype == np.int32 and cptr[0] == 0 and cptr[-1] == n_pre * K
    # every column lists its pre neurons in pre order (stable), and perm reorders the flat weights
    flat_rows = np.repeat(np.arange(n_pre), K)
    np.testing.assert_array_equal(flat_rows[perm], crows)
    np.testing.assert_array_equal(indices.reshape(-1)[perm], np.repeat(np.arange(n_post), np.diff(cptr)))
    assert all(np.all(np.diff(crows[cptr[j]:cptr[j + 1]]) >= 0) for j in range(n_post))
    data, rows2, ptr2 = be.fixed_conn_num_to_csc(w, indices, shape=(n_pre, n_post))
    np.testing.assert_array_equal(rows2, crows); np.testing.assert_array_equal(ptr2, cptr)
    dense = np.zeros((n_pre, n_post), np.float64)
    np.add.at(dense, (flat_rows, indices.reshape(-1)), w.reshape(-1))
    np.testing.assert_allclose(sp.csc_matrix((data, rows2, ptr2), shape=(n_pre, n_post)).toarray(), dense, rtol=1e-6)
    h_data, _, _ = be.fixed_conn_num_to_csc(np.float32(2.0), indices, shape=(n_pre, n_post))      # homogeneous stays size 1
    assert h_data.shape == (1,) and h_data[0] == 2.0
    v = rng.random(n_pre) < 0.4
    csc = be.CSC((data, rows2, ptr2), shape=(n_pre, n_post))
    fcn = be.FixedNumPerPre((w, indices), shape=(n_pre, n_post))
    np.testing.assert_allclose(be.BinaryArray(v) @ csc, be.BinaryArray(v) @ fcn, rtol=1e-5, atol=1e-6)
    t = be.fixed_conn_num_csc_structure(torch.from_numpy(indices).cuda(), shape=(n_pre, n_post))
    assert all(isinstance(x, torch.Tensor) and x.is_cuda for x in t)
    with pytest.raises(AssertionError):
        be.fixed_conn_num_csc_structure(indices, shape=(n_pre + 1, n_post))


def test_indexed_products_equal_the_reindexed_matrix():
    """binary_csrmv/mm_indexed (reference ``_csr/binary_indexed.py``): the CSC view of a CSR matrix with the weights left
    in CSR order and reached through the permutation — same numbers as the products on the CSR matrix itself."""
    import brainevent_amd as be
    from oracle import oracle_np as O
    rng = np.random.default_rng(5)
    m, k = 60, 45
    lens = rng.integers(0, 9, m)
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    idx = rng.integers(0, k, ptr[-1]).astype(np.int32)
    for w in (rng.random(ptr[-1]).astype(np.float32), np.array([0.5], np.float32)):
        cptr, cidx, perm = be.csr_to_csc_index(ptr, idx, shape=(m, k))
        sm, sk = rng.random(m) < 0.4, rng.random(k) < 0.4
        # the CSC arrays are the CSR arrays of the transpose (k x m): scatter over it = gather over the original, and back
        np.testing.assert_allclose(be.binary_csrmv_indexed(w, cidx, cptr, perm, sk, shape=(k, m), transpose=True),
                                   O.binary_csrmv(w, idx, ptr, sk, (m, k), False), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(be.binary_csrmv_indexed(w, cidx, cptr, perm, sm, shape=(k, m), transpose=False),
                                   O.binary_csrmv(w, idx, ptr, sm, (m, k), True), rtol=1e-5, atol=1e-5)
        B = rng.random((k, 4)) < 0.4
        np.testing.assert_allclose(be.binary_csrmm_indexed(w, cidx, cptr, perm, B, shape=(k, m), transpose=True),
                                   O.binary_csrmm(w, idx, ptr, B, (m, k), False), rtol=1e-5, atol=1e-5)
        # the operator objects (device operands, backend resolution as for every other operator)
        dw, di, dp, dperm = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (w, cidx, cptr, perm))
        got = be.binary_csrmv_indexed_p(dw, di, dp, dperm, torch.from_numpy(sk).cuda(), shape=(k, m), transpose=True)
        np.testing.assert_allclose(got.cpu().numpy(), O.binary_csrmv(w, idx, ptr, sk, (m, k), False), rtol=1e-5, atol=1e-5)
        got = be.binary_csrmm_indexed_p(dw, di, dp, dperm, torch.from_numpy(B).cuda(), shape=(k, m), transpose=True)
        np.testing.assert_allclose(got.cpu().numpy(), O.binary_csrmm(w, idx, ptr, B, (m, k), False), rtol=1e-5, atol=1e-5)
    assert 'indexed' in be.binary_csrmv_indexed_p.tags and be.binary_csrmv_indexed_p.available_backends('gpu') == ['hip']


def test_tocsc_tocsr_keep_the_matrix():
    import brainevent_amd as be
    rng = np.random.default_rng(6)
    dense = (rng.random((23, 31)) < 0.2) * rng.standard_normal((23, 31)).astype(np.float32)
    csr = be.CSR.fromdense(dense)
    csc = csr.tocsc()
    assert isinstance(csc, be.CSC) and csc.shape == csr.shape and csr.tocsr() is csr and csc.tocsc() is csc
    np.testing.assert_array_equal(csc.todense(), dense)
    back = csc.tocsr()
    assert isinstance(back, be.CSR)
    np.testing.assert_array_equal(back.todense(), dense)
    np.testing.assert_array_equal(back.indices.cpu().numpy(), csr.indices.cpu().numpy())
    s = rng.random(23) < 0.5
    np.testing.assert_allclose(be.BinaryArray(s) @ csc, s.astype(np.float32) @ dense, rtol=1e-5, atol=1e-5)


def test_odd_operand_layouts(be, oracle):
    """Strided spike vectors, views with storage offsets (4-byte aligned only), int64 indices, non-contiguous (data, indices)
    of a FixedNumPerPre, transposed views of a dense matrix: converted or consumed as they are, never misread."""
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(0)
    m, k, K = 300, 500, 12
    idx = rng.integers(0, k, (m, K)).astype(np.int32)
    w = rng.uniform(0.1, 1, (m, K)).astype(np.float32)
    ptr = (np.arange(m + 1) * K).astype(np.int32)
    v, vk = rng.random(m) < 0.3, rng.random(k) < 0.3
    ref_t = oracle.binary_csrmv(w.reshape(-1), idx.reshape(-1), ptr, v, (m, k), True)
    ref_n = oracle.binary_csrmv(w.reshape(-1), idx.reshape(-1), ptr, vk, (m, k), False)
    wt, it, pt = torch.tensor(w.reshape(-1), device=dev), torch.tensor(idx.reshape(-1), device=dev), torch.tensor(ptr, device=dev)
    vt, vkt = torch.tensor(v, device=dev), torch.tensor(vk, device=dev)

    def same(got, ref):
        got = got.cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
        np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)

    big = torch.zeros(2 * m, dtype=torch.bool, device=dev); big[::2] = vt
    same(be.binary_csrmv(wt, it, pt, big[::2], shape=(m, k), transpose=True), ref_t)
    bigk = torch.zeros(2 * k, dtype=torch.float32, device=dev); bigk[1::2] = vkt.float()
    same(be.binary_csrmv(wt, it, pt, bigk[1::2], shape=(m, k), transpose=False), ref_n)
    pad_w = torch.cat([torch.full((7,), 9e9, device=dev), wt])
    pad_i = torch.cat([torch.full((5,), k - 1, dtype=torch.int32, device=dev), it])
    same(be.binary_csrmv(pad_w[7:], pad_i[5:], pt, vt, shape=(m, k), transpose=True), ref_t)
    same(be.binary_csrmv(pad_w[7:], pad_i[5:], pt, vkt, shape=(m, k), transpose=False), ref_n)
    csr = be.CSR((w.reshape(-1), idx.reshape(-1), ptr), shape=(m, k))
    same(be.BinaryArray(big[::2]) @ csr, ref_t)
    same(csr @ be.BinaryArray(vkt), ref_n)
    idx_t = torch.tensor(np.ascontiguousarray(idx.T), device=dev).T          # shape (m, K), strides (1, m)
    w_t = torch.tensor(np.ascontiguousarray(w.T), device=dev).T
    same(be.BinaryArray(v) @ be.FixedNumPerPre((w_t, idx_t), shape=(m, k)), ref_t)
    W = rng.normal(0, 1, (k, m)).astype(np.float32)
    Wv = torch.tensor(W, device=dev).T                                       # (m, k) view of a (k, m) array
    same(be.BinaryArray(v) @ Wv, oracle.binary_densemv(W.T.astype(np.float64), v, True))
    same(Wv @ be.BinaryArray(vk), oracle.binary_densemv(W.T.astype(np.float64), vk, False))
    out = be.binary_csrmm(wt, it, pt, torch.zeros((m, 0), dtype=torch.bool, device=dev), shape=(m, k), transpose=True)
    assert tuple(out.shape) == (k, 0)


@pytest.mark.parametrize('route', ['plan', 'binned', 'gather'])
def test_offset_views_through_the_fast_routes(be, oracle, route):
    """Matrix arrays that start 4 bytes (not 16) into their storage, behind poison: the plan build, the binned passes and the
    lanes-per-row gather read them with vector loads where alignment allows and never touch what lies in front."""
    from brainevent_amd._csr import ScatterPlan, BinnedScatter
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(5)
    m, k = 3000, 200_000
    lens = rng.integers(20, 60, m)
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    nnz = int(ptr[-1])
    idx = rng.integers(0, k, nnz).astype(np.int32)
    w = rng.uniform(0.1, 1, nnz).astype(np.float32)
    pad_w = torch.cat([torch.full((1,), 9e9, device=dev), torch.tensor(w, device=dev), torch.full((64,), 9e9, device=dev)])
    pad_i = torch.cat([torch.full((3,), k - 1, dtype=torch.int32, device=dev), torch.tensor(idx, device=dev),
                       torch.full((64,), k - 1, dtype=torch.int32, device=dev)])
    wv, iv, pt = pad_w[1:1 + nnz], pad_i[3:3 + nnz], torch.tensor(ptr, device=dev)
    if route == 'gather':
        v = rng.random(k) < 0.2
        v[k - 1] = True
        got = be.binary_csrmv(wv, iv, pt, torch.tensor(v, device=dev), shape=(m, k), transpose=False)
        ref = oracle.binary_csrmv(w, idx, ptr, v, (m, k), False)
    else:
        v = rng.random(m) < 0.3
        ws = ScatterPlan.build(wv, iv, pt, shape=(m, k)) if route == 'plan' else BinnedScatter(wv, m, k, nnz, max_active_fraction=0.5, indices=iv)
        got = be.binary_csrmv(wv, iv, pt, torch.tensor(v, device=dev), shape=(m, k), transpose=True, workspace=ws)
        ref = oracle.binary_csrmv(w, idx, ptr, v, (m, k), True)
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
