"""Bit-packed / compacted event encodings against the oracle, and as operands of the
scatter / gather kernels.  Construction cases mirror the reference's own
``brainevent/_event/compact_binary_test.py`` (shapes, dtypes, counts, sorted active ids)."""
import numpy as np
import pytest
import torch

from oracle import oracle_np as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def be():
    import brainevent_amd as be
    return be


@pytest.mark.parametrize('shape,axis', [((100,), 0), ((1,), 0), ((32,), 0), ((33,), 0), ((64,), 0), ((7, 100), 1),
                                        ((7, 100), 0), ((50, 8), 1), ((3, 5, 70), 2), ((3, 5, 70), 1), ((0,), 0)])
def test_bitpack_matches_oracle(be, shape, axis):
    rng = np.random.default_rng(sum(shape) + axis)
    x = rng.random(shape) > 0.6
    got = be.bitpack(x, axis)
    assert got.dtype == np.uint32
    np.testing.assert_array_equal(got, O.bitpack(x, axis))
    got_t = be.bitpack(torch.from_numpy(x).cuda(), axis)
    assert isinstance(got_t, torch.Tensor) and got_t.dtype == torch.int32
    np.testing.assert_array_equal(got_t.cpu().numpy().view(np.uint32), O.bitpack(x, axis))


def test_bitpack_many_short_rows_and_nonbool(be):
    rng = np.random.default_rng(5)
    x = (rng.random((70000, 40)) > 0.5)
    np.testing.assert_array_equal(be.bitpack(x, 1), O.bitpack(x, 1))
    f = np.array([0.0, 1.0, -2.0, 0.0, 3.5], np.float32)          # encoders: non-zero is True (negatives too)
    np.testing.assert_array_equal(be.bitpack(f, 0), O.bitpack(f, 0))
    assert int(be.bitpack(f, 0)[0]) == 0b10110


class TestCompactBinaryConstruction:
    def test_1d_basic(self, be):
        x = np.random.RandomState(0).rand(100) > 0.7
        cb = be.CompactBinary.from_array(x)
        assert cb.packed.dtype == np.uint32 and cb.packed.shape == ((100 + 31) // 32,)
        assert cb.active_ids.shape == (100,) and cb.active_ids.dtype == np.int32
        ids, n = O.compact_1d(x)
        assert int(cb.n_active[0]) == n == int(x.sum())
        np.testing.assert_array_equal(np.sort(cb.active_ids[:n]), ids)
        np.testing.assert_array_equal(cb.packed, O.bitpack(x, 0))
        np.testing.assert_array_equal(cb.value, x)
        assert cb.shape == (100,) and cb.ndim == 1 and cb.size == 100 and cb.n_orig == 100 and cb.batch_size is None

    def test_float_input(self, be):
        cb = be.CompactBinary.from_array(np.array([0.0, 1.0, 0.0, 2.0, 0.0], np.float32))
        assert int(cb.n_active[0]) == 2

    def test_light_skips_compaction(self, be):
        x = np.array([False, True, True, False])
        cb = be.CompactBinary.from_array_light(x)
        assert cb.packed.shape == (1,)
        np.testing.assert_array_equal(cb.active_ids, np.zeros(4, np.int32))
        np.testing.assert_array_equal(cb.n_active, np.zeros(1, np.int32))

    def test_compact_only_vector(self, be):
        x = np.array([0.0, 1.0, 0.0, 1.0, 0.0, 1.0], np.float32)
        cb = be.CompactBinary.compacy_only_vector(x)
        assert cb.packed.shape == (0,)
        assert int(cb.n_active[0]) == 3
        np.testing.assert_array_equal(np.sort(cb.active_ids[:3]), [1, 3, 5])
        with pytest.raises(ValueError, match="only supports 1D arrays"):
            be.CompactBinary.compacy_only_vector(np.zeros((2, 3), bool))

    def test_2d(self, be):
        x = np.random.RandomState(1).rand(50, 8) > 0.7
        cb = be.CompactBinary.from_array(x)
        assert cb.packed.dtype == np.uint32 and cb.packed.shape == (50, 1)
        ids, n = O.compact_2d(x)
        assert int(cb.n_active[0]) == n
        np.testing.assert_array_equal(np.sort(cb.active_ids[:n]), ids)
        np.testing.assert_array_equal(cb.packed, O.bitpack(x, 1))
        x2 = np.random.RandomState(2).rand(20, 100) > 0.8
        assert be.CompactBinary.from_array(x2).packed.shape == (20, 4)
        assert be.CompactBinary.from_array(x2).shape == (20, 100)

    def test_zeros_ones_and_errors(self, be):
        assert int(be.CompactBinary.from_array(np.zeros(64, bool)).n_active[0]) == 0
        assert int(be.CompactBinary.from_array(np.zeros((32, 8), bool)).n_active[0]) == 0
        cb = be.CompactBinary.from_array(np.ones(64, bool))
        np.testing.assert_array_equal(np.sort(cb.active_ids[:64]), np.arange(64))
        cb = be.CompactBinary.from_array(np.ones((32, 8), bool))
        np.testing.assert_array_equal(np.sort(cb.active_ids[:32]), np.arange(32))
        with pytest.raises(ValueError):
            be.CompactBinary.from_array(np.zeros((2, 2, 2), bool))
        with pytest.raises(ValueError):
            be.CompactBinary.from_array(np.zeros(4, bool), bit_width=16)

    def test_from_packed_roundtrip_and_device(self, be):
        x = torch.tensor([False, True, False, True], device='cuda')
        cb = be.CompactBinary.from_array(x)
        assert isinstance(cb.packed, torch.Tensor) and cb.packed.is_cuda
        rb = be.CompactBinary.from_packed(cb.packed, cb.active_ids, cb.n_active, cb.value, n_orig=cb.n_orig,
                                          batch_size=cb.batch_size, bit_width=cb.bit_width)
        assert rb.packed is cb.packed and rb.n_orig == 4 and int(rb.n_active[0]) == 2


def test_bitpacked_binary_container(be):
    x = np.random.default_rng(3).random((6, 70)) > 0.5
    bp = be.BinaryArray(x).bitpack()
    assert isinstance(bp, be.BitPackedBinary)
    assert bp.shape == (6, 70) and bp.ndim == 2 and bp.original_shape == (6, 70)
    assert len(bp.packed) == 2
    np.testing.assert_array_equal(bp.packed[0], O.bitpack(x, 0))
    np.testing.assert_array_equal(bp.packed[1], O.bitpack(x, 1))
    t = bp.T
    assert t.shape == (70, 6)
    np.testing.assert_array_equal(t.packed[0], O.bitpack(x.T, 0))
    np.testing.assert_array_equal(t.packed[1], O.bitpack(x.T, 1))
    np.testing.assert_array_equal(t.value, x.T)
    W = np.random.default_rng(4).standard_normal((70, 9)).astype(np.float32)
    np.testing.assert_allclose(bp @ W, O.binary_densemm(W, x.T, transpose=True).T, rtol=1e-5, atol=1e-5)
    with pytest.raises(be.MathError):
        be.BitPackedBinary(np.zeros((2, 2, 2), bool)) @ W[:2]


def _rand_csr(rng, m, k, max_len, homo):
    lens = rng.integers(0, max_len, m)
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    idx = rng.integers(0, k, ptr[-1]).astype(np.int32)
    w = np.array([1.5], np.float32) if homo else rng.random(ptr[-1]).astype(np.float32)
    return w, idx, ptr


@pytest.mark.parametrize('homo', [True, False])
@pytest.mark.parametrize('m,k', [(257, 300), (4000, 70001), (31, 5)])
def test_packed_operands_on_csr(be, homo, m, k):
    """BitPackedBinary / CompactBinary as the event operand: all four operator forms, direct and planned routes."""
    rng = np.random.default_rng(m + k)
    w, idx, ptr = _rand_csr(rng, m, k, 40, homo)
    csr = be.CSR((w, idx, ptr), shape=(m, k))
    s_m = rng.random(m) < 0.2
    s_k = rng.random(k) < 0.2
    for make in (lambda s: be.BinaryArray(s).bitpack(), be.CompactBinary.from_array):
        ref_t = O.binary_csrmv(w, idx, ptr, s_m, shape=(m, k), transpose=True)
        ref_nt = O.binary_csrmv(w, idx, ptr, s_k, shape=(m, k), transpose=False)
        np.testing.assert_allclose(make(s_m) @ csr, ref_t, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(csr @ make(s_k), ref_nt, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(csr.T @ make(s_m), ref_t, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(make(s_k) @ csr.T, ref_nt, rtol=1e-5, atol=1e-5)
    # planned route + packed-only vector (what the spike exchange delivers)
    from brainevent_amd._csr import ScatterPlan
    csr.buffers['scatter_plan'] = ScatterPlan.build(csr.data, csr.indices, csr.indptr, shape=(m, k), slice_shift=9)
    words = be.bitpack(torch.from_numpy(s_m).cuda(), 0)
    got = be.BitPackedBinary.from_packed(words, m) @ csr
    assert isinstance(got, torch.Tensor)
    np.testing.assert_allclose(got.cpu().numpy(), ref_t, rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(be.BitPackedBinary.from_packed(words, m).value.cpu().numpy(), s_m)


def test_packed_operands_bitwise_equal_to_plain(be):
    """Same kernels after the compaction => bitwise identical results on the integer-accumulating routes."""
    rng = np.random.default_rng(77)
    m, k = 3000, 50000
    w, idx, ptr = _rand_csr(rng, m, k, 200, homo=False)
    from brainevent_amd._csr import ScatterPlan, BinnedScatter
    s = torch.from_numpy(rng.random(m) < 0.1).cuda()
    for ws in ('plan', 'binned'):
        csr = be.CSR((torch.from_numpy(w).cuda(), torch.from_numpy(idx).cuda(), torch.from_numpy(ptr).cuda()), shape=(m, k))
        csr.buffers['scatter_plan'] = (ScatterPlan.build(csr.data, csr.indices, csr.indptr, shape=(m, k)) if ws == 'plan'
                                       else BinnedScatter(csr.data, m, k, idx.size))
        a = be.BinaryArray(s) @ csr
        b = be.BinaryArray(s).bitpack() @ csr
        if ws == 'plan':      # fixed-order integer sums: bitwise reproducible whatever the order of the active list
            assert torch.equal(a, b), ws
        else:                 # the binned route merges its per-part sums with float atomics
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('homo', [True, False])
def test_packed_operands_on_fixed_num(be, homo):
    rng = np.random.default_rng(9)
    n_pre, n_post, nc = 500, 777, 12
    idx = rng.integers(0, n_post, (n_pre, nc)).astype(np.int32)
    w = np.array([0.5], np.float32) if homo else rng.random((n_pre, nc)).astype(np.float32)
    M = be.FixedNumPerPre((w, idx), shape=(n_pre, n_post))
    s_pre = rng.random(n_pre) < 0.3
    s_post = rng.random(n_post) < 0.3
    np.testing.assert_allclose(be.BinaryArray(s_pre).bitpack() @ M,
                               O.binary_fcnmv(w, idx, s_pre, shape=(n_pre, n_post), transpose=True), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(M @ be.CompactBinary.from_array(s_post),
                               O.binary_fcnmv(w, idx, s_post, shape=(n_pre, n_post), transpose=False), rtol=1e-5, atol=1e-5)


def test_float_payload_keeps_op_semantics(be):
    """Negative floats are set bits for the encoder (non-zero) but inactive for the product (> 0): the containers
    must not feed packed words to the kernels for float payloads."""
    w = np.array([1.0, 2.0, 3.0, 4.0], np.float32)
    idx = np.array([0, 2, 1, 2], np.int32)
    ptr = np.array([0, 2, 4], np.int32)
    csr = be.CSR((w, idx, ptr), shape=(2, 3))
    v = np.array([1.0, -1.0], np.float32)
    np.testing.assert_allclose(be.BinaryArray(v).bitpack() @ csr, [1, 0, 2])      # reference KAT _csr/binary_test.py:370-393
    np.testing.assert_allclose(be.CompactBinary.from_array(v) @ csr, [1, 0, 2])


def test_packed_on_jitc_and_dense(be):
    rng = np.random.default_rng(2)
    s = rng.random(64) < 0.3
    M = be.JITCScalarR((1.5, 0.2, 123), shape=(64, 48))
    ref = be.BinaryArray(s) @ M
    np.testing.assert_array_equal(be.BinaryArray(s).bitpack() @ M, ref)
    np.testing.assert_array_equal(be.CompactBinary.from_array(s) @ M, ref)
    W = rng.standard_normal((64, 5)).astype(np.float32)
    np.testing.assert_allclose(be.CompactBinary.from_array(s) @ W, O.binary_densemv(W, s, transpose=True), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('route', ['direct', 'plan', 'binned'])
@pytest.mark.parametrize('homo', [True, False])
def test_compacted_ids_as_scatter_operand(be, route, homo):
    """A CompactBinary built on the device hands its id list to the scatter kernels (BE_SPIKE_IDS): no second compaction."""
    from brainevent_amd import _array as A
    from brainevent_amd._event import event_operand
    from brainevent_amd._csr import ScatterPlan, BinnedScatter
    rng = np.random.default_rng(31)
    m, k = 5000, 40000
    w, idx, ptr = _rand_csr(rng, m, k, 120, homo)
    s = rng.random(m) < 0.15
    cb = be.CompactBinary.from_array(torch.from_numpy(s).cuda())
    assert isinstance(event_operand(cb, scatter=True), A.ActiveIds)
    assert isinstance(event_operand(cb, scatter=False), A.PackedSpikes)
    assert not isinstance(event_operand(be.CompactBinary.from_array(s), scatter=True), A.ActiveIds)   # host-built: not trusted
    csr = be.CSR((torch.from_numpy(w).cuda(), torch.from_numpy(idx).cuda(), torch.from_numpy(ptr).cuda()), shape=(m, k))
    csr.buffers['scatter_plan'] = {'direct': None,
                                   'plan': ScatterPlan.build(csr.data, csr.indices, csr.indptr, shape=(m, k)),
                                   'binned': BinnedScatter(csr.data, m, k, idx.size, indices=csr.indices)}[route]
    ref = O.binary_csrmv(w, idx, ptr, s, shape=(m, k), transpose=True)
    got = cb @ csr
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose((csr.T @ cb).cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
    if route == 'plan':
        assert torch.equal(got, be.BinaryArray(torch.from_numpy(s).cuda()) @ csr)
    # empty list and full list
    for sv in (np.zeros(m, bool), np.ones(m, bool)):
        cbe = be.CompactBinary.from_array(torch.from_numpy(sv).cuda())
        np.testing.assert_allclose((cbe @ csr).cpu().numpy(), O.binary_csrmv(w, idx, ptr, sv, shape=(m, k), transpose=True),
                                   rtol=1e-5, atol=1e-4)


def test_compacted_ids_on_fixed_num_and_abi_contract(be):
    import ctypes
    from brainevent_amd import _array as A, _lib
    rng = np.random.default_rng(8)
    n_pre, n_post, nc = 600, 900, 10
    idx = rng.integers(0, n_post, (n_pre, nc)).astype(np.int32)
    w = rng.random((n_pre, nc)).astype(np.float32)
    s = rng.random(n_pre) < 0.3
    M = be.FixedNumPerPre((torch.from_numpy(w).cuda(), torch.from_numpy(idx).cuda()), shape=(n_pre, n_post))
    cb = be.CompactBinary.from_array(torch.from_numpy(s).cuda())
    np.testing.assert_allclose((cb @ M).cpu().numpy(), O.binary_fcnmv(w, idx, s, shape=(n_pre, n_post), transpose=True),
                               rtol=1e-5, atol=1e-5)
    # the id list is a single-vector encoding: a batch is refused with BE_ERR_UNSUPPORTED, never guessed at
    ids = A.ActiveIds(cb.active_ids, cb.n_active, n_pre)
    f = _lib.fn('be_binary_csrmm_t', ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                                    ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int,
                                                    ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                                    ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p])
    out = torch.empty((2, n_post), dtype=torch.float32, device='cuda')
    ws = A.workspace(1 << 20)
    rc = f(A.ptr(M.data), 0, A.BE_F32, A.ptr(M.indices), None, 0, nc, ctypes.c_void_p(ids.data_ptr()), A.BE_SPIKE_IDS,
           A.ptr(out), n_pre, n_post, 2, A.ptr(ws), ws.numel(), A.stream_ptr())
    assert rc == -5


# ---------------------------------------------------------------------------------------------------
# round 4: packed events into the dense and JITC kernels (the reference's kernels pack on entry,
# brainevent/_jit_scalar/binary_jitsmv.cu:107-125, _jit_scalar/binary_jitsmm.cu:15-26; here words that exist — a
# BitPackedBinary, what the spike exchange delivers — are consumed as they are: BE_SPIKE_BITS, no unpack launch)
# ---------------------------------------------------------------------------------------------------
def _packed_only(be, x):
    """A 1-D container that exists ONLY as words on the device (what `exchange.gather_events` returns)."""
    xt = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    return be.BitPackedBinary.from_packed(be.bitpack(xt, 0).reshape(-1), x.shape[0])


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_packed_vector_into_dense_products_without_unpacking(be, dtype):
    rng = np.random.default_rng(41)
    k, n = 1000, 777
    W = torch.from_numpy(rng.standard_normal((k, n)).astype(np.float32)).cuda().to(dtype)
    for fire in (0.02, 0.6):
        s_rows, s_cols = rng.random(k) < fire, rng.random(n) < fire
        for x, prod, plain in ((s_rows, lambda e: e @ W, lambda v: be.BinaryArray(v) @ W),
                               (s_cols, lambda e: W @ e, lambda v: W @ be.BinaryArray(v))):
            ev = _packed_only(be, x)
            got = prod(ev)
            assert ev._value is None, 'the packed-only operand was unpacked'
            want = plain(torch.from_numpy(x).cuda())
            assert torch.equal(got, want)
    # Dense container, and a CompactBinary built on the device (it carries packed words too)
    D = be.Dense(W)
    ev = _packed_only(be, s_rows)
    assert torch.equal(ev @ D, be.BinaryArray(torch.from_numpy(s_rows).cuda()) @ W) and ev._value is None


def test_packed_batches_through_the_c_abi_of_dense_and_jitc(be):
    """be_binary_densemm / be_binary_jitmm with spike_dtype = BE_SPIKE_BITS: batch rows of ceil(k / 32) words, 1 ... 40 rows,
    both directions — the same bits as the byte operand (the masks are the same; only how they are built differs)."""
    import ctypes as ct
    from brainevent_amd import _array as A, _lib
    rng = np.random.default_rng(42)
    k, n = 333, 450
    W = torch.from_numpy(rng.standard_normal((k, n)).astype(np.float32)).cuda()
    f_ws = _lib.fn('be_binary_densemm_workspace_bytes', ct.c_int64, [ct.c_int64, ct.c_int64, ct.c_int64, ct.c_int, ct.c_int])
    f = _lib.fn('be_binary_densemm', ct.c_int, [ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_int64, ct.c_int64,
                                               ct.c_int64, ct.c_int, ct.c_void_p, ct.c_int64, ct.c_void_p])
    for nb in (1, 7, 32, 40):
        for transpose in (1, 0):
            in_len, out_len = (k, n) if transpose else (n, k)
            S = torch.from_numpy(rng.random((nb, in_len)) < 0.15).cuda()
            words = be.bitpack(S, 1).contiguous()
            assert words.shape == (nb, (in_len + 31) // 32)
            ws = A.workspace(f_ws(k, n, nb, transpose, A.BE_F32))
            outs = []
            for sp, sd in ((S, A.BE_SPIKE_BOOL), (words, A.BE_SPIKE_BITS)):
                out = torch.empty((nb, out_len), dtype=torch.float32, device='cuda')
                _lib.check(f(A.ptr(W), A.BE_F32, A.ptr(sp), sd, A.ptr(out), k, n, nb, transpose, A.ptr(ws), ws.numel(), A.stream_ptr()))
                outs.append(out)
            assert torch.equal(outs[0], outs[1]), (nb, transpose)
    # JITC mm, both kernels (gather / scatter), scalar and uniform weights
    f_ws = _lib.fn('be_binary_jitmm_workspace_bytes', ct.c_int64, [ct.c_int64] * 4 + [ct.c_int])
    f = _lib.fn('be_binary_jitmm', ct.c_int, [ct.c_int, ct.c_double, ct.c_double, ct.c_int, ct.c_int64, ct.c_uint32, ct.c_void_p, ct.c_int,
                                             ct.c_void_p, ct.c_int64, ct.c_int64, ct.c_int64, ct.c_int64, ct.c_int, ct.c_void_p,
                                             ct.c_int64, ct.c_void_p])
    shape1, in_len, out_len, clen = 900, 900, 700, 20
    for mode, w0, w1 in ((0, 0.5, 0.0), (1, 0.1, 0.8)):
        for gather in (1, 0):
            for nb in (3, 33):
                S = torch.from_numpy(rng.random((nb, in_len)) < 0.2).cuda()
                words = be.bitpack(S, 1).contiguous()
                ws = A.workspace(f_ws(shape1, in_len, out_len, nb, gather))
                outs = []
                for sp, sd in ((S, A.BE_SPIKE_BOOL), (words, A.BE_SPIKE_BITS)):
                    out = torch.empty((nb, out_len), dtype=torch.float32, device='cuda')
                    _lib.check(f(mode, w0, w1, A.BE_F32, clen, 17, A.ptr(sp), sd, A.ptr(out), shape1, in_len, out_len, nb, gather,
                                 A.ptr(ws), ws.numel(), A.stream_ptr()))
                    outs.append(out)
                assert torch.equal(outs[0], outs[1]), (mode, gather, nb)


@pytest.mark.parametrize('cls_name,params', [('JITCScalarR', (np.float32(0.5), 0.05, 11)), ('JITCScalarC', (np.float32(0.5), 0.05, 11)),
                                             ('JITCUniformR', (np.float32(0.1), np.float32(0.9), 0.05, 12)),
                                             ('JITCNormalC', (np.float32(0.2), np.float32(1.1), 0.05, 13))])
@pytest.mark.parametrize('corder', [False, True])
def test_packed_vector_into_jitc_products_without_unpacking(be, cls_name, params, corder):
    rng = np.random.default_rng(43)
    shape = (600, 850)
    M = getattr(be, cls_name)(params, shape=shape, corder=corder)
    sr, sc = rng.random(shape[0]) < 0.1, rng.random(shape[1]) < 0.1
    for x, prod in ((sr, lambda e: e @ M), (sc, lambda e: M @ e)):
        ev = _packed_only(be, x)
        got = prod(ev)
        assert ev._value is None, 'the packed-only operand was unpacked'
        want = prod(be.BinaryArray(torch.from_numpy(x).cuda()))
        assert torch.equal(torch.as_tensor(got), torch.as_tensor(want))
