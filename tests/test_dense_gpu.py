"""GPU parity for BinaryArray @ dense (binary_densemv / binary_densemm) against the numpy oracle and the
reference's known-answer tests (brainevent/_event/binary_test.py:45-129)."""
import numpy as np
import pytest
import torch

from test_csr_gpu import spikes_of

pytestmark = pytest.mark.gpu
TOL = {np.float32: 1e-5, np.float64: 1e-12, np.float16: 2e-3}


@pytest.mark.parametrize('transpose', [True, False])
@pytest.mark.parametrize('kind', ['bool', 'u8', 'float'])
@pytest.mark.parametrize('dtype', [np.float32, np.float64, np.float16])
@pytest.mark.parametrize('shape', [(37, 53), (64, 128), (300, 1000)])
def test_densemv(be, oracle, transpose, kind, dtype, shape):
    rng = np.random.default_rng(shape[0])
    W = rng.normal(0, 1, shape).astype(dtype)
    k = shape[0] if transpose else shape[1]
    s = spikes_of(rng, k, 0.3, kind)
    got = be.binary_densemv(W, s, transpose=transpose)
    ref = oracle.binary_densemv(W.astype(np.float64), s, transpose)
    assert got.dtype == dtype
    np.testing.assert_allclose(got.astype(np.float64), ref, rtol=TOL[dtype], atol=TOL[dtype] * 10)


@pytest.mark.parametrize('p', [0.01, 0.5])
def test_densemv_nt_gather_and_stream_paths(be, oracle, p):
    rng = np.random.default_rng(1)
    W = rng.normal(0, 1, (200, 4096)).astype(np.float32)
    s = spikes_of(rng, 4096, p, 'bool')
    np.testing.assert_allclose(be.binary_densemv(W, s, transpose=False), oracle.binary_densemv(W.astype(np.float64), s, False),
                               rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize('transpose', [True, False])
@pytest.mark.parametrize('kind', ['bool', 'float'])
@pytest.mark.parametrize('dtype', [np.float32, np.float16])
@pytest.mark.parametrize('nb', [1, 3, 8, 33])
def test_densemm(be, oracle, transpose, kind, dtype, nb):
    rng = np.random.default_rng(nb)
    shape = (96, 264)
    W = rng.normal(0, 1, shape).astype(dtype)
    k = shape[0] if transpose else shape[1]
    S = np.stack([spikes_of(rng, k, 0.25, kind) for _ in range(nb)], axis=1)
    got = be.binary_densemm(W, S, transpose=transpose)
    ref = oracle.binary_densemm(W.astype(np.float64), S, transpose)
    assert got.shape == ref.shape and got.dtype == dtype
    np.testing.assert_allclose(got.astype(np.float64), ref, rtol=TOL[dtype], atol=TOL[dtype] * 10)


def test_densemm_bf16_torch(be, oracle):
    rng = np.random.default_rng(0)
    W = torch.tensor(rng.normal(0, 1, (128, 512)), dtype=torch.bfloat16, device='cuda')
    S = torch.tensor(rng.random((128, 16)) < 0.2, device='cuda')
    got = be.binary_densemm(W, S, transpose=True)
    ref = oracle.binary_densemm(W.float().cpu().numpy().astype(np.float64), S.cpu().numpy(), True)
    assert got.dtype == torch.bfloat16 and tuple(got.shape) == (512, 16)
    np.testing.assert_allclose(got.float().cpu().numpy(), ref, rtol=1e-2, atol=5e-2)


def test_binaryarray_dense_kats(be):
    # brainevent/_event/binary_test.py:45-99 and the docstrings at brainevent/_event/binary.py:183-185, :258-261
    W = np.array([[1., 2.], [3., 4.], [5., 6.]], np.float32)
    np.testing.assert_array_equal(be.BinaryArray(np.array([0, 1, 1], np.uint8)) @ W, [8., 10.])
    np.testing.assert_array_equal(be.BinaryArray(np.array([True, False, True])) @ W, [6., 8.])
    np.testing.assert_array_equal(be.BinaryArray(np.array([[0, 1, 1], [1, 0, 1]], np.uint8)) @ W, [[8., 10.], [6., 8.]])
    W2 = np.array([[1., 2., 3.], [4., 5., 6.]], np.float32)
    np.testing.assert_array_equal(W2 @ be.BinaryArray(np.array([0, 1, 1], np.uint8)), [5., 11.])
    np.testing.assert_array_equal(W2 @ be.BinaryArray(np.array([True, False, True])), [4., 10.])
    np.testing.assert_array_equal(W2 @ be.BinaryArray(np.array([[0, 1], [1, 0], [1, 1]], np.uint8)), [[5., 4.], [11., 10.]])


def test_binaryarray_dense_errors(be):
    W = np.ones((3, 2), np.float32)
    with pytest.raises(AssertionError):
        be.BinaryArray(np.array([1, 0], np.uint8)) @ W             # dim mismatch
    with pytest.raises(AssertionError):
        be.BinaryArray(np.array([1, 0, 1], np.uint8)) @ np.ones(3, np.float32)   # 1-D right operand
    with pytest.raises(be.MathError):
        be.BinaryArray(np.asarray(1, np.uint8)) @ W                # 0-D
    with pytest.raises(be.MathError):
        be.BinaryArray(np.ones((2, 2, 3), np.uint8)) @ W           # 3-D


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('shape,nb', [((4096, 1024), 8), ((4200, 1048), 33), ((8192, 72), 16)])
def test_densemm_mfma_both_directions(be, oracle, dtype, shape, nb):
    """Shapes that take the MFMA kernels (>= 8 batch rows; no-transpose additionally needs >= 4096 weight rows),
    including row / column counts that are not tile multiples and a k that is 8 (not 16) aligned."""
    rng = np.random.default_rng(shape[1] + nb)
    W = torch.tensor(rng.normal(0, 1, shape), dtype=dtype, device='cuda')
    Wd = W.float().cpu().numpy().astype(np.float64)
    tol = 2e-3 if dtype == torch.float16 else 2e-2
    S_nt = torch.tensor(rng.random((shape[1], nb)) < 0.3, device='cuda')           # W[m,k] @ S[k,nb]
    got = be.binary_densemm(W, S_nt, transpose=False)
    ref = oracle.binary_densemm(Wd, S_nt.cpu().numpy(), False)
    assert tuple(got.shape) == (shape[0], nb) and got.dtype == dtype
    np.testing.assert_allclose(got.float().cpu().numpy(), ref, rtol=tol, atol=tol * np.abs(ref).max())
    S_t = torch.tensor(rng.random((shape[0], nb)) < 0.3, device='cuda')            # W[k,n].T @ S[k,nb]
    got = be.binary_densemm(W, S_t, transpose=True)
    ref = oracle.binary_densemm(Wd, S_t.cpu().numpy(), True)
    assert tuple(got.shape) == (shape[1], nb)
    np.testing.assert_allclose(got.float().cpu().numpy(), ref, rtol=tol, atol=tol * np.abs(ref).max())


@pytest.mark.parametrize('seed', range(int(__import__('os').environ.get('BE_STRESS_SEEDS', 10))))
def test_dense_randomized_shapes(be, oracle, seed):
    """Random weight shapes (odd sizes, vector-unfriendly strides), batch sizes on both sides of the MFMA threshold,
    very sparse and very dense spikes, all weight dtypes, both directions, mv and mm."""
    rng = np.random.default_rng(3000 + seed)
    rows = int(rng.choice([1, 7, 64, 1000, 4096, 5003]))
    cols = int(rng.choice([1, 8, 24, 100, 1024, 2056]))
    dtype = [torch.float32, torch.float16, torch.bfloat16, torch.float64][seed % 4]
    W = torch.tensor(rng.normal(0, 1, (rows, cols)), dtype=dtype, device='cuda')
    Wd = W.double().cpu().numpy()
    tol = {torch.float32: 1e-5, torch.float64: 1e-10, torch.float16: 2e-3, torch.bfloat16: 2e-2}[dtype]
    for fire in (0.01, 0.9):
        for nb in (1, 3, 8, 40):
            S = rng.random((cols, nb)) < fire
            got = be.binary_densemm(W, torch.tensor(S, device='cuda'), transpose=False)
            ref = oracle.binary_densemm(Wd, S, False)
            np.testing.assert_allclose(got.double().cpu().numpy(), ref, rtol=tol, atol=tol * max(1.0, float(np.abs(ref).max())))
            S = rng.random((rows, nb)) < fire
            got = be.binary_densemm(W, torch.tensor(S, device='cuda'), transpose=True)
            ref = oracle.binary_densemm(Wd, S, True)
            np.testing.assert_allclose(got.double().cpu().numpy(), ref, rtol=tol, atol=tol * max(1.0, float(np.abs(ref).max())))
        s = rng.random(cols) < fire
        got = be.binary_densemv(W, torch.tensor(s, device='cuda'), transpose=False)
        ref = oracle.binary_densemv(Wd, s, False)
        np.testing.assert_allclose(got.double().cpu().numpy(), ref, rtol=tol, atol=tol * max(1.0, float(np.abs(ref).max())))
        s = rng.random(rows) < fire
        got = be.binary_densemv(W, torch.tensor(s, device='cuda'), transpose=True)
        ref = oracle.binary_densemv(Wd, s, True)
        np.testing.assert_allclose(got.double().cpu().numpy(), ref, rtol=tol, atol=tol * max(1.0, float(np.abs(ref).max())))


def test_dense_container(be, oracle):
    """``Dense`` (reference ``brainevent/_dense/main.py:60-490``): the representation contract + event-driven ``@``."""
    rng = np.random.default_rng(12)
    W = rng.standard_normal((37, 21)).astype(np.float32)
    D = be.Dense(W)
    assert D.shape == (37, 21) and D.ndim == 2 and D.T.shape == (21, 37)
    np.testing.assert_array_equal(D.todense(), W)
    np.testing.assert_array_equal(D.T.todense(), W.T)
    np.testing.assert_array_equal(D[3], W[3])
    with pytest.raises(ValueError):
        be.Dense(W, shape=(21, 37))
    with pytest.raises(ValueError):
        be.Dense(W[0])
    s_rows, s_cols = rng.random(37) < 0.4, rng.random(21) < 0.4
    np.testing.assert_allclose(be.BinaryArray(s_rows) @ D, oracle.binary_densemv(W, s_rows, True), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(D @ be.BinaryArray(s_cols), oracle.binary_densemv(W, s_cols, False), rtol=1e-5, atol=1e-5)
    S_b = rng.random((5, 37)) < 0.4                                   # batch-major events @ D
    np.testing.assert_allclose(be.BinaryArray(S_b) @ D, S_b.astype(np.float32) @ W, rtol=1e-5, atol=1e-5)
    S_c = rng.random((21, 21)) < 0.4                                  # square: orientation must not be guessed
    np.testing.assert_allclose(D @ be.BinaryArray(S_c), W @ S_c.astype(np.float32), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(be.BinaryArray(s_rows).bitpack() @ D, oracle.binary_densemv(W, s_rows, True), rtol=1e-5, atol=1e-5)
    D2 = D.with_data(W * 2)
    np.testing.assert_allclose(be.BinaryArray(s_rows) @ D2, 2 * oracle.binary_densemv(W, s_rows, True), rtol=1e-5, atol=1e-5)
    out = be.BinaryArray(torch.tensor(s_rows, device='cuda')) @ be.Dense(torch.tensor(W, device='cuda'))
    assert isinstance(out, torch.Tensor)
    with pytest.raises(NotImplementedError):
        D @ np.ones(21, np.float32)


@pytest.mark.parametrize('shape,nb', [((4096, 520), 9), ((5003, 1028), 32), ((4133, 8), 17), ((8192, 4100), 40)])
def test_densemm_f32_mfma_both_directions(be, oracle, shape, nb):
    """f32 weights, >= 8 batch rows (no-transpose: also >= 4096 weight rows): both products run on v_mfma_f32_32x32x2_f32
    (exact products w * {0, 1}; sums in another order than the vector kernel's).  Row counts that are not tile multiples, k
    that is 4- but not 8-aligned, more than 32 batch rows (two passes), float spikes."""
    rng = np.random.default_rng(shape[0] + nb)
    W = torch.tensor(rng.normal(0, 1, shape), dtype=torch.float32, device='cuda')
    Wd = W.double().cpu().numpy()
    for fire in (0.02, 0.6):
        S = rng.random((shape[1], nb)) < fire
        ref = oracle.binary_densemm(Wd, S, False)
        got = be.binary_densemm(W, torch.tensor(S, device='cuda'), transpose=False)
        assert tuple(got.shape) == (shape[0], nb) and got.dtype == torch.float32
        np.testing.assert_allclose(got.double().cpu().numpy(), ref, rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(ref).max())))
        gotf = be.binary_densemm(W, torch.tensor(np.where(S, 0.7, -1.0).astype(np.float32), device='cuda'), transpose=False)
        np.testing.assert_array_equal(gotf.cpu().numpy(), got.cpu().numpy())
        # the vector kernel (< 8 batch rows) on the first columns
        vec = be.binary_densemm(W, torch.tensor(S[:, :7].copy(), device='cuda'), transpose=False)
        np.testing.assert_allclose(vec.cpu().numpy(), got[:, :7].cpu().numpy(), rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(ref).max())))
        # transpose=True (out[n, b] = sum over active k of W[k, n]): the union-row MFMA kernel, against oracle and vector kernel
        St = rng.random((shape[0], nb)) < fire
        reft = oracle.binary_densemm(Wd, St, True)
        gott = be.binary_densemm(W, torch.tensor(St, device='cuda'), transpose=True)
        assert tuple(gott.shape) == (shape[1], nb)
        np.testing.assert_allclose(gott.double().cpu().numpy(), reft, rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(reft).max())))
        vect = be.binary_densemm(W, torch.tensor(St[:, :7].copy(), device='cuda'), transpose=True)
        np.testing.assert_allclose(vect.cpu().numpy(), gott[:, :7].cpu().numpy(), rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(reft).max())))
    # a non-finite weight in a row WITHOUT a spike must not leak (0 * inf inside an MFMA): rows without spikes are zeroed
    W2 = W.clone(); W2[3, :] = float('inf')
    St = rng.random((shape[0], nb)) < 0.3; St[3, :] = False
    gott = be.binary_densemm(W2, torch.tensor(St, device='cuda'), transpose=True)
    assert torch.isfinite(gott).all()


@pytest.mark.parametrize('dtype', [torch.float16, torch.float32, torch.bfloat16])
@pytest.mark.parametrize('nb', [1, 8, 32])
def test_dense_weights_at_odd_storage_offsets(be, dtype, nb):
    """A weight matrix that starts 1 or 3 elements into its storage (rows 2- or 4-byte aligned only, NaN in front and behind)
    through the vector and the MFMA kernels of both directions."""
    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev); g.manual_seed(3)
    n = 1032
    for off in (1, 3):
        flat = torch.full((n * n + 16,), float('nan'), dtype=dtype, device=dev)
        W = flat[off:off + n * n].view(n, n)
        W.copy_(torch.randn((n, n), device=dev, generator=g).to(dtype))
        S = torch.rand((nb, n), device=dev, generator=g) < 0.05
        if nb == 1:
            pairs = ((be.BinaryArray(S[0]) @ W, S[0].float() @ W.float()), (W @ be.BinaryArray(S[0]), W.float() @ S[0].float()))
        else:
            pairs = ((be.BinaryArray(S) @ W, S.float() @ W.float()), (W @ be.BinaryArray(S.T.contiguous()), W.float() @ S.T.float()))
        tol = 1e-5 if dtype == torch.float32 else 2e-2
        for got, ref in pairs:
            assert float((got.float() - ref).abs().max() / ref.abs().max()) < tol


@pytest.mark.parametrize('k', [16, 24, 32, 40, 72])
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16, torch.float32])
def test_batched_no_transpose_with_a_short_contraction(be, k, dtype):
    """W [4200, k] @ S.T with 8 and 32 batch columns: the MFMA kernels read 32 k per step and clamp their last loads into the
    row, which needs k >= 32 (f32: 8); shorter rows take the vector kernel.  NaN behind the matrix must stay out."""
    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev); g.manual_seed(11)
    m = 4200
    flat = torch.full((m * k + 64,), float('nan'), dtype=dtype, device=dev)
    W = flat[:m * k].view(m, k)
    W.copy_(torch.randn((m, k), device=dev, generator=g).to(dtype))
    for nb in (8, 32):
        S = torch.rand((k, nb), device=dev, generator=g) < 0.4
        got = (W @ be.BinaryArray(S)).float()
        ref = W.float() @ S.float()
        tol = 1e-5 if dtype == torch.float32 else 2e-2
        assert torch.isfinite(got).all()
        assert float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-6)) < tol


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16, torch.float32])
def test_mfma_paths_add_only_selected_rows_when_weights_are_not_finite(be, oracle, dtype):
    """The reference's dense loops only ever ADD the weight rows (columns) a batch row selects
    (brainevent/_dense/binary.py:589-632): an inf / NaN weight reaches the batch rows that have a spike on it and no other.
    Inside an MFMA 0 * inf = NaN would reach the whole tile; the library redoes such outputs by selection.  One inf and one
    NaN in a weight row that is active in ONE batch row of 32, both directions, shapes that take the MFMA kernels."""
    rng = np.random.default_rng(77)
    tol = {torch.float32: 1e-5, torch.float16: 2e-3, torch.bfloat16: 2e-2}[dtype]
    nb, hot = 32, 5                                    # batch rows; the one that selects the poisoned row / column

    def check(got, ref):
        got = got.float().cpu().numpy().astype(np.float64)
        fin = np.isfinite(ref)
        assert np.array_equal(np.isnan(got), np.isnan(ref)), 'NaN where the reference has none (or the reverse)'
        assert np.array_equal(np.isposinf(got), np.isposinf(ref)) and np.array_equal(np.isneginf(got), np.isneginf(ref))
        np.testing.assert_allclose(got[fin], ref[fin], rtol=tol, atol=tol * np.abs(ref[fin]).max())
        assert (~fin).sum() == 2 and fin[:, [b for b in range(nb) if b != hot]].all()

    # transpose=True: out[n, b] = sum over rows k active in S[k, b] of W[k, n]       (S @ W, union rows, MFMA from 8 batch rows on)
    k, n = 640, 520
    Wn = rng.normal(0, 1, (k, n))
    r0 = 123
    Wn[r0, 17], Wn[r0, 300] = np.inf, np.nan
    W = torch.tensor(Wn, dtype=dtype, device='cuda')
    S = rng.random((k, nb)) < 0.3
    S[r0, :] = False
    S[r0, hot] = True
    got = be.binary_densemm(W, torch.tensor(S, device='cuda'), transpose=True)
    with np.errstate(invalid='ignore'):
        ref = oracle.binary_densemm(W.double().cpu().numpy(), S, True)
    check(got, ref)

    # transpose=False: out[m, b] = sum over columns k active in S[k, b] of W[m, k]    (W @ S.T, MFMA from 4096 weight rows on)
    m, k = 4100, 264
    Wn = rng.normal(0, 1, (m, k))
    c0 = 200
    Wn[9, c0], Wn[4099, c0] = -np.inf, np.nan
    W = torch.tensor(Wn, dtype=dtype, device='cuda')
    S = rng.random((k, nb)) < 0.3
    S[c0, :] = False
    S[c0, hot] = True
    got = be.binary_densemm(W, torch.tensor(S, device='cuda'), transpose=False)
    with np.errstate(invalid='ignore'):
        ref = oracle.binary_densemm(W.double().cpu().numpy(), S, False)
    check(got, ref)


def test_batch_first_events_reach_the_kernels_without_a_copy(be):
    """Round 4: `BinaryArray(S [n, k]) @ X` hands the ops the view `S.T`; its transpose already is the batch-major buffer the
    kernels take, so `spikes_batch_major` returns S's own memory (before: two transposing copies per call — 13.6 us of the C5
    step).  A genuinely column-major operand (k, n) is still transposed once, and both give the same product."""
    import torch
    from brainevent_amd import _array as A
    g = torch.Generator(device='cuda'); g.manual_seed(5)
    n, k, m = 8, 640, 96
    S = torch.rand((n, k), device='cuda', generator=g) < 0.2
    bm, sd = A.spikes_batch_major(S.T)
    assert bm.data_ptr() == S.data_ptr() and tuple(bm.shape) == (n, k) and sd == A.BE_SPIKE_BOOL
    Sf = torch.where(S, torch.rand((n, k), device='cuda', generator=g) + 0.1, torch.zeros((), device='cuda'))
    bmf, sdf = A.spikes_batch_major(Sf.T)
    assert bmf.data_ptr() == Sf.data_ptr() and sdf == A.BE_SPIKE_FLOAT
    col_major = S.T.contiguous()                              # (k, n) stored as such: needs the one transpose
    bm2, _ = A.spikes_batch_major(col_major)
    assert bm2.data_ptr() != col_major.data_ptr() and torch.equal(bm2, S)
    W = torch.randn((k, m), device='cuda', generator=g)
    a = be.BinaryArray(S) @ W
    b = be.binary_densemm(W, col_major, transpose=True).T
    c = be.binary_densemm(W, S.T, transpose=True).T
    ref = S.float() @ W
    assert torch.equal(a, b) and torch.equal(a, c)
    assert torch.allclose(a, ref, rtol=1e-5, atol=1e-4)
