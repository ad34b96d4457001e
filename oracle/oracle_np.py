"""oracle_np.py — numpy restatement of the reference's CPU algorithms for the hot path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg as the checker.  The product package (brainevent_amd/) never imports this module.

Parity status: pinned — against the reference's known-answer tests and against vectors generated
from the reference's own numpy golden model (see tests/golden/ and oracle/gen_golden.py).

Every function cites the reference lines it follows (paths relative to the reference checkout).
"""
import math

import numpy as np

MV_STRIDE = 32   # brainevent/_misc.py:37
MM_STRIDE = 4    # brainevent/_misc.py:38


# ----------------------------------------------------------------------------------------------------
# event predicate  (brainevent/include/cuda_common.h:120-131; brainevent/_csr/binary.py:399-451)
# ----------------------------------------------------------------------------------------------------
def active(v):
    v = np.asarray(v)
    if v.dtype == np.bool_ or np.issubdtype(v.dtype, np.integer):
        return v != 0
    return v > 0


# ----------------------------------------------------------------------------------------------------
# CSR  (brainevent/_csr/binary.py:387-489 mv, :1029-1160 mm)
# ----------------------------------------------------------------------------------------------------
def binary_csrmv(weights, indices, indptr, v, shape, transpose):
    """transpose: posts[indices[j]] += w for active rows (:399-405, :446-451);
    else posts[i] = sum_j w[j] * e(v[indices[j]]) (:421-428, :466-472).  Accumulates in the weight dtype."""
    weights = np.asarray(weights).reshape(-1)
    indices = np.asarray(indices)
    indptr = np.asarray(indptr)
    act = active(v)
    homo = weights.size == 1
    m, k = shape
    row_of = np.repeat(np.arange(m), np.diff(indptr))
    w_all = np.broadcast_to(weights, indices.shape) if homo else weights
    if transpose:
        out = np.zeros(k, dtype=weights.dtype)
        sel = act[row_of]
        np.add.at(out, indices[sel], w_all[sel])
    else:
        out = np.zeros(m, dtype=weights.dtype)
        sel = act[indices]
        np.add.at(out, row_of[sel], w_all[sel])
    return out


def binary_csrmm(weights, indices, indptr, B, shape, transpose):
    """Column-wise application of the mv definition (brainevent/_csr/binary.py:1029-1160)."""
    B = np.asarray(B)
    cols = [binary_csrmv(weights, indices, indptr, B[:, l], shape, transpose) for l in range(B.shape[1])]
    rows = shape[1] if transpose else shape[0]
    if not cols:
        return np.zeros((rows, 0), dtype=np.asarray(weights).dtype)
    return np.stack(cols, axis=1)


# ----------------------------------------------------------------------------------------------------
# fixed-number connectivity  (brainevent/_fcn/binary.py:156-253 mv, :677-766 mm)
# ----------------------------------------------------------------------------------------------------
def binary_fcnmv(weights, indices, spikes, shape, transpose):
    """transpose: posts[indices[i, j]] += w[i, j] for active i (:167-200);
    else posts[i] = sum_j w[i, j] * e(spikes[indices[i, j]]) (:201-253)."""
    indices = np.asarray(indices)
    n_rows, n_conn = indices.shape
    weights = np.asarray(weights)
    w = weights.reshape(-1) if weights.size == 1 else weights.reshape(-1)
    indptr = np.arange(n_rows + 1, dtype=np.int64) * n_conn
    return binary_csrmv(w, indices.reshape(-1), indptr, spikes, (n_rows, shape[1]), transpose)


def binary_fcnmm(weights, indices, matrix, shape, transpose):
    """Matrix operand (:677-766): column-wise application; result (shape[1], n) if transpose else (shape[0], n)."""
    matrix = np.asarray(matrix)
    cols = [binary_fcnmv(weights, indices, matrix[:, l], shape, transpose) for l in range(matrix.shape[1])]
    rows = shape[1] if transpose else shape[0]
    if not cols:
        return np.zeros((rows, 0), dtype=np.asarray(weights).dtype)
    return np.stack(cols, axis=1)


# ----------------------------------------------------------------------------------------------------
# float-operand twins  (brainevent/_csr/float.py:153-207 mv, :670-744 mm; brainevent/_fcn/float.py — the same loops over
# rows of one length).  Sums are taken in float64 here and rounded once: the checker, not a bit-for-bit model of numba's
# fastmath accumulation order.
# ----------------------------------------------------------------------------------------------------
def csrmv(weights, indices, indptr, v, shape, transpose):
    """transpose: posts[indices[j]] += w[j] * v[i] for every row i (:188-193, :163-170);
    else posts[i] = sum_j w[j] * v[indices[j]] (:195-202, :172-180; one shared weight: ``w * sum``)."""
    weights = np.asarray(weights).reshape(-1)
    indices = np.asarray(indices)
    indptr = np.asarray(indptr)
    x = np.asarray(v, dtype=np.float64)
    m, k = shape
    row_of = np.repeat(np.arange(m), np.diff(indptr))
    w_all = (np.broadcast_to(weights, indices.shape) if weights.size == 1 else weights).astype(np.float64)
    if transpose:
        out = np.zeros(k, dtype=np.float64)
        np.add.at(out, indices, w_all * x[row_of])
    else:
        out = np.zeros(m, dtype=np.float64)
        np.add.at(out, row_of, w_all * x[indices])
    return out.astype(weights.dtype)


def csrmm(weights, indices, indptr, B, shape, transpose):
    """Matrix operand (brainevent/_csr/float.py:670-744): the mv definition per column of ``B``."""
    B = np.asarray(B)
    cols = [csrmv(weights, indices, indptr, B[:, l], shape, transpose) for l in range(B.shape[1])]
    rows = shape[1] if transpose else shape[0]
    if not cols:
        return np.zeros((rows, 0), dtype=np.asarray(weights).dtype)
    return np.stack(cols, axis=1)


def fcnmv(weights, indices, vector, shape, transpose):
    """Rows of one length (brainevent/_fcn/float.py:33-134): ``indices [rows, n_conn]``."""
    indices = np.asarray(indices)
    n_rows, n_conn = indices.shape
    indptr = np.arange(n_rows + 1, dtype=np.int64) * n_conn
    return csrmv(np.asarray(weights).reshape(-1), indices.reshape(-1), indptr, vector, (n_rows, shape[1]), transpose)


def fcnmm(weights, indices, matrix, shape, transpose):
    matrix = np.asarray(matrix)
    cols = [fcnmv(weights, indices, matrix[:, l], shape, transpose) for l in range(matrix.shape[1])]
    rows = shape[1] if transpose else shape[0]
    if not cols:
        return np.zeros((rows, 0), dtype=np.asarray(weights).dtype)
    return np.stack(cols, axis=1)


# ----------------------------------------------------------------------------------------------------
# dense  (brainevent/_dense/binary.py:168-211 mv, :579-632 mm)
# ----------------------------------------------------------------------------------------------------
def binary_densemv(weights, spikes, transpose):
    """transpose: posts += weights[i] for active i (:178-190); else posts += weights[:, i] (:193-206)."""
    weights = np.asarray(weights)
    act = active(spikes)
    return weights[act].sum(axis=0, dtype=weights.dtype) if transpose else weights[:, act].sum(axis=1, dtype=weights.dtype)


def binary_densemm(weights, spikes, transpose):
    """out[:, i_n] = sum of weights rows (transpose, :589-606) / columns (:609-632) active in spikes[:, i_n]."""
    spikes = np.asarray(spikes)
    cols = [binary_densemv(weights, spikes[:, l], transpose) for l in range(spikes.shape[1])]
    weights = np.asarray(weights)
    rows = weights.shape[1] if transpose else weights.shape[0]
    return np.stack(cols, axis=1) if cols else np.zeros((rows, 0), weights.dtype)


# ----------------------------------------------------------------------------------------------------
# light_rng  (brainevent/_numba_random.py:385-502) — uint32 arithmetic with C wraparound
# ----------------------------------------------------------------------------------------------------
_M32 = 0xFFFFFFFF


def lr_mix32(x):
    """:385-393"""
    x &= _M32
    x ^= x >> 16
    x = (x * 0x7feb352d) & _M32
    x ^= x >> 15
    x = (x * 0x846ca68b) & _M32
    x ^= x >> 16
    return x


def lr_bounded(r, bound):
    """:396-398  (r * bound) >> 32"""
    return ((r & _M32) * (bound & _M32)) >> 32


def lr_next(x):
    """:401-409  xorshift32 (13, 17, 5); 0 -> 0x6d2b79f5"""
    x &= _M32
    x ^= (x << 13) & _M32
    x ^= x >> 17
    x ^= (x << 5) & _M32
    return 0x6d2b79f5 if x == 0 else x


def lr_init(seed, row, chunk_id, lane):
    """:412-421"""
    x = (seed & _M32) ^ 0xd1b54a35
    x ^= ((row & _M32) * 0x85ebca6b) & _M32
    x ^= ((chunk_id & _M32) * 0xc2b2ae35) & _M32
    x ^= ((lane & _M32) * 0x27d4eb2d) & _M32
    x = lr_mix32(x)
    return 0x6d2b79f5 if x == 0 else x


def lr_initial_q(state, cl):
    """:489-502  stationary residual, two draws per rejection round; returns (q, state)"""
    n = (cl - 1) & _M32
    while True:
        state = lr_next(state)
        q = lr_bounded(state, n)
        state = lr_next(state)
        gate = lr_bounded(state, n)
        if gate < ((n - q) & _M32):
            return q, state


def lr_uniform01(seed, row, col):
    """:424-430  24-bit uniform variate of an edge (float32)"""
    h = (seed & _M32) ^ 0xa0761d65
    h ^= ((row & _M32) * 0xe7037ed1) & _M32
    h ^= ((col & _M32) * 0x8ebc6af1) & _M32
    h = lr_mix32(h)
    return np.float32(h & 0x00FFFFFF) * np.float32(1.0 / 16777216.0)


def lr_normal01(seed, row, col):
    """:433-486  Acklam probit of the 24-bit hash, evaluated in float32"""
    f = np.float32
    u = f(lr_uniform01(seed, row, col))
    lo, hi = f(1e-10), f(1.0 - 1e-10)
    u = lo if u < lo else (hi if u > hi else u)
    a1, a2, a3, a4, a5, a6 = f(-39.696830), f(220.94609), f(-275.92851), f(138.35775), f(-30.664799), f(2.5066283)
    b1, b2, b3, b4, b5 = f(-54.476099), f(161.58584), f(-155.69898), f(66.801312), f(-13.280681)
    c1, c2, c3, c4, c5, c6 = f(-0.007784894), f(-0.32239646), f(-2.4007583), f(-2.5497325), f(4.3746641), f(2.9381640)
    d1, d2, d3, d4 = f(0.007784696), f(0.32246713), f(2.4451342), f(3.7544087)
    one = f(1.0)
    if u < f(0.02425):
        v = f(np.sqrt(f(-2.0) * np.log(u)))
        z = f((((((c1 * v + c2) * v + c3) * v + c4) * v + c5) * v + c6) / ((((d1 * v + d2) * v + d3) * v + d4) * v + one))
        z = f(-z)
    elif u > f(0.97575):
        v = f(np.sqrt(f(-2.0) * np.log(one - u)))
        z = f((((((c1 * v + c2) * v + c3) * v + c4) * v + c5) * v + c6) / ((((d1 * v + d2) * v + d3) * v + d4) * v + one))
    else:
        v = f(u - f(0.5))
        r = f(v * v)
        z = f((((((a1 * r + a2) * r + a3) * r + a4) * r + a5) * r + a6) * v /
              (((((b1 * r + b2) * r + b3) * r + b4) * r + b5) * r + one))
    return f(z)


def conn_length(prob):
    """brainevent/_data.py:1212-1245 (ceil(2/prob) as int32); 0 stands for prob == 0
    (brainevent/_jit_uniform/_test_util.py:72-76)."""
    prob = float(prob)
    if prob == 0.0:
        return 0
    return max(2, int(math.ceil(2.0 / prob)))


def default_chunk_size(n_cols, target_chunks=4):
    """brainevent/_misc.py:74-122"""
    return max(1, (int(n_cols) + int(target_chunks) - 1) // int(target_chunks))


def jit_iter_edges(seed, clen, n_rows, walk_len, stride, chunk_size):
    """Yield (rng_row, rng_col) of every generated edge: the walk of
    brainevent/_jit_scalar/binary.py:340-377 (gather) / :381-416 (scatter); rows are the walk owners."""
    cl = max(2, int(clen))
    n_chunks = 0 if walk_len <= 0 else (int(walk_len) + chunk_size - 1) // chunk_size
    for row in range(int(n_rows)):
        for chunk_id in range(n_chunks):
            cs = chunk_id * chunk_size
            width = min(cs + chunk_size, int(walk_len)) - cs
            for lane in range(int(stride)):
                state = lr_init(seed, row, chunk_id, lane)
                q, state = lr_initial_q(state, cl)
                lj = lane + stride * q
                while lj < width:
                    yield row, cs + lj
                    state = lr_next(state)
                    q = q + 1 + lr_bounded(state, cl - 1)
                    lj = lane + stride * q


def jit_generator_matrix(mode, w0, w1, prob, seed, *, shape, transpose, corder, matrix_mode='mv', dtype=np.float64):
    """Dense G[rng_row, rng_col] in the RNG orientation: rows = walk owners.
    gather (corder=True): rows = outputs, cols = inputs; scatter: rows = inputs, cols = outputs.
    mode: 's' (w0), 'u' (w0 + u01 * (w1 - w0)), 'n' (w0 + n01 * w1); duplicates cannot occur (q strictly grows)."""
    in_len = shape[0] if transpose else shape[1]
    out_len = shape[1] if transpose else shape[0]
    n_rows, walk = (out_len, in_len) if corder else (in_len, out_len)
    G = np.zeros((n_rows, walk), dtype=dtype)
    clen = conn_length(prob)
    if clen == 0:
        return G
    stride = MV_STRIDE if matrix_mode == 'mv' else MM_STRIDE
    cs = default_chunk_size(shape[1])
    seed &= _M32
    for r, c in jit_iter_edges(seed, clen, n_rows, walk, stride, cs):
        if mode == 's':
            G[r, c] = w0
        elif mode == 'u':
            G[r, c] = dtype(w0) + dtype(lr_uniform01(seed, r, c)) * (dtype(w1) - dtype(w0))
        else:
            G[r, c] = dtype(w0) + dtype(lr_normal01(seed, r, c)) * dtype(w1)
    return G


def binary_jitmv(mode, w0, w1, prob, vector, seed, *, shape, transpose, corder, wdtype=np.float32):
    """binary_jit{s,u,n}mv (brainevent/_jit_scalar/binary.py:289-421, _jit_uniform/binary.py:292-415,
    _jit_normal/binary.py:307-410) evaluated through the dense generator matrix: every edge weight is
    formed in the weight dtype (as the numba kernels do: ``w_low0 + u01 * span``), sums run in float64
    (numba: ``out = np.float64(0.)``)."""
    G = jit_generator_matrix(mode, w0, w1, prob, seed, shape=shape, transpose=transpose, corder=corder, matrix_mode='mv',
                             dtype=wdtype).astype(np.float64)
    act = active(vector).astype(np.float64)
    return G @ act if corder else G.T @ act


def binary_jitmm(mode, w0, w1, prob, B, seed, *, shape, transpose, corder, wdtype=np.float32):
    """binary_jit{s,u,n}mm (brainevent/_jit_scalar/binary.py:833-957): stride-4 matrix, shared by all columns."""
    G = jit_generator_matrix(mode, w0, w1, prob, seed, shape=shape, transpose=transpose, corder=corder, matrix_mode='mm',
                             dtype=wdtype).astype(np.float64)
    act = active(B).astype(np.float64)
    return G @ act if corder else G.T @ act


def jitmv(mode, w0, w1, prob, vector, seed, *, shape, transpose, corder, wdtype=np.float32):
    """Float-operand twin jit{s,u,n}mv (brainevent/_jit_scalar/float.py:838-905 and the uniform / normal files): the same
    generator matrix as :func:`binary_jitmv` against the values of ``vector`` (sums in float64)."""
    G = jit_generator_matrix(mode, w0, w1, prob, seed, shape=shape, transpose=transpose, corder=corder, matrix_mode='mv',
                             dtype=wdtype).astype(np.float64)
    x = np.asarray(vector, dtype=np.float64)
    return G @ x if corder else G.T @ x


def jitmm(mode, w0, w1, prob, B, seed, *, shape, transpose, corder, wdtype=np.float32):
    """Float-operand twin jit{s,u,n}mm (brainevent/_jit_scalar/float.py:1331-1420): the stride-4 matrix."""
    G = jit_generator_matrix(mode, w0, w1, prob, seed, shape=shape, transpose=transpose, corder=corder, matrix_mode='mm',
                             dtype=wdtype).astype(np.float64)
    X = np.asarray(B, dtype=np.float64)
    return G @ X if corder else G.T @ X


# =====================================================================================================
# event encodings
# =====================================================================================================
def bitpack(arr, axis):
    """uint32 words along ``axis``; bit b of word w = element 32 w + b; non-zero is True
    (reference brainevent/_event/bitpack_binary.py:32-75)."""
    a = np.asarray(arr) != 0
    axis = axis % a.ndim
    n = a.shape[axis]
    nw = (n + 31) // 32
    pad = [(0, 0)] * a.ndim
    pad[axis] = (0, nw * 32 - n)
    a = np.pad(a, pad).astype(np.uint64)
    shp = list(a.shape)
    shp[axis] = nw
    shp.insert(axis + 1, 32)
    a = a.reshape(shp)
    sh = [1] * a.ndim
    sh[axis + 1] = 32
    shifts = np.arange(32, dtype=np.uint64).reshape(sh)
    return (a << shifts).sum(axis=axis + 1).astype(np.uint32)


def compact_1d(x):
    """(sorted active positions, count) of a 1-D array; active = non-zero (reference brainevent/_event/compact.py:81-92)."""
    ids = np.flatnonzero(np.asarray(x) != 0).astype(np.int32)
    return ids, ids.size


def compact_2d(x):
    """Rows of ``x (n, batch)`` active in any batch column (reference brainevent/_event/compact.py:115-126)."""
    ids = np.flatnonzero((np.asarray(x) != 0).any(axis=1)).astype(np.int32)
    return ids, ids.size
