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
