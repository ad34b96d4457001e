"""Float-operand twins of the CSR / fixed-number products (last clause): ``csrmv``, ``csrmm``, ``fcnmv``,
``fcnmm`` — the same matrices against a dense vector or matrix, every element of which counts.

Reference surface (read as text): ``brainevent/_csr/float.py:49-150`` (``csrmv``), ``:391-556`` (``csrmv_p_call``),
``:559-668`` (``csrmm``), ``:938-1090`` (``csrmm_p_call``), CPU loops ``:153-207`` / ``:670-744``;
``brainevent/_fcn/float.py:33-134`` (``fcnmv``), ``:136-240`` (``fcnmm``).

  transpose=False: ``A[m, k] @ v[k] -> [m]``            / ``A[m, k] @ B[k, n] -> [m, n]``
  transpose=True : ``A[m, k].T @ v[m] -> [k]``          / ``A[m, k].T @ B[m, n] -> [k, n]``

Output dtype = weights dtype; the operand is cast to it.  Not on the event-driven hot path (no autodiff, no units here): these
exist so that a container accepts a dense operand at all, through hand-written kernels like everything else — gather rows in
aligned groups of four entries, scatter through float atomics (``csrc/be_float.hip``)."""
import ctypes
from typing import Optional

import numpy as np
import torch

from . import _array as A
from ._lib import check, fn
from ._csr import _check_csr_structure_dtypes
from ._misc import _as_indptr, _as_int32_indices
from ._op import OpKernel

__all__ = ['csrmv', 'csrmm', 'csrmv_p', 'csrmm_p', 'csrmv_p_call', 'csrmm_p_call', 'fcnmv', 'fcnmm', 'fcnmv_p', 'fcnmm_p',
           'fcnmv_p_call', 'fcnmm_p_call']

c_i64, c_int, c_vp = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p
_ARGS = [c_vp, c_int, c_int, c_vp, c_vp, c_int, c_i64, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_int, c_vp, c_i64, c_vp]


def _aligned16(t: torch.Tensor) -> torch.Tensor:
    """The kernels read indices / weights in aligned groups of four: a view that starts off a 16-byte boundary is copied."""
    return t if t.data_ptr() % 16 == 0 else t.clone()


def _float_csr(weights, indices, indptr, row_len, operand, *, m: int, k: int, transpose: bool) -> torch.Tensor:
    """``operand [rows_in] | [rows_in, n]`` -> ``[rows_out] | [rows_out, n]`` through ``be_csrmm`` (``indptr=None``: rows of
    ``row_len`` entries)."""
    w = A.to_device(weights)
    flat = _aligned16(w.reshape(-1))
    homo = flat.numel() == 1
    idx = _aligned16(A.to_device(indices).reshape(-1))
    x = A.to_device(operand, dtype=w.dtype)
    vec = x.ndim == 1
    n = 1 if vec else int(x.shape[1])
    rows_out = k if transpose else m
    out = torch.empty((rows_out,) if vec else (rows_out, n), dtype=w.dtype, device=A.device())
    if rows_out == 0 or n == 0:
        return out
    if idx.numel() == 0:                 # no stored entry: every output is an empty sum
        return out.zero_()
    f_ws = fn('be_csrmm_workspace_bytes', c_i64, [c_i64, c_i64, c_i64, c_int, c_int])
    ws = A.workspace(f_ws(m, k, n, int(transpose), A.wcode(w)))
    is64 = int(indptr is not None and indptr.dtype == torch.int64)
    check(fn('be_csrmm', c_int, _ARGS)(A.ptr(flat), int(homo), A.wcode(w), A.ptr(idx), A.ptr(indptr), is64, int(row_len), A.ptr(x),
                                       A.ptr(out), m, k, n, int(idx.numel()), int(transpose), A.ptr(ws), ws.numel(),
                                       A.stream_ptr()), 'be_csrmm')
    return out


# ------------------------------------------------------------------------------------------------ CSR
def _csrmv_hip(weights, indices, indptr, vector, *, shape, transpose):
    return _float_csr(weights, indices, indptr, -1, vector, m=int(shape[0]), k=int(shape[1]), transpose=transpose)


csrmv_p = OpKernel('csrmv')
csrmv_p.def_kernel('hip', 'gpu', _csrmv_hip, asdefault=True)
csrmv_p.def_tags('csr', 'float')
csrmm_p = OpKernel('csrmm')
csrmm_p.def_kernel('hip', 'gpu', _csrmv_hip, asdefault=True)
csrmm_p.def_tags('csr', 'float')


def _check_weights(weights):
    dt = weights.dtype
    ok = dt.is_floating_point if isinstance(weights, torch.Tensor) else np.issubdtype(dt, np.floating)
    assert ok, 'Weights must be a floating-point type.'


def csrmv_p_call(weights, indices, indptr, vector, *, shape, transpose, backend=None):
    """Validate, then dispatch (reference ``brainevent/_csr/float.py:391-556``)."""
    assert indptr.ndim == 1, "Indptr must be 1D."
    assert indices.ndim == 1, "Indices must be 1D."
    _check_csr_structure_dtypes(indices, indptr)
    if transpose:
        assert shape[0] == vector.shape[0], "Shape mismatch for transpose operation."
    else:
        assert shape[1] == vector.shape[0], "Shape mismatch for non-transpose operation."
    _check_weights(weights)
    if weights.ndim == 0:
        weights = weights.reshape(1)
    return [csrmv_p(weights, indices, indptr, vector, shape=shape, transpose=transpose, backend=backend)]


csrmv_p.def_call(csrmv_p_call)


def csrmm_p_call(weights, indices, indptr, B, *, shape, transpose, backend=None):
    """Validate, then dispatch (reference ``brainevent/_csr/float.py:938-1090``)."""
    assert indptr.ndim == 1, "Indptr must be 1D."
    assert indices.ndim == 1, "Indices must be 1D."
    assert B.ndim == 2, "Matrix B must be 2D."
    _check_csr_structure_dtypes(indices, indptr)
    if transpose:
        assert shape[0] == B.shape[0], "Shape mismatch for transpose operation."
    else:
        assert shape[1] == B.shape[0], "Shape mismatch for non-transpose operation."
    _check_weights(weights)
    if weights.ndim == 0:
        weights = weights.reshape(1)
    return [csrmm_p(weights, indices, indptr, B, shape=shape, transpose=transpose, backend=backend)]


csrmm_p.def_call(csrmm_p_call)


def _structure(indices, indptr, who):
    idx = A.to_device(indices)
    ptr_ = A.to_device(indptr)
    if idx.dtype != torch.int32:
        idx = _as_int32_indices(idx, None, who, check_values=False)
    if ptr_.dtype not in (torch.int32, torch.int64):
        ptr_ = _as_indptr(ptr_, idx.shape[0], 'auto', who)
    return idx, ptr_


def _operand(x):
    return x if isinstance(x, torch.Tensor) else np.asarray(x)


def csrmv(data, indices, indptr, v, *, shape, transpose: bool = False, backend: Optional[str] = None):
    """``A @ v`` (``transpose=False``) or ``A.T @ v`` for a CSR matrix and a dense vector (reference
    ``brainevent/_csr/float.py:49-150``): every element of ``v`` contributes, whatever its sign."""
    as_np = A.wants_numpy(data, indices, indptr, v)
    idx, ptr_ = _structure(indices, indptr, 'csrmv')
    res = csrmv_p_call(A.to_device(data), idx, ptr_, _operand(v), shape=tuple(shape), transpose=transpose, backend=backend)[0]
    return A.to_result(res, as_np)


def csrmm(data, indices, indptr, B, *, shape, transpose: bool = False, backend: Optional[str] = None):
    """``A @ B`` (``transpose=False``) or ``A.T @ B`` for a CSR matrix and a dense matrix ``B`` (reference
    ``brainevent/_csr/float.py:559-668``)."""
    as_np = A.wants_numpy(data, indices, indptr, B)
    idx, ptr_ = _structure(indices, indptr, 'csrmm')
    res = csrmm_p_call(A.to_device(data), idx, ptr_, _operand(B), shape=tuple(shape), transpose=transpose, backend=backend)[0]
    return A.to_result(res, as_np)


# ------------------------------------------------------------------------------------------------ fixed-number connectivity
def _fcn_hip(weights, indices, operand, *, shape, transpose):
    """``indices [rows, n_conn]``: row ``i`` lists the columns of its ``n_conn`` entries; ``shape = (rows, columns)``."""
    rows, n_conn = int(indices.shape[0]), int(indices.shape[1])
    return _float_csr(weights, indices, None, n_conn, operand, m=rows, k=int(shape[1]), transpose=transpose)


fcnmv_p = OpKernel('fcnmv')
fcnmv_p.def_kernel('hip', 'gpu', _fcn_hip, asdefault=True)
fcnmv_p.def_tags('fcn', 'float')
fcnmm_p = OpKernel('fcnmm')
fcnmm_p.def_kernel('hip', 'gpu', _fcn_hip, asdefault=True)
fcnmm_p.def_tags('fcn', 'float')


def _check_fcn(weights, indices, operand, shape, transpose, matrix: bool):
    assert indices.ndim == 2, "indices must be [rows, n_conn]."
    assert int(indices.shape[0]) == int(shape[0]), f"indices rows {indices.shape[0]} != shape[0] {shape[0]}"
    w_n = int(np.prod(tuple(weights.shape))) if weights.ndim else 1
    assert w_n == 1 or tuple(weights.shape) == tuple(indices.shape), (
        f"weights must be a scalar / size-1 array or match indices {tuple(indices.shape)}, got {tuple(weights.shape)}")
    assert operand.ndim == (2 if matrix else 1), "a matrix operand is 2D, a vector operand 1D."
    need = shape[0] if transpose else shape[1]
    assert int(operand.shape[0]) == int(need), f"operand has {operand.shape[0]} rows, the product needs {need}"
    _check_weights(weights)


def fcnmv_p_call(weights, indices, vector, *, shape, transpose, backend=None):
    _check_fcn(weights, indices, vector, shape, transpose, matrix=False)
    return [fcnmv_p(weights.reshape(1) if weights.ndim == 0 else weights, indices, vector, shape=shape, transpose=transpose,
                    backend=backend)]


def fcnmm_p_call(weights, indices, matrix, *, shape, transpose, backend=None):
    _check_fcn(weights, indices, matrix, shape, transpose, matrix=True)
    return [fcnmm_p(weights.reshape(1) if weights.ndim == 0 else weights, indices, matrix, shape=shape, transpose=transpose,
                    backend=backend)]


fcnmv_p.def_call(fcnmv_p_call)
fcnmm_p.def_call(fcnmm_p_call)


def _fcn_indices(indices):
    idx = A.to_device(indices)
    if idx.dtype != torch.int32:
        idx = _as_int32_indices(idx.reshape(-1), None, 'fcn', check_values=False).reshape(idx.shape)
    return idx


def fcnmv(weights, indices, vector, *, shape, transpose: bool, backend: Optional[str] = None):
    """``W @ v`` / ``W.T @ v`` for fixed-number connectivity (reference ``brainevent/_fcn/float.py:33-134``): ``indices [rows,
    n_conn]`` lists each row's columns, ``weights`` matches it or is one shared value; ``shape = (rows, columns)``."""
    as_np = A.wants_numpy(weights, indices, vector)
    res = fcnmv_p_call(A.to_device(weights), _fcn_indices(indices), _operand(vector), shape=tuple(shape), transpose=transpose,
                       backend=backend)[0]
    return A.to_result(res, as_np)


def fcnmm(weights, indices, matrix, *, shape, transpose: bool, backend: Optional[str] = None):
    """``W @ M`` / ``W.T @ M`` for fixed-number connectivity (reference ``brainevent/_fcn/float.py:136-240``)."""
    as_np = A.wants_numpy(weights, indices, matrix)
    res = fcnmm_p_call(A.to_device(weights), _fcn_indices(indices), _operand(matrix), shape=tuple(shape), transpose=transpose,
                       backend=backend)[0]
    return A.to_result(res, as_np)
