"""Index-structure conversions between CSR, COO and CSC (reference ``brainevent/_misc.py:871-1085`` and ``:1516-1700``:
``csr_to_coo_index``, ``coo_to_csc_index``, ``coo2csr``, ``csr_to_csc_index``, ``csc_to_csr_index``) and the
fixed-connection-number companions (``:1135-1153``, ``:1255-1320``: ``fixed_conn_num_csr_indptr``,
``fixed_conn_num_csc_structure``, ``fixed_conn_num_to_csc``).

One-off preprocessing, done on the device with stable sorts (entries that share a column keep their row order, as the
reference's ``argsort(..., stable=True)``); numpy in -> numpy out, device tensors in -> device tensors out.  Coordinates
are int32; offsets and permutations are int32 until the entry count needs int64.  Limited to what one device sort holds
(about 2^31 entries)."""
from typing import Tuple

import numpy as np
import torch

from . import _array as A

__all__ = ['csr_to_coo_index', 'coo_to_csc_index', 'coo2csr', 'csr_to_csc_index', 'csc_to_csr_index',
           'fixed_conn_num_csr_indptr', 'fixed_conn_num_csc_structure', 'fixed_conn_num_to_csc']


def _offset_dtype(nnz: int) -> torch.dtype:
    return torch.int64 if nnz > np.iinfo(np.int32).max else torch.int32


def _finish(as_np: bool, *tensors):
    return tuple(None if t is None else (t.cpu().numpy() if as_np else t) for t in tensors)


def csr_to_coo_index(indptr, indices):
    """``(pre_ids, post_ids)``: the row of every stored element, and ``indices`` itself."""
    as_np = A.wants_numpy(indptr, indices)
    ptr = A.to_device(indptr).to(torch.int64)
    idx = A.to_device(indices)
    m = int(ptr.numel()) - 1
    rows = torch.repeat_interleave(torch.arange(m, dtype=torch.int32, device=ptr.device), ptr[1:] - ptr[:-1])
    return _finish(as_np, rows, idx)


def _group(keys: torch.Tensor, values: torch.Tensor, n_groups: int):
    """Stable grouping of ``values`` by ``keys`` in ``[0, n_groups)``: (offsets, grouped values, permutation)."""
    nnz = int(keys.numel())
    if nnz > (1 << 31):
        raise MemoryError("index conversion: the device sort of more than 2^31 entries is not supported.")
    od = _offset_dtype(nnz)
    order = torch.argsort(keys, stable=True)
    counts = torch.bincount(keys.to(torch.int64), minlength=n_groups)
    offsets = torch.zeros(n_groups + 1, dtype=torch.int64, device=keys.device)
    torch.cumsum(counts, 0, out=offsets[1:])
    return offsets.to(od), values[order].to(torch.int32).contiguous(), order.to(od)


def coo_to_csc_index(pre_ids, indices, *, shape: Tuple[int, int]):
    """``(csc_indptr, csc_indices, post_positions)`` from COO ``(pre_ids, indices)``; ``post_positions[j]`` is the COO
    slot stored at CSC slot ``j``."""
    as_np = A.wants_numpy(pre_ids, indices)
    rows, cols = A.to_device(pre_ids), A.to_device(indices)
    return _finish(as_np, *_group(cols, rows, int(shape[1])))


def coo2csr(row_ids, col_ids, *, shape: Tuple[int, int]):
    """``(csr_indptr, csr_indices, positions)`` from COO ``(row_ids, col_ids)``."""
    as_np = A.wants_numpy(row_ids, col_ids)
    rows, cols = A.to_device(row_ids), A.to_device(col_ids)
    return _finish(as_np, *_group(rows, cols, int(shape[0])))


def csr_to_csc_index(csr_indptr, csr_indices, *, shape: Tuple[int, int], include_perm: bool = True, method: str = 'coo',
                     column_block_size: int = 4096):
    """``(csc_indptr, csc_indices, post_positions)`` of the same matrix; ``post_positions`` reorders a CSR data array into
    CSC order (``None`` when ``include_perm=False``).  ``method`` and ``column_block_size`` are accepted for signature
    compatibility: there is one implementation here (device stable sort)."""
    assert isinstance(shape, (tuple, list)) and len(shape) == 2, "Shape must have exactly two dimensions (rows, columns)"
    assert shape[0] > 0 and shape[1] > 0, "Shape dimensions must be positive integers"
    if method not in ('coo', 'numpy', 'gpu_column_block'):
        raise ValueError(f"Unknown csr_to_csc_index method {method!r}; expected 'coo', 'numpy', or 'gpu_column_block'.")
    as_np = A.wants_numpy(csr_indptr, csr_indices)
    rows, cols = csr_to_coo_index(A.to_device(csr_indptr), A.to_device(csr_indices))
    ptr, idx, perm = _group(cols, rows, int(shape[1]))
    return _finish(as_np, ptr, idx, perm if include_perm else None)


def csc_to_csr_index(csc_indptr, csc_indices, *, shape: Tuple[int, int], include_perm: bool = True):
    """Inverse companion: the CSC arrays of ``W (n_rows, n_cols)`` are the CSR arrays of ``W.T``."""
    return csr_to_csc_index(csc_indptr, csc_indices, shape=(int(shape[1]), int(shape[0])), include_perm=include_perm)


def fixed_conn_num_csr_indptr(indices):
    """The implicit CSR ``indptr`` of fixed-number connectivity ``indices (n_pre, n_conn)``: ``arange(n_pre + 1) * n_conn``,
    int32 until the entry count needs int64 (reference ``_misc.py:1135-1153``)."""
    assert indices.ndim == 2, f'Indices must be 2D, got {indices.ndim}D.'
    n_pre, n_conn = int(indices.shape[0]), int(indices.shape[1])
    dt = _offset_dtype(n_pre * n_conn)
    if isinstance(indices, np.ndarray):
        return np.arange(n_pre + 1, dtype=np.int64 if dt == torch.int64 else np.int32) * n_conn
    return torch.arange(n_pre + 1, dtype=dt, device=indices.device) * n_conn


def fixed_conn_num_csc_structure(indices, *, shape: Tuple[int, int]):
    """``(csc_indptr, csc_indices, perm)`` of row-major fixed-number connectivity: for every post neuron the pre neurons
    that target it (stable: pre order kept), and the permutation that reorders the flattened weights into CSC order
    (reference ``_misc.py:1255-1296``).  Offsets and ``perm`` follow the entry count's dtype, coordinates are int32."""
    assert indices.ndim == 2, f'Indices must be 2D, got {indices.ndim}D.'
    n_pre, n_post = int(shape[0]), int(shape[1])
    assert int(indices.shape[0]) == n_pre, (
        f'Pre size mismatch: indices.shape[0] ({indices.shape[0]}) != shape[0] ({n_pre})')
    as_np = A.wants_numpy(indices)
    idx = A.to_device(indices)
    ptr, rows, perm = csr_to_csc_index(fixed_conn_num_csr_indptr(idx), idx.reshape(-1), shape=(n_pre, n_post))
    return _finish(as_np, ptr, rows.to(torch.int32), perm)


def fixed_conn_num_to_csc(weights, indices, *, shape: Tuple[int, int]):
    """``(csc_data, csc_indices, csc_indptr)``: the CSC mirror of fixed-number weights and connectivity; a size-1 weight
    stays size-1 (reference ``_misc.py:1299-1320``)."""
    as_np = A.wants_numpy(weights, indices)
    w = A.to_device(weights)
    if w.ndim == 0:
        w = w.reshape(1)
    if w.ndim == 1:
        assert w.numel() == 1, f'When weights is 1D, it should be a scalar (size 1), got {w.numel()}.'
    elif w.ndim != 2:
        raise ValueError(f'weight dim should be 2, 1, or 0, but got {w.ndim}')
    ptr, rows, perm = fixed_conn_num_csc_structure(A.to_device(indices), shape=shape)
    data = w.reshape(1) if w.ndim == 1 else w.reshape(-1)[perm.long()]
    return _finish(as_np, data, rows, ptr)
