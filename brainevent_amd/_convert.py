"""Index-structure conversions between CSR, COO and CSC (reference ``brainevent/_misc.py:871-1085`` and ``:1516-1700``:
``csr_to_coo_index``, ``coo_to_csc_index``, ``coo2csr``, ``csr_to_csc_index``, ``csc_to_csr_index``) and the
fixed-connection-number companions (``:1135-1153``, ``:1255-1320``: ``fixed_conn_num_csr_indptr``,
``fixed_conn_num_csc_structure``, ``fixed_conn_num_to_csc``).

One-off preprocessing, done on the device with stable sorts (entries that share a column keep their row order, as the
reference's ``argsort(..., stable=True)``); numpy in -> numpy out, device tensors in -> device tensors out.  Coordinates
are int32; offsets and permutations are int32 until the entry count needs int64.

Two implementations of the CSR -> CSC conversion, named as in the reference (``method=``): ``'coo'`` / ``'numpy'`` — a stable
device sort (limited to what one sort holds, 2^31 - 1 entries; beyond that the call takes the other route by itself) — and
``'gpu_column_block'`` — :class:`CscBuilder`: the library's count / scan / fill kernels (``csrc/be_convert.hip``,
``be_csr_to_csc_*``), 64-bit offsets, any entry count, column blocks of any size; the entries of a column come out in
unspecified order, as from the reference's kernel (``_csr/csr_to_csc.cu:26-27``).  The mirrors of the containers
(``CSR.build_mirror``, ``FixedNumConn.build_mirror``) are built by the second one."""
import ctypes
from typing import Optional, Tuple

import numpy as np
import torch

from . import _array as A
from ._lib import check, fn

c_i64, c_int, c_vp = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p

__all__ = ['csr_to_coo_index', 'coo_to_csc_index', 'coo2csr', 'csr_to_csc_index', 'csc_to_csr_index',
           'fixed_conn_num_csr_indptr', 'fixed_conn_num_csc_structure', 'fixed_conn_num_to_csc', 'CscBuilder']


def _offset_dtype(nnz: int) -> torch.dtype:
    return torch.int64 if nnz > np.iinfo(np.int32).max else torch.int32


def _i32(t: torch.Tensor) -> torch.Tensor:
    return t if t.dtype == torch.int32 else t.to(torch.int32)


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


class CscBuilder:
    """Column-block CSR -> CSC conversion on the device (``be_csr_to_csc_count / _indptr / _fill_block``; reference
    ``brainevent/_misc.py:1380-1513`` + ``brainevent/_csr/csr_to_csc.cu``).  The constructor counts the columns and scans
    them into ``csc_indptr`` (int64, device); :meth:`block` then produces any column block ``[c0, c1)`` — its row ids, its
    weights moved along, optionally the permutation — without the rest of the CSC arrays being resident.  ``indptr=None`` with
    ``row_len`` describes fixed-length rows (``FixedNumPerPre`` indices, reference ``fixed_conn_num_csc_structure``)."""

    def __init__(self, indptr, indices, *, shape: Tuple[int, int], row_len: int = -1):
        self.m, self.k = int(shape[0]), int(shape[1])
        self.indices = A.to_device(indices).reshape(-1)
        assert self.indices.dtype == torch.int32, "column ids must be int32"
        self.indptr = None if indptr is None else A.to_device(indptr)
        assert self.indptr is None or self.indptr.dtype in (torch.int32, torch.int64)
        assert self.indptr is not None or row_len >= 0
        self.row_len = int(row_len) if self.indptr is None else -1
        self.nnz = int(self.indices.numel())
        dev, st = A.device(), A.stream_ptr()
        self.counts = torch.empty(self.k, dtype=torch.int64, device=dev)
        check(fn('be_csr_to_csc_count', c_int, [c_vp, c_i64, c_i64, c_vp, c_vp])(
            A.ptr(self.indices), self.nnz, self.k, A.ptr(self.counts), st), 'be_csr_to_csc_count')
        self.csc_indptr = torch.empty(self.k + 1, dtype=torch.int64, device=dev)
        scratch = A.workspace(fn('be_csr_to_csc_scratch_bytes', c_i64, [c_i64])(self.k))
        total = c_i64(0)
        check(fn('be_csr_to_csc_indptr', c_int, [c_vp, c_i64, c_vp, c_int, ctypes.POINTER(c_i64), c_vp, c_i64, c_vp])(
            A.ptr(self.counts), self.k, A.ptr(self.csc_indptr), 1, ctypes.byref(total), A.ptr(scratch), scratch.numel(), st),
            'be_csr_to_csc_indptr')
        if int(total.value) != self.nnz:
            raise ValueError(f"csr_to_csc: {self.nnz - int(total.value)} of {self.nnz} column ids lie outside [0, {self.k}).")

    @property
    def max_col_count(self) -> int:
        return int(self.counts.max().item()) if self.k > 0 and self.nnz > 0 else 0

    def block_entries(self, c0: int, c1: int) -> int:
        return int((self.csc_indptr[c1] - self.csc_indptr[c0]).item())

    def block_indptr(self, c0: int, c1: int, dtype=None) -> torch.Tensor:
        """``csc_indptr`` of the columns ``[c0, c1)`` relative to the block (first element 0)."""
        ptr = self.csc_indptr[c0:c1 + 1] - self.csc_indptr[c0]
        return ptr if dtype is None else ptr.to(dtype)

    def block(self, c0: int, c1: int, data: Optional[torch.Tensor] = None, perm: bool = False):
        """``(rows, data_block, perm)`` of the columns ``[c0, c1)``: row ids (int32), the weights moved along (``None``
        without per-entry ``data``) and the source positions (int32 until the entry count needs int64; ``None`` unless asked)."""
        c0, c1 = int(c0), int(c1)
        assert 0 <= c0 <= c1 <= self.k
        dev, st = A.device(), A.stream_ptr()
        n = self.block_entries(c0, c1)
        rows = torch.empty(n, dtype=torch.int32, device=dev)
        w_out, wb = None, 0
        if data is not None and data.numel() > 1:
            data = A.to_device(data).reshape(-1)
            assert data.numel() == self.nnz, "data must hold one weight per stored entry"
            wb = data.element_size()
            w_out = torch.empty(n, dtype=data.dtype, device=dev)
        p_out, p64 = None, int(self.nnz > np.iinfo(np.int32).max)
        if perm:
            p_out = torch.empty(n, dtype=torch.int64 if p64 else torch.int32, device=dev)
        cursor = torch.empty(max(c1 - c0, 1), dtype=torch.int64, device=dev)
        is64 = int(self.indptr is not None and self.indptr.dtype == torch.int64)
        f = fn('be_csr_to_csc_fill_block', c_int,
               [c_vp, c_vp, c_int, c_i64, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_vp])
        check(f(A.ptr(self.indices), A.ptr(self.indptr), is64, self.row_len, self.m, self.nnz, c0, c1, A.ptr(self.csc_indptr),
                A.ptr(cursor), A.ptr(rows), A.ptr(p_out), p64, A.ptr(data if wb else None), wb, A.ptr(w_out), st),
              'be_csr_to_csc_fill_block')
        return rows, w_out, p_out

    def offsets(self) -> torch.Tensor:
        """``csc_indptr`` in the offset dtype of the entry count (int32 until it needs int64)."""
        return self.csc_indptr.to(_offset_dtype(self.nnz))


def gather_by_perm(src: torch.Tensor, perm: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``out[i] = src[perm[i]]`` on the device (``be_gather_by_perm``; 2 / 4 / 8-byte elements, int32 / int64 ``perm``)."""
    src, perm = A.to_device(src).reshape(-1), A.to_device(perm).reshape(-1)
    assert perm.dtype in (torch.int32, torch.int64)
    if out is None:
        out = torch.empty(perm.numel(), dtype=src.dtype, device=src.device)
    assert out.dtype == src.dtype and out.numel() == perm.numel() and out.is_contiguous()
    check(fn('be_gather_by_perm', c_int, [c_vp, c_int, c_vp, c_int, c_i64, c_vp, c_vp])(
        A.ptr(src), src.element_size(), A.ptr(perm), int(perm.dtype == torch.int64), perm.numel(), A.ptr(out), A.stream_ptr()),
        'be_gather_by_perm')
    return out


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
    nnz = int(A.to_device(csr_indices).numel())
    if method == 'gpu_column_block' or nnz > np.iinfo(np.int32).max:
        try:
            column_block_size = int(column_block_size)
        except (TypeError, ValueError) as exc:
            raise ValueError("column_block_size must be a positive integer") from exc
        if column_block_size <= 0:
            raise ValueError("column_block_size must be a positive integer")
        b = CscBuilder(csr_indptr, _i32(A.to_device(csr_indices)), shape=shape)
        rows, _, perm = b.block(0, b.k, perm=include_perm)       # everything stays on the device: one block
        return _finish(as_np, b.offsets(), rows, perm)
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
    if idx.numel() > np.iinfo(np.int32).max:        # beyond one device sort: the column-block kernels (implicit indptr)
        b = CscBuilder(None, _i32(idx), shape=(n_pre, n_post), row_len=int(idx.shape[1]))
        rows, _, perm = b.block(0, n_post, perm=True)
        return _finish(as_np, b.offsets(), rows, perm)
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
