"""Index dtype policy and structure validators for the hot path.

Restates the contracts of the reference's ``brainevent/_misc.py``:
``_resolve_indptr_dtype`` (:213-241), ``_as_int32_indices`` (:271-316), ``_as_indptr`` (:316-335),
``_check_compressed_structure`` (:335-378), ``check_fixed_conn_num_shape`` (:697-869),
``_normalize_chunk_size`` (:74-122) and the walk strides ``_MV_STRIDE`` / ``_MM_STRIDE`` (:37-38).
Validation that needs the values runs on the device (torch reductions) and reads back scalars only.
"""
from typing import Optional, Tuple

import numpy as np
import torch

_INT32_MAX = int(np.iinfo(np.int32).max)

#: residue-class stride of the light-RNG walk: part of the *drawn matrix* (mv and mm differ).
_MV_STRIDE = 32
_MM_STRIDE = 4


def _normalize_chunk_size(n_cols, chunk_size=None, target_chunks: int = 4) -> int:
    if chunk_size is None:
        target_chunks = int(target_chunks)
        if target_chunks <= 0:
            raise ValueError("target_chunks must be positive")
        chunk_size = max(1, (int(n_cols) + target_chunks - 1) // target_chunks)
    chunk_size = int(chunk_size)
    if chunk_size <= 0:
        raise ValueError("chunk_size must be positive")
    return chunk_size


def _is_int_dtype(dt: torch.dtype) -> bool:
    return dt in (torch.int8, torch.int16, torch.int32, torch.int64, torch.uint8)


def _resolve_indptr_dtype(nse: int, requested="auto") -> torch.dtype:
    nse = int(nse)
    if isinstance(requested, str):
        if requested != "auto":
            raise ValueError(f"indptr_dtype must be 'auto', int32, or int64; got {requested!r}.")
        return torch.int64 if nse > _INT32_MAX else torch.int32
    dt = np.dtype(requested) if not isinstance(requested, torch.dtype) else None
    if requested is torch.int32 or (dt is not None and dt == np.dtype(np.int32)):
        if nse > _INT32_MAX:
            raise OverflowError(f"nnz={nse} exceeds the int32 range ({_INT32_MAX}); request "
                                "indptr_dtype='auto' or int64.")
        return torch.int32
    if requested is torch.int64 or (dt is not None and dt == np.dtype(np.int64)):
        return torch.int64
    raise ValueError(f"indptr_dtype must be 'auto', int32, or int64; got {requested!r}.")


def _as_int32_indices(indices: torch.Tensor, secondary_dim: Optional[int], context: str,
                      check_values: bool = True) -> torch.Tensor:
    if not _is_int_dtype(indices.dtype):
        raise TypeError(f"{context}: indices must be an integer array; got dtype {indices.dtype}.")
    if secondary_dim is not None and int(secondary_dim) > _INT32_MAX + 1:
        raise OverflowError(f"{context}: secondary dimension {int(secondary_dim)} exceeds the "
                            "int32-representable coordinate range.")
    if check_values and indices.numel():
        min_v = int(indices.min())
        max_v = int(indices.max())
        if min_v < 0:
            raise ValueError(f"{context}: indices must be non-negative; got minimum {min_v}.")
        if secondary_dim is not None and max_v >= int(secondary_dim):
            raise ValueError(f"{context}: index {max_v} is out of bounds for secondary "
                             f"dimension {int(secondary_dim)}.")
        if max_v > _INT32_MAX:
            raise OverflowError(f"{context}: index {max_v} exceeds the int32 range; "
                                "secondary-axis coordinates must fit int32.")
    return indices if indices.dtype == torch.int32 else indices.to(torch.int32)


def _as_indptr(indptr: torch.Tensor, nse: int, indptr_dtype, context: str) -> torch.Tensor:
    if not _is_int_dtype(indptr.dtype):
        raise TypeError(f"{context}: indptr must be an integer array; got dtype {indptr.dtype}.")
    target = _resolve_indptr_dtype(nse, indptr_dtype)
    return indptr if indptr.dtype == target else indptr.to(target)


def _check_compressed_structure(indices: torch.Tensor, indptr: torch.Tensor, shape, format: str = "csr",
                                check_values: bool = True) -> None:
    fmt = format.lower()
    if fmt not in ("csr", "csc"):
        raise ValueError(f"format must be 'csr' or 'csc'; got {format!r}.")
    primary_dim = shape[0] if fmt == "csr" else shape[1]
    if indices.ndim != 1:
        raise ValueError(f"{fmt} indices must be a 1D array; got ndim {indices.ndim}.")
    if indptr.ndim != 1:
        raise ValueError(f"{fmt} indptr must be a 1D array; got ndim {indptr.ndim}.")
    if indices.dtype != torch.int32:
        raise TypeError(f"{fmt} indices must be int32; got {indices.dtype}.")
    if indptr.dtype not in (torch.int32, torch.int64):
        raise TypeError(f"{fmt} indptr must be int32 or int64; got {indptr.dtype}.")
    if indptr.shape[0] != int(primary_dim) + 1:
        raise ValueError(f"{fmt} indptr length must be primary dimension + 1 "
                         f"({int(primary_dim) + 1}); got {indptr.shape[0]}.")
    if not check_values:
        return
    if indptr.numel() and int(indptr[0]) != 0:
        raise ValueError(f"{fmt} indptr[0] must be 0; got {int(indptr[0])}.")
    if indptr.numel() > 1 and bool((indptr[1:] < indptr[:-1]).any()):
        raise ValueError(f"{fmt} indptr must be monotonically non-decreasing.")
    if int(indptr[-1]) != int(indices.shape[0]):
        raise ValueError(f"{fmt} indptr[-1] ({int(indptr[-1])}) must equal the number of "
                         f"stored elements ({int(indices.shape[0])}).")


def check_fixed_conn_num_shape(weights, indices, vector, shape: Tuple[int, int], transpose: bool,
                               require_scalar_weight: bool = False):
    """Shape contract of the fixed-number-connectivity ops (reference ``_misc.py:697-869``).

    ``indices`` is ``(n_pre, n_conn)``; ``weights`` has the same shape or one element; the
    vector has ``shape[0]`` entries when ``transpose`` (scatter) and ``shape[1]`` otherwise.
    Returns ``(out_len, n_pre, n_post)``.
    """
    if indices.ndim != 2:
        raise ValueError(f"indices must be 2D (n_pre, n_conn); got ndim {indices.ndim}.")
    n_pre, n_post = int(shape[0]), int(shape[1])
    assert indices.shape[0] == n_pre, (
        f"Pre size mismatch, got {indices.shape[0]} != {n_pre}")
    if weights.numel() != 1:
        assert tuple(weights.shape) == tuple(indices.shape), (
            f"The shape of weights {tuple(weights.shape)} and indices {tuple(indices.shape)} should be the same.")
    elif require_scalar_weight:
        pass
    if vector is not None:
        if transpose:
            assert vector.shape[0] == n_pre, f"vector length {vector.shape[0]} != shape[0] {n_pre}"
        else:
            assert vector.shape[0] == n_post, f"vector length {vector.shape[0]} != shape[1] {n_post}"
    return (n_post if transpose else n_pre), n_pre, n_post
