"""Fixed-number connectivity (ELL): ``FixedNumPerPre`` / ``FixedNumPerPost`` and ``binary_fcnmv`` / ``binary_fcnmm``.

Reference surface mirrored (read as text): ``brainevent/_fcn/main.py:199-460`` (``FixedNumConn`` dispatch),
``:781-854`` (``FixedNumPerPre``: ``indices (n_pre, n_conn)`` are post ids), ``:1042-1115`` (``FixedNumPerPost``:
``indices (n_post, n_conn)`` are pre ids), ``brainevent/_fcn/binary.py:43-143`` / ``:564-675`` (functional ops),
``:156-253`` / ``:677-766`` (CPU semantics), ``:450-509`` / ``:1077-1137`` (``*_p_call`` validation).

Semantics for ``indices[n_rows, n_conn]`` and ``shape = (n_rows, n_cols)`` as the functional ops see it:
  transpose=True  (scatter): ``out[indices[i, c]] += w[i, c]`` for every active ``i``      -> ``out[n_cols]``
  transpose=False (gather) : ``out[i] = sum_c w[i, c] * e(s[indices[i, c]])``              -> ``out[n_rows]``

An ELL matrix is a CSR matrix with an implicit ``indptr`` (``row r = [r*n_conn, (r+1)*n_conn)``), so the
kernels are the CSR ones (``csrc/be_csr.hip``) reached through the ``be_binary_fcn*`` symbols.  The
unfavourable direction (``FixedNumPerPre @ spk``, ``spk @ FixedNumPerPost``) runs event-driven through the CSC
mirror (reference ``_fcn/main.py:280-326``: ``_weight_indices`` + the perm-fused CSR kernel) — here a
:class:`brainevent_amd._csr.Mirror` built by the column-block kernels with the weights moved along, on first use for
matrices large enough for it to pay (``_csr.AUTO_MIRROR_MIN_NNZ``) or by ``prepare(mirror=True)`` — and the gather
kernel otherwise.
"""
import ctypes
from typing import Dict, Optional

import numpy as np
import torch

from . import _array as A
from ._data import DataRepresentation
from ._csr import ScatterPlan, BinnedScatter, _plan_call, binned_batch, _csrmm_generic
from . import _csr as _csr_mod
from ._event import BinaryArray, is_event, event_operand
from ._lib import check, fn
from ._misc import _as_int32_indices, check_fixed_conn_num_shape
from ._op import OpKernel

__all__ = ['FixedNumConn', 'FixedNumPerPre', 'FixedNumPerPost', 'binary_fcnmv', 'binary_fcnmm',
           'binary_fcnmv_p', 'binary_fcnmm_p', 'binary_fcnmv_p_call', 'binary_fcnmm_p_call']

c_i64, c_int, c_vp = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p
_FCN_MM_ARGS = [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp]


def _variant(homo: bool, w: torch.Tensor, sd: int) -> str:
    return f"{'homo' if homo else 'hetero'}_{A.wsuffix(w)}_{'bool' if sd == A.BE_SPIKE_BOOL else 'float'}"


def _fcn_batched(weights, indices, spikes_bm, sd, *, shape, transpose, workspace=None):
    """``spikes_bm [n_batch, len]`` -> ``[n_batch, out_len]`` through the ``be_binary_fcnmm_*`` symbols."""
    n_rows, n_conn = int(indices.shape[0]), int(indices.shape[1])
    n_cols = int(shape[1])
    nb = int(spikes_bm.shape[0])
    homo = weights.numel() == 1
    out_len = n_cols if transpose else n_rows
    out = torch.empty((nb, out_len), dtype=weights.dtype, device=weights.device)
    if out_len == 0 or nb == 0:
        return out
    if n_rows == 0 or n_cols == 0 or n_conn == 0:
        return out.zero_()
    if transpose:
        if isinstance(workspace, ScatterPlan):
            _plan_call(workspace, weights, spikes_bm, sd, out)
            return out
        if isinstance(workspace, BinnedScatter):
            binned_batch(workspace, weights, indices, None, n_conn, spikes_bm, sd, out)
            return out
        f_ws = fn('be_binary_csrmm_t_workspace_bytes', c_i64, [c_i64, c_i64, c_i64, c_int])
        ws = A.workspace(f_ws(n_rows, n_cols, nb, A.wcode(weights)))
        f = fn('be_binary_fcnmm_scatter_' + _variant(homo, weights, sd), c_int, _FCN_MM_ARGS)
    else:
        f_ws = fn('be_binary_csrmm_nt_workspace_bytes', c_i64, [c_i64, c_i64, c_i64])
        ws = A.workspace(f_ws(n_rows, n_cols, nb))
        f = fn('be_binary_fcnmm_gather_' + _variant(homo, weights, sd), c_int, _FCN_MM_ARGS)
    if sd >= A.BE_SPIKE_BITS:
        _csrmm_generic(weights, indices, None, n_conn, spikes_bm, sd, out, n_rows, n_cols, nb, ws, transpose)
        return out
    check(f(A.ptr(weights), A.ptr(indices), A.ptr(spikes_bm), A.ptr(out), n_rows, n_cols, n_conn, nb, A.ptr(ws),
            ws.numel(), A.stream_ptr()), f.__name__)
    return out


def _binary_fcnmv_hip(weights, indices, spikes, *, shape, transpose, workspace=None):
    s, sd = A.spikes_to_device(spikes)
    return _fcn_batched(weights, indices, s.reshape(1, -1), sd, shape=shape, transpose=transpose, workspace=workspace)[0]


def _binary_fcnmm_hip(weights, indices, matrix, *, shape, transpose, workspace=None):
    spikes_bm, sd = A.spikes_batch_major(matrix)
    return _fcn_batched(weights, indices, spikes_bm, sd, shape=shape, transpose=transpose, workspace=workspace).T


binary_fcnmv_p = OpKernel('binary_fcnmv')
binary_fcnmv_p.def_kernel('hip', 'gpu', _binary_fcnmv_hip, asdefault=True)
binary_fcnmv_p.def_tags('fcn', 'binary')
binary_fcnmm_p = OpKernel('binary_fcnmm')
binary_fcnmm_p.def_kernel('hip', 'gpu', _binary_fcnmm_hip, asdefault=True)
binary_fcnmm_p.def_tags('fcn', 'binary')


def binary_fcnmv_p_call(weights, indices, spikes, *, shape, transpose, backend=None, workspace=None):
    """Validation + dispatch (reference ``brainevent/_fcn/binary.py:450-509``).  Returns a 1-tuple."""
    check_fixed_conn_num_shape(weights, indices, spikes, shape, transpose)
    assert weights.dtype.is_floating_point, 'Weights must be a floating-point type.'
    weights = weights.reshape(1) if weights.numel() == 1 else weights
    return (binary_fcnmv_p(weights, indices, spikes, shape=shape, transpose=transpose, workspace=workspace,
                           backend=backend),)


binary_fcnmv_p.def_call(binary_fcnmv_p_call)


def binary_fcnmm_p_call(weights, indices, matrix, *, shape, transpose, backend=None, workspace=None):
    """Validation + dispatch of the matrix op (reference ``brainevent/_fcn/binary.py:1077-1137``)."""
    assert matrix.ndim == 2, "matrix must be 2D."
    check_fixed_conn_num_shape(weights, indices, matrix, shape, transpose)
    assert weights.dtype.is_floating_point, 'Weights must be a floating-point type.'
    weights = weights.reshape(1) if weights.numel() == 1 else weights
    return (binary_fcnmm_p(weights, indices, matrix, shape=shape, transpose=transpose, workspace=workspace,
                           backend=backend),)


binary_fcnmm_p.def_call(binary_fcnmm_p_call)


def _prep(weights, indices):
    w = A.to_device(weights)
    idx = A.to_device(indices)
    if idx.dtype != torch.int32:
        idx = _as_int32_indices(idx, None, 'binary_fcn', check_values=False)
    return w, idx


def binary_fcnmv(weights, indices, spikes, *, shape, transpose: bool = False, backend: Optional[str] = None):
    """Event-driven product with a fixed-number-connectivity matrix (reference ``_fcn/binary.py:43-143``).

    ``transpose=False``: ``y[i] = sum_c w[i,c] * e(s[indices[i,c]])`` (``y`` has ``shape[0]`` entries);
    ``transpose=True`` : ``y[indices[i,c]] += w[i,c]`` for active ``s[i]`` (``y`` has ``shape[1]`` entries).
    """
    as_np = A.wants_numpy(weights, indices, spikes)
    w, idx = _prep(weights, indices)
    s = spikes if isinstance(spikes, torch.Tensor) else np.asarray(spikes)
    r = binary_fcnmv_p_call(w, idx, s, shape=tuple(shape), transpose=transpose, backend=backend)[0]
    return A.to_result(r, as_np)


def binary_fcnmm(weights, indices, matrix, *, shape, transpose: bool = False, backend: Optional[str] = None):
    """Matrix-operand version (reference ``_fcn/binary.py:564-675``): ``matrix`` is ``(shape[1], n)`` for
    ``transpose=False`` -> ``(shape[0], n)``; ``(shape[0], n)`` for ``transpose=True`` -> ``(shape[1], n)``."""
    as_np = A.wants_numpy(weights, indices, matrix)
    w, idx = _prep(weights, indices)
    M = matrix if isinstance(matrix, torch.Tensor) else np.asarray(matrix)
    r = binary_fcnmm_p_call(w, idx, M, shape=tuple(shape), transpose=transpose, backend=backend)[0]
    return A.to_result(r, as_np)


# =====================================================================================================
# containers
# =====================================================================================================
def _validate_fixed_conn_indices(indices, *, expected_rows: int, kind: str):
    if indices.ndim != 2:
        raise ValueError(f'{kind} indices must be 2D, got {indices.ndim}D.')
    if indices.shape[0] != expected_rows:
        raise ValueError(f'{kind} row number mismatch. {indices.shape[0]} != {expected_rows}')
    if indices.dtype.is_floating_point or indices.dtype == torch.bool:
        raise ValueError(f'{kind} indices must be integer type, got {indices.dtype}.')


def _contains_invalid_indices(indices, *, upper_bound: int):
    if indices.numel() == 0:
        return
    lo, hi = int(indices.min()), int(indices.max())
    if lo < 0 or hi >= upper_bound:
        raise ValueError('Found invalid indices in the connection matrix. '
                         f'All indices must be in the range [0, {upper_bound - 1}]. '
                         f'But found indices with min {lo} and max {hi}.')


class FixedNumConn(DataRepresentation):
    """Base of the two ELL containers (reference ``_fcn/main.py:199-460``)."""

    def __init__(self, data, indices=None, *, shape, backend: Optional[str] = None, buffers: Optional[Dict] = None,
                 check_indices: bool = True):
        args = data if indices is None else (data, indices)
        assert len(args) == 2, "Expected two arguments: data, indices."
        self._numpy_result = A.wants_numpy(*args)
        self.data = A.to_device(args[0])
        idx = A.to_device(args[1])
        self.shape = (int(shape[0]), int(shape[1]))
        rows, upper = self._rows_and_upper()
        _validate_fixed_conn_indices(idx, expected_rows=rows, kind=self._kind)
        self.indices = _as_int32_indices(idx, upper, f'{type(self).__name__} indices', check_values=False)
        if self.data.numel() != 1 and tuple(self.data.shape) != tuple(self.indices.shape):
            raise ValueError(f"Data shape {tuple(self.data.shape)} must match indices shape "
                             f"{tuple(self.indices.shape)}. But got {tuple(self.data.shape)} != {tuple(self.indices.shape)}")
        self.backend = backend
        self._init_buffers(buffers)
        if check_indices:
            _contains_invalid_indices(self.indices, upper_bound=upper)

    # -- subclass hooks ---------------------------------------------------------------------------
    _kind = 'Connection'

    def _rows_and_upper(self):
        raise NotImplementedError

    @property
    def _a_shape(self):
        raise NotImplementedError

    def _ell_transpose(self, transpose_W: bool) -> bool:
        raise NotImplementedError

    # -- properties -------------------------------------------------------------------------------
    num_conn = property(lambda self: int(self.indices.shape[1]))
    nse = property(lambda self: int(self.indices.numel()))
    dtype = property(lambda self: self.data.dtype)
    ndim = property(lambda self: 2)

    # -- per-matrix workspace -----------------------------------------------------------------------
    def _scatter_workspace(self):
        if 'scatter_plan' in self.buffers:     # re-derived when ``self.data`` was modified in place since
            plan = self.buffers['scatter_plan'] = _csr_mod.fresh_scatter_workspace(self.buffers['scatter_plan'], self.data,
                                                                                   self.indices, None)
            return plan
        n_rows, n_cols = self._a_shape
        plan = None
        plan = _csr_mod.make_scatter_workspace(_csr_mod.choose_scatter_route(self.nse, n_rows, n_cols, self.data), self.data,
                                               self.indices, None, n_rows, n_cols, self.nse, row_len=self.num_conn)
        self.buffers['scatter_plan'] = plan
        return plan

    def prepare(self, mirror: bool = False):
        """Build the scatter workspace now; ``mirror=True`` also the CSC mirror of the unfavourable direction."""
        self._scatter_workspace()
        if mirror:
            self.build_mirror()
        return self

    def refresh_weights(self):
        """Bring the cached workspace up to date after ``self.data`` was modified in place (the products check it on every
        call; needed explicitly only between replays of a captured HIP graph)."""
        if 'scatter_plan' in self.buffers:
            self._scatter_workspace()
        if self.buffers.get('mirror') is not None:
            self._fresh_mirror()
        return self

    # -- CSC mirror of the unfavourable direction (reference ``_weight_indices``, ``_fcn/main.py:280-300``) ---------------
    def build_mirror(self, *, keep_raw: Optional[bool] = None, keep_perm: Optional[bool] = None):
        """The transposed structure with the weights moved along and a scatter workspace of its own
        (:func:`brainevent_amd._csr.build_mirror_of` over the implicit ``indptr``): afterwards ``FixedNumPerPre @ spk`` /
        ``spk @ FixedNumPerPost`` scatter over the active entries of ``spk`` instead of reading every stored row."""
        if self.buffers.get('mirror') is not None:
            return self.buffers['mirror']
        n_rows, n_cols = self._a_shape
        self.buffers['mirror'] = _csr_mod.build_mirror_of(self.data, self.indices, None, self.num_conn, n_rows, n_cols,
                                                          keep_raw=keep_raw, keep_perm=keep_perm)
        return self.buffers['mirror']

    def _fresh_mirror(self, auto: bool = False):
        mr = self.buffers.get('mirror')
        n_rows, n_cols = self._a_shape
        if mr is None:
            if not auto or 'mirror' in self.buffers:
                return None
            if not _csr_mod.auto_mirror_wanted(self.nse, n_rows, n_cols, self.data):
                self.buffers['mirror'] = None
                return None
            mr = self.build_mirror()

            import weakref
            owner = weakref.ref(self)

            def gather(s):      # the streaming gather over the fixed-length rows: the mirror's one-off cross-check
                c = owner()
                ref = binary_fcnmv_p_call(c.data, c.indices, s, shape=c._a_shape, transpose=False, backend=c.backend)[0]
                return ref, lambda: c.buffers.__setitem__('mirror', None)
            mr.check = gather
            return mr
        if mr.is_stale(self.data):
            mr = self.buffers['mirror'] = mr.refreshed(self.data, self.indices, None, self.num_conn, n_rows, n_cols)
        return mr

    # -- dispatch (reference ``_binary_matvec`` / ``_binary_matmat`` / ``_dispatch``) -----------------
    def _binary_matvec(self, s, transpose_W: bool):
        ell_t = self._ell_transpose(transpose_W)
        if not ell_t:
            mr = self._fresh_mirror(auto=True)
            if mr is not None:           # unfavourable direction, event-driven (reference ``_fcn/main.py:317-326``)
                check_fixed_conn_num_shape(self.data, self.indices, s, self._a_shape, False)
                return mr.apply(s, backend=self.backend)
        ws = self._scatter_workspace() if ell_t else None
        return binary_fcnmv_p_call(self.data, self.indices, s, shape=self._a_shape, transpose=ell_t,
                                   backend=self.backend, workspace=ws)[0]

    def _binary_matmat(self, matrix, transpose_W: bool):
        ell_t = self._ell_transpose(transpose_W)
        if not ell_t:
            mr = self._fresh_mirror(auto=True)
            if mr is not None:
                assert matrix.ndim == 2, "matrix must be 2D."
                check_fixed_conn_num_shape(self.data, self.indices, matrix, self._a_shape, False)
                return mr.apply(matrix, backend=self.backend)
        ws = self._scatter_workspace() if ell_t else None
        return binary_fcnmm_p_call(self.data, self.indices, matrix, shape=self._a_shape, transpose=ell_t,
                                   backend=self.backend, workspace=ws)[0]

    def _dispatch(self, other, transpose_W: bool):
        ell_t = self._ell_transpose(transpose_W)
        if not is_event(other):     # a dense operand: the float twins (reference ``_fcn/main.py:308-460`` dispatches them alike)
            from ._float import fcnmv_p_call, fcnmm_p_call
            x = other if isinstance(other, torch.Tensor) else np.asarray(other)
            # scatter direction: a gather over the mirror beats float atomics (built on first use only while it keeps its raw arrays)
            mr = self._fresh_mirror(auto=self.nse <= _csr_mod.MIRROR_KEEP_RAW_MAX_NNZ) if ell_t else None
            if mr is not None and (mr.released or mr.indices is None):
                mr = None
            if x.ndim not in (1, 2):
                raise NotImplementedError(f"matmul with object of shape {tuple(x.shape)}")
            if mr is not None:
                from ._float import csrmv_p_call, csrmm_p_call
                if x.ndim == 1:
                    r = csrmv_p_call(mr.data, mr.indices, mr.indptr, x, shape=tuple(mr.shape), transpose=False, backend=self.backend)[0]
                else:
                    r = csrmm_p_call(mr.data, mr.indices, mr.indptr, x.T if transpose_W else x, shape=tuple(mr.shape), transpose=False,
                                     backend=self.backend)[0]
                    r = r.T if transpose_W else r
            elif x.ndim == 1:
                r = fcnmv_p_call(self.data, self.indices, x, shape=self._a_shape, transpose=ell_t, backend=self.backend)[0]
            elif x.ndim == 2:
                r = fcnmm_p_call(self.data, self.indices, x.T if transpose_W else x, shape=self._a_shape, transpose=ell_t,
                                 backend=self.backend)[0]
                r = r.T if transpose_W else r
            else:
                raise NotImplementedError(f"matmul with object of shape {tuple(x.shape)}")
            return A.to_result(r, self._numpy_result) if A.wants_numpy(x) else r
        # scatter kernels take compacted id lists as they are — the favourable direction, and the other one once its mirror exists
        scatter = other.ndim == 1 and (ell_t or self._fresh_mirror(auto=True) is not None)
        value = event_operand(other, scatter=scatter)
        if value.ndim == 1:
            r = self._binary_matvec(value, transpose_W)
        elif value.ndim == 2:
            # binary_fcnmm returns (out_len, n) for an operand (in_len, n): ``events @ M`` hands it the transposed events and
            # transposes the result back.  (The orientation is fixed here, not guessed from the shapes: a square result —
            # batch size equal to the output length — would make such a guess ambiguous.)
            if transpose_W:
                expected = (value.shape[0], self.shape[1])
                r = self._binary_matmat(value.T, transpose_W).T
            else:
                expected = (self.shape[0], value.shape[1])
                r = self._binary_matmat(value, transpose_W)
            if tuple(r.shape) != tuple(expected):
                raise ValueError(f'binary matmat output shape mismatch: got {tuple(r.shape)}, expected {expected}.')
        else:
            raise NotImplementedError(f"matmul with object of shape {value.shape}")
        return A.to_result(r, self._numpy_result) if A.wants_numpy(value) else r

    def __matmul__(self, other):
        return self._dispatch(other, transpose_W=False)

    def __rmatmul__(self, other):
        return self._dispatch(other, transpose_W=True)

    # -- conversions (reference ``_fcn/main.py:857-897`` / ``:1118-1160`` fromdense, ``tocsr`` / ``tocsc``) ---------------
    @classmethod
    def fromdense(cls, mat, *, num_conn=None, backend=None):
        """Encode a dense ``(num_pre, num_post)`` matrix: ``num_conn`` connections per pre row (``FixedNumPerPre``) or per
        post column (``FixedNumPerPost``); explicit zeros are absent; short rows are padded with a zero-weight entry at
        index 0; ``num_conn=None`` requires a uniform count.  Host-side helper."""
        dense = mat.cpu().numpy() if isinstance(mat, torch.Tensor) else np.asarray(mat)
        if dense.ndim != 2:
            raise ValueError(f"{cls.__name__}.fromdense expects a 2-D matrix; got {dense.ndim}-D.")
        view = dense if cls is FixedNumPerPre or issubclass(cls, FixedNumPerPre) else dense.T
        mask = view != 0
        nnz = mask.sum(axis=1)
        if num_conn is None:
            if view.shape[0] == 0:
                num_conn = 0
            elif not bool((nnz == nnz[0]).all()):
                raise ValueError(f"{cls.__name__}.fromdense: rows have a non-uniform number of connections (min {int(nnz.min())}, "
                                 f"max {int(nnz.max())}). Pass num_conn= to pad to a fixed count, or use CSR.fromdense / "
                                 f"CSC.fromdense for an irregular matrix.")
            else:
                num_conn = int(nnz[0])
        num_conn = int(num_conn)
        if view.shape[0] and bool((nnz > num_conn).any()):
            raise ValueError(f"{cls.__name__}.fromdense: num_conn={num_conn} is too small; a row has {int(nnz.max())} connections.")
        data = np.zeros((view.shape[0], num_conn), dtype=dense.dtype)
        indices = np.zeros((view.shape[0], num_conn), dtype=np.int32)
        for r in range(view.shape[0]):
            cols = np.flatnonzero(mask[r])
            data[r, :cols.size] = view[r, cols]
            indices[r, :cols.size] = cols
        obj = cls((data, indices), shape=dense.shape, backend=backend)
        obj._numpy_result = not isinstance(mat, torch.Tensor)
        return obj

    def _as_compressed(self):
        """(data, indices, indptr) of the stored rows: row ``r`` holds ``num_conn`` entries."""
        n_rows, n_conn = int(self.indices.shape[0]), int(self.indices.shape[1])
        indptr = torch.arange(n_rows + 1, dtype=torch.int64, device=self.indices.device) * n_conn
        if indptr[-1].item() <= np.iinfo(np.int32).max:
            indptr = indptr.to(torch.int32)
        data = self.data if self.data.numel() == 1 else self.data.reshape(-1)
        return data, self.indices.reshape(-1), indptr

    def tocsr(self):
        """The same matrix as a :class:`CSR` (``FixedNumPerPre`` rows are CSR rows; ``FixedNumPerPost`` is re-encoded)."""
        from ._csr import CSR, CSC
        data, idx, ptr = self._as_compressed()
        native = (CSR if isinstance(self, FixedNumPerPre) else CSC)._from_parts(data, idx, ptr, shape=self.shape, backend=self.backend,
                                                                             numpy_result=self._numpy_result)
        return native.tocsr()

    def tocsc(self):
        """The same matrix as a :class:`CSC`."""
        from ._csr import CSR, CSC
        data, idx, ptr = self._as_compressed()
        native = (CSR if isinstance(self, FixedNumPerPre) else CSC)._from_parts(data, idx, ptr, shape=self.shape, backend=self.backend,
                                                                             numpy_result=self._numpy_result)
        return native.tocsc()

    def todense(self):
        idx = self.indices.cpu().numpy()
        w = (self.data.float() if self.data.dtype == torch.bfloat16 else self.data).cpu().numpy()
        vals = np.broadcast_to(w.reshape(-1), (idx.size,)) if w.size == 1 else w.reshape(-1)
        rows = np.repeat(np.arange(idx.shape[0]), idx.shape[1])
        n_rows, n_cols = self._a_shape
        dense = np.zeros((n_rows, n_cols), dtype=vals.dtype)
        np.add.at(dense, (rows, idx.reshape(-1)), vals)
        return dense if tuple(self._a_shape) == tuple(self.shape) else dense.T


class FixedNumPerPre(FixedNumConn):
    """Each pre-synaptic neuron has ``n_conn`` post targets: ``indices (n_pre, n_conn)`` hold post ids
    (reference ``_fcn/main.py:781-854``).  ``spk @ M`` is the favourable (scatter) direction."""
    _kind = 'Post-synaptic'

    def _rows_and_upper(self):
        return self.shape[0], self.shape[1]

    num_pre = property(lambda self: int(self.indices.shape[0]))
    num_post = property(lambda self: int(self.shape[1]))

    @property
    def _a_shape(self):
        return tuple(self.shape)

    def _ell_transpose(self, transpose_W: bool) -> bool:
        return bool(transpose_W)

    def with_data(self, data):
        data = A.to_device(data)
        assert data.shape == self.data.shape and data.dtype == self.data.dtype
        obj = FixedNumPerPre((data, self.indices), shape=self.shape, backend=self.backend, check_indices=False)
        obj._numpy_result = self._numpy_result
        return obj

    def transpose(self, axes=None):
        assert axes is None, "transpose does not support axes argument."
        obj = FixedNumPerPost((self.data, self.indices), shape=self.shape[::-1], backend=self.backend, check_indices=False)
        obj._numpy_result = self._numpy_result
        return obj

    T = property(lambda self: self.transpose())


class FixedNumPerPost(FixedNumConn):
    """Each post-synaptic neuron has ``n_conn`` pre sources: ``indices (n_post, n_conn)`` hold pre ids
    (reference ``_fcn/main.py:1042-1115``).  ``M @ spk`` is the favourable (scatter) direction."""
    _kind = 'Pre-synaptic'

    def _rows_and_upper(self):
        return self.shape[1], self.shape[0]

    num_post = property(lambda self: int(self.indices.shape[0]))
    num_pre = property(lambda self: int(self.shape[0]))

    @property
    def _a_shape(self):
        return tuple(self.shape)[::-1]

    def _ell_transpose(self, transpose_W: bool) -> bool:
        return not bool(transpose_W)

    def with_data(self, data):
        data = A.to_device(data)
        assert data.shape == self.data.shape and data.dtype == self.data.dtype
        obj = FixedNumPerPost((data, self.indices), shape=self.shape, backend=self.backend, check_indices=False)
        obj._numpy_result = self._numpy_result
        return obj

    def transpose(self, axes=None):
        assert axes is None, "transpose does not support axes argument."
        obj = FixedNumPerPre((self.data, self.indices), shape=self.shape[::-1], backend=self.backend, check_indices=False)
        obj._numpy_result = self._numpy_result
        return obj

    T = property(lambda self: self.transpose())
