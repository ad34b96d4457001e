"""Event-driven dense products: ``binary_densemv`` / ``binary_densemm`` (the ``BinaryArray @ ndarray`` path).

Reference surface (read as text): ``brainevent/_dense/binary.py:79-165`` (``binary_densemv``), ``:347-432``
(``binary_densemv_p_call``), ``:487-576`` (``binary_densemm``), ``:818-912`` (``binary_densemm_p_call``),
``:168-211`` / ``:579-632`` (CPU semantics).

  transpose=False: ``weights[m, k] @ spikes[k]      -> out[m]``     /  ``weights[m, k] @ spikes[k, n] -> out[m, n]``
  transpose=True : ``spikes[k] @ weights[k, n]      -> out[n]``     /  ``weights[k, m].T @ spikes[k, n] -> out[m, n]``

Non-bool, non-float spikes are cast to bool first (reference ``:162-163``); float spikes are active when
``> 0`` (the documented binary semantics; the reference's ``jax_raw`` multiply-by-value shortcut is not followed).
"""
import ctypes
from typing import Optional

import numpy as np
import torch

from . import _array as A
from ._data import DataRepresentation
from ._lib import check, fn
from ._op import OpKernel

__all__ = ['Dense', 'binary_densemv', 'binary_densemm', 'binary_densemv_p', 'binary_densemm_p', 'binary_densemv_p_call',
           'binary_densemm_p_call']

c_i64, c_int, c_vp = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p
_MM_ARGS = [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp]


def _dense_batched(weights: torch.Tensor, spikes_bm: torch.Tensor, sd: int, transpose: bool) -> torch.Tensor:
    """``spikes_bm [nb, k]`` -> ``[nb, out_len]`` through ``be_binary_densemm_{transpose,no_transpose}_*``."""
    rows_w, cols_w = int(weights.shape[0]), int(weights.shape[1])
    nb = int(spikes_bm.shape[0])
    out_len = cols_w if transpose else rows_w
    out = torch.empty((nb, out_len), dtype=weights.dtype, device=weights.device)
    if out_len == 0 or nb == 0:
        return out
    f_ws = fn('be_binary_densemm_workspace_bytes', c_i64, [c_i64, c_i64, c_i64, c_int, c_int])
    ws = A.workspace(f_ws(rows_w, cols_w, nb, int(transpose), A.wcode(weights)))
    if sd == A.BE_SPIKE_BITS:      # bit-packed rows [nb, ceil(k / 32)] words: the generic entry point (the per-variant names cover bool / float)
        f = fn('be_binary_densemm', c_int, [c_vp, c_int, c_vp, c_int, c_vp, c_i64, c_i64, c_i64, c_int, c_vp, c_i64, c_vp])
        check(f(A.ptr(weights), A.wcode(weights), A.ptr(spikes_bm), sd, A.ptr(out), rows_w, cols_w, nb, int(transpose), A.ptr(ws),
                ws.numel(), A.stream_ptr()), 'be_binary_densemm')
        return out
    name = (f"be_binary_densemm_{'transpose' if transpose else 'no_transpose'}_{A.wsuffix(weights)}_"
            f"{'bool' if sd == A.BE_SPIKE_BOOL else 'float'}")
    f = fn(name, c_int, _MM_ARGS)
    check(f(A.ptr(weights), A.ptr(spikes_bm), A.ptr(out), rows_w, cols_w, nb, A.ptr(ws), ws.numel(), A.stream_ptr()), name)
    return out


def _binary_densemv_hip(weights, spikes, *, transpose):
    w = A.to_device(weights)
    s, sd = A.spikes_to_device(spikes)
    return _dense_batched(w, s.reshape(1, -1), sd, transpose)[0]


def _binary_densemm_hip(weights, spikes, *, transpose):
    w = A.to_device(weights)
    spikes_bm, sd = A.spikes_batch_major(spikes)
    return _dense_batched(w, spikes_bm, sd, transpose).T


binary_densemv_p = OpKernel('binary_densemv')
binary_densemv_p.def_kernel('hip', 'gpu', _binary_densemv_hip, asdefault=True)
binary_densemv_p.def_tags('dense', 'binary')
binary_densemm_p = OpKernel('binary_densemm')
binary_densemm_p.def_kernel('hip', 'gpu', _binary_densemm_hip, asdefault=True)
binary_densemm_p.def_tags('dense', 'binary')


def binary_densemv_p_call(weights, spikes, *, transpose, backend=None):
    assert weights.ndim == 2 and spikes.ndim == 1, "weights must be 2D and spikes 1D."
    if transpose:
        assert spikes.shape[0] == weights.shape[0], (
            f"shapes {tuple(spikes.shape)} and {tuple(weights.shape)} not aligned: "
            f"{spikes.shape[0]} (dim 0) != {weights.shape[0]} (dim 0)")
    else:
        assert spikes.shape[0] == weights.shape[1], (
            f"spikes shape {tuple(spikes.shape)} and weights shape {tuple(weights.shape)} are not compatible")
    return [binary_densemv_p(weights, spikes, transpose=transpose, backend=backend)]


binary_densemv_p.def_call(binary_densemv_p_call)


def binary_densemm_p_call(weights, spikes, *, transpose, backend=None):
    assert weights.ndim == 2 and spikes.ndim == 2, "weights and spikes must be 2D."
    if transpose:
        assert weights.shape[0] == spikes.shape[0], (
            f"weights shape {tuple(weights.shape)} and spikes shape {tuple(spikes.shape)} do not match for event matrix "
            f"multiplication: weights dim 0 ({weights.shape[0]}) != spikes dim 0 ({spikes.shape[0]})")
    else:
        assert weights.shape[1] == spikes.shape[0], (
            f"weights.shape[1] ({weights.shape[1]}) != spikes.shape[0] ({spikes.shape[0]}), "
            f"weights: {tuple(weights.shape)}, spikes: {tuple(spikes.shape)}")
    return [binary_densemm_p(weights, spikes, transpose=transpose, backend=backend)]


binary_densemm_p.def_call(binary_densemm_p_call)


def _as_arr(x):
    return x if isinstance(x, (torch.Tensor, A.PackedSpikes)) else np.asarray(x)


def _float_weights(w):
    dt = w.dtype
    ok = dt.is_floating_point if isinstance(w, torch.Tensor) else np.issubdtype(dt, np.floating)
    assert ok, 'Weights must be a floating-point type.'


def binary_densemv(weights, spikes, *, transpose, backend: Optional[str] = None):
    """``weights[m,k] @ spikes[k]`` (``transpose=False``) or ``spikes[k] @ weights[k,n]`` (``transpose=True``)."""
    as_np = A.wants_numpy(weights, spikes)
    w, s = _as_arr(weights), _as_arr(spikes)
    _float_weights(w)
    return A.to_result(binary_densemv_p_call(w, s, transpose=transpose, backend=backend)[0], as_np)


def binary_densemm(weights, spikes, *, transpose, backend: Optional[str] = None):
    """``weights[m,k] @ spikes[k,n]`` (``transpose=False``) or ``weights[k,m].T @ spikes[k,n]`` (``transpose=True``)."""
    as_np = A.wants_numpy(weights, spikes)
    w, s = _as_arr(weights), _as_arr(spikes)
    _float_weights(w)
    return A.to_result(binary_densemm_p_call(w, s, transpose=transpose, backend=backend)[0], as_np)


class Dense(DataRepresentation):
    """Explicit dense matrix with the representation contract of the sparse families (reference
    ``brainevent/_dense/main.py:60-490``, minus units, pytree plumbing and plasticity): ``data``, ``shape``,
    ``backend``, ``buffers``, ``with_data``, ``todense``, ``T`` / ``transpose`` and event-driven ``@``.

    ``Dense @ events`` -> ``binary_densemv/mm(transpose=False)``; ``events @ Dense`` -> ``transpose=True``
    (``_dense/main.py:426-476``).  Non-event operands are outside the accelerated path."""

    def __init__(self, data, shape=None, backend: Optional[str] = None, buffers: Optional[dict] = None):
        self._numpy_result = not isinstance(data, torch.Tensor)
        self.data = A.to_device(data)
        if self.data.ndim != 2:
            raise ValueError(f"Dense data must be two-dimensional, got {self.data.ndim}D.")
        if shape is not None and tuple(shape) != tuple(self.data.shape):
            raise ValueError(f"shape {tuple(shape)} does not match the data shape {tuple(self.data.shape)}.")
        self.shape = (int(self.data.shape[0]), int(self.data.shape[1]))
        self.backend = backend
        self._init_buffers(buffers)

    dtype = property(lambda self: self.data.dtype)
    ndim = property(lambda self: 2)

    def with_data(self, data) -> 'Dense':
        d = A.to_device(data)
        assert tuple(d.shape) == tuple(self.data.shape) and d.dtype == self.data.dtype
        out = Dense(d, shape=self.shape, backend=self.backend, buffers=self.buffers)
        out._numpy_result = self._numpy_result
        return out

    def todense(self):
        return A.to_result(self.data, self._numpy_result)

    def transpose(self, axes=None) -> 'Dense':
        assert axes is None, f"axes must be None, got {axes}."
        out = Dense(self.data.T.contiguous(), shape=self.shape[::-1], backend=self.backend, buffers=self.buffers)
        out._numpy_result = self._numpy_result
        return out

    T = property(lambda self: self.transpose())

    def __getitem__(self, index):
        return self.todense()[index]

    def _event(self, other):
        from ._event import is_event, event_operand
        if not is_event(other):
            raise NotImplementedError("only event operands are on the accelerated path (plain dense matmul is out of scope).")
        return event_operand(other)        # (1-D bit-packed containers hand their words over: BE_SPIKE_BITS)

    def __matmul__(self, other):          # dense @ events
        ev = self._event(other)
        if ev.ndim == 1:
            r = binary_densemv_p_call(self.data, ev, transpose=False, backend=self.backend)[0]
        elif ev.ndim == 2:
            r = binary_densemm_p_call(self.data, ev, transpose=False, backend=self.backend)[0]
        else:
            raise NotImplementedError(f"matmul with object of shape {ev.shape}")
        return A.to_result(r, self._numpy_result and A.wants_numpy(ev))

    def __rmatmul__(self, other):         # events @ dense
        ev = self._event(other)
        if ev.ndim == 1:
            r = binary_densemv_p_call(self.data, ev, transpose=True, backend=self.backend)[0]
        elif ev.ndim == 2:
            r = binary_densemm_p_call(self.data, ev.T, transpose=True, backend=self.backend)[0].T
        else:
            raise NotImplementedError(f"matmul with object of shape {ev.shape}")
        return A.to_result(r, self._numpy_result and A.wants_numpy(ev))

    def __repr__(self):
        return f"Dense(shape={self.shape}, dtype={self.dtype}, backend={self.backend})"
