"""Event containers: ``EventRepresentation`` and ``BinaryArray``.

Follows the reference's container + ``@`` dispatch (``brainevent/_event/base.py:75-359``,
``brainevent/_event/binary.py:31-321``): a dense right operand is multiplied here
(``binary_densemv`` / ``binary_densemm``); anything else is handed to the operand's
``__rmatmul__`` / ``__matmul__``.  Values are numpy arrays or torch tensors (no JAX pytree).
"""
import numpy as np
import torch

from ._error import MathError

__all__ = ['EventRepresentation', 'BinaryArray']


def _is_known_type(x) -> bool:
    return isinstance(x, (np.ndarray, torch.Tensor, np.generic, list, tuple, int, float, bool))


def _raw(x):
    if isinstance(x, (np.ndarray, torch.Tensor)):
        return x
    return np.asarray(x)


class EventRepresentation:
    """Array wrapper marking its payload as an event (spike) array."""
    __array_priority__ = 100
    __slots__ = ('_value',)

    def __init__(self, value):
        if isinstance(value, EventRepresentation):
            value = value.value
        if isinstance(value, (list, tuple)):
            value = np.asarray(value)
        self._value = value

    @property
    def value(self):
        return self._value

    @value.setter
    def value(self, v):
        if isinstance(v, EventRepresentation):
            v = v.value
        self._value = v

    def with_value(self, value):
        return type(self)(value)

    @property
    def dtype(self):
        return self._value.dtype

    @property
    def shape(self):
        return tuple(self._value.shape)

    @property
    def ndim(self):
        return self._value.ndim

    @property
    def size(self):
        return int(np.prod(self.shape)) if self.shape else 1

    @property
    def T(self):
        return self._value.T

    def transpose(self, *axes):
        if isinstance(self._value, torch.Tensor):
            return self._value.permute(*axes) if axes else self._value.T
        return self._value.transpose(*axes)

    def __len__(self):
        return len(self._value)

    def __getitem__(self, index):
        return self._value[index]

    def __repr__(self):
        return f"{type(self).__name__}(value={self._value!r})"

    def __array__(self, dtype=None):
        v = self._value
        if isinstance(v, torch.Tensor):
            v = v.cpu().numpy()
        return np.asarray(v, dtype=dtype)


class BinaryArray(EventRepresentation):
    """Binary (0/1) event vector or matrix with event-driven ``@``.

    An element is active when ``True`` / non-zero (bool and integer payloads) or ``> 0`` (float
    payloads).  ``s @ W`` for dense ``W[k, n]`` computes ``y[j] = sum_{i: s[i] active} W[i, j]``.
    """
    __slots__ = ()

    def _check_ndim(self):
        if self.ndim not in (1, 2):
            raise MathError(f"Matrix multiplication is only supported for 1D and 2D arrays. "
                            f"Got {self.ndim}D array.")

    def __matmul__(self, oc):
        if _is_known_type(oc):
            from ._dense import binary_densemv, binary_densemm
            oc = _raw(oc)
            self._check_ndim()
            assert oc.ndim == 2, (f"Right operand must be a 2D array in matrix multiplication. "
                                  f"Got {oc.ndim}D array.")
            assert self.shape[-1] == oc.shape[0], (f"Incompatible dimensions for matrix multiplication: "
                                                   f"{self.shape[-1]} and {oc.shape[0]}.")
            if self.ndim == 1:
                return binary_densemv(oc, self.value, transpose=True)
            return binary_densemm(oc, self.value.T, transpose=True).T
        return oc.__rmatmul__(self)

    def __rmatmul__(self, oc):
        if _is_known_type(oc):
            from ._dense import binary_densemv, binary_densemm
            oc = _raw(oc)
            self._check_ndim()
            assert oc.ndim == 2, (f"Left operand must be a 2D array in matrix multiplication. "
                                  f"Got {oc.ndim}D array.")
            assert oc.shape[-1] == self.shape[0], (f"Incompatible dimensions for matrix multiplication: "
                                                   f"{oc.shape[-1]} and {self.shape[0]}.")
            if self.ndim == 1:
                return binary_densemv(oc, self.value, transpose=False)
            return binary_densemm(oc, self.value, transpose=False)
        return oc.__matmul__(self)
