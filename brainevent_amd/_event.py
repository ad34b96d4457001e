"""Event containers: ``EventRepresentation`` and ``BinaryArray``.

Follows the reference's container + ``@`` dispatch (``brainevent/_event/base.py:75-359``,
``brainevent/_event/binary.py:31-321``): a dense right operand is multiplied here
(``binary_densemv`` / ``binary_densemm``); anything else is handed to the operand's
``__rmatmul__`` / ``__matmul__``.  Values are numpy arrays or torch tensors (no JAX pytree).
"""
import numpy as np
import torch

from ._error import MathError

__all__ = ['EventRepresentation', 'BinaryArray', 'BitPackedBinary', 'CompactBinary', 'bitpack']


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
        if type(self) is EventRepresentation:      # abstract in the reference too (``_event/base_test.py:26-28``)
            raise TypeError("EventRepresentation is abstract; use BinaryArray (or another concrete event type).")
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

    def bitpack(self):
        """Bit-packed twin of this array (reference ``brainevent/_event/binary.py:81-109``)."""
        return BitPackedBinary(self.value)

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
            if self.ndim == 1:        # (a 1-D bit-packed / compacted container hands its words over: BE_SPIKE_BITS, no unpack)
                return binary_densemv(oc, event_operand(self), transpose=True)
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
                return binary_densemv(oc, event_operand(self), transpose=False)
            return binary_densemm(oc, self.value, transpose=False)
        return oc.__matmul__(self)


# =====================================================================================================
# bit-packed and compacted encodings (reference brainevent/_event/bitpack_binary.py, compact_binary.py).
# In the reference these are containers only (no kernel on this path reads them); here the scatter and
# gather kernels consume the packed words directly (C ABI spike code BE_SPIKE_BITS) — it is also what the
# multi-GPU spike exchange carries.
# =====================================================================================================
def _nonzero_mask(t: torch.Tensor) -> torch.Tensor:
    """The encoders' activity rule is "non-zero" (reference ``bitpack_binary.py:52``, ``compact.py:81``, ``:115``); the
    matmul ops' rule for float payloads is ``> 0``.  The two agree for bool / integer / non-negative payloads, and the
    containers below hand packed words to the kernels only for non-float payloads."""
    return t if t.dtype == torch.bool else (t != 0)


def _is_float_payload(v) -> bool:
    dt = v.dtype
    return dt.is_floating_point if isinstance(dt, torch.dtype) else np.issubdtype(dt, np.floating)


def bitpack(arr, axis: int):
    """Pack a binary array into uint32 words along ``axis``: bit ``b`` of word ``w`` is element ``32 w + b``
    (reference ``brainevent/_event/bitpack_binary.py:32-75``; non-zero values are ``True``).  numpy in -> numpy ``uint32`` out; device tensor in -> ``int32`` tensor holding the words."""
    import ctypes
    from . import _array as A
    from ._lib import fn, check
    as_np = not isinstance(arr, torch.Tensor)
    t = A.to_device(arr)
    axis = axis % t.ndim
    n = int(t.shape[axis])
    nw = (n + 31) // 32
    moved = t.movedim(axis, -1).contiguous()
    lead = tuple(moved.shape[:-1])
    rows = int(np.prod(lead)) if lead else 1
    words = torch.zeros((rows, nw), dtype=torch.int32, device=t.device)
    if rows and n:
        if rows <= 65535:
            sp, sd = _nonzero_mask(moved.reshape(rows, n)).contiguous(), A.BE_SPIKE_BOOL
            f = fn('be_pack_spikes_batched', ctypes.c_int,
                   [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p])
            check(f(A.ptr(sp.contiguous()), sd, n, rows, A.ptr(words), A.stream_ptr()), 'be_pack_spikes_batched')
        else:   # very many short rows (e.g. packing the batch axis of an (n, batch) matrix): plain tensor arithmetic
            m = _nonzero_mask(moved.reshape(rows, n))
            m = torch.nn.functional.pad(m, (0, nw * 32 - n)).view(rows, nw, 32).to(torch.int64)
            sh = torch.arange(32, device=t.device, dtype=torch.int64)
            v = (m << sh).sum(dim=-1)
            words = torch.where(v >= 2 ** 31, v - 2 ** 32, v).to(torch.int32)
    out = words.view(*lead, nw).movedim(-1, axis)
    if as_np:
        return np.ascontiguousarray(out.cpu().numpy()).view(np.uint32)
    return out


class BitPackedBinary(EventRepresentation):
    """Binary events kept both as the original array (``value``) and bit-packed along every axis (``packed``,
    computed on first use).  Reference: ``brainevent/_event/bitpack_binary.py:78-339``.

    ``BitPackedBinary.from_packed(words, n)`` wraps a 1-D vector that exists only as packed words (what the
    multi-GPU spike exchange delivers); its ``value`` is unpacked on demand."""
    __slots__ = ('_packed', '_original_shape')

    def __init__(self, arr):
        super().__init__(arr)
        self._original_shape = tuple(self._value.shape)
        self._packed = [None] * len(self._original_shape)

    @classmethod
    def from_packed(cls, words: torch.Tensor, n: int):
        obj = object.__new__(cls)
        obj._value = None
        obj._original_shape = (int(n),)
        assert words.ndim == 1 and words.numel() >= (int(n) + 31) // 32
        obj._packed = [words if words.dtype == torch.int32 else words.view(torch.int32)]
        return obj

    @property
    def value(self):
        if self._value is None:           # packed-only vector: unpack once
            import ctypes
            from . import _array as A
            from ._lib import fn, check
            n = self._original_shape[0]
            out = torch.empty(n, dtype=torch.bool, device=self._packed[0].device)
            f = fn('be_unpack_spikes', ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p])
            check(f(A.ptr(self._packed[0]), n, A.ptr(out), A.stream_ptr()), 'be_unpack_spikes')
            self._value = out
        return self._value

    @property
    def packed(self):
        """Tuple with one packed array per axis (``packed[i]`` packs along axis ``i``)."""
        for ax in range(len(self._packed)):
            if self._packed[ax] is None:
                self._packed[ax] = bitpack(self._value, ax)
        return tuple(self._packed)

    @property
    def original_shape(self):
        return self._original_shape

    shape = property(lambda self: self._original_shape)
    ndim = property(lambda self: len(self._original_shape))
    dtype = property(lambda self: self.value.dtype)

    def _packed_operand(self):
        """1-D vectors are handed to the kernels as packed words (``None`` for other ranks)."""
        from . import _array as A
        if self.ndim != 1 or (self._value is not None and _is_float_payload(self._value)):
            return None
        as_np = self._value is not None and not isinstance(self._value, torch.Tensor)
        if self._packed[0] is None:
            self._packed[0] = bitpack(self._value, 0)
        w = self._packed[0]
        if not isinstance(w, torch.Tensor):
            w = A.to_device(np.ascontiguousarray(w).view(np.int32))
        return A.PackedSpikes(w, self._original_shape[0], numpy_result=as_np)

    __matmul__ = BinaryArray.__matmul__
    __rmatmul__ = BinaryArray.__rmatmul__
    _check_ndim = BinaryArray._check_ndim

    def dot(self, oc):
        return self.__matmul__(oc)

    @property
    def T(self):
        return self.transpose()

    def transpose(self, *axes):
        if not axes:
            perm = tuple(reversed(range(self.ndim)))
        elif len(axes) == 1 and isinstance(axes[0], (tuple, list)):
            perm = tuple(axes[0])
        else:
            perm = tuple(axes)
        v = self.value
        obj = object.__new__(BitPackedBinary)
        obj._value = v.permute(*perm) if isinstance(v, torch.Tensor) else np.transpose(v, perm)
        obj._original_shape = tuple(self._original_shape[i] for i in perm)
        # new packed[i] packs along new axis i, which was old axis perm[i]
        old = self._packed
        obj._packed = [None if old[perm[i]] is None else
                       (old[perm[i]].permute(*perm) if isinstance(old[perm[i]], torch.Tensor) else np.transpose(old[perm[i]], perm))
                       for i in range(self.ndim)]
        return obj


class CompactBinary:
    """Bit-packed words plus the compacted list of active positions (reference
    ``brainevent/_event/compact_binary.py:53-430``).

    1-D ``(n,)``: ``packed (ceil(n/32),)``, ``active_ids (n,)`` int32 whose first ``n_active[0]`` entries are the
    active positions (in no particular order).  2-D ``(n, batch)``: ``packed (n, ceil(batch/32))`` packs the batch
    axis and ``active_ids`` lists the rows active in any batch column."""
    __slots__ = ('_packed', '_active_ids', '_n_active', '_value', '_n_orig', '_batch_size', '_bit_width', '_own_ids')
    __array_priority__ = 100

    def __init__(self, packed, active_ids, n_active, value, n_orig, batch_size=None, bit_width=32):
        self._packed, self._active_ids, self._n_active, self._value = packed, active_ids, n_active, value
        self._n_orig, self._batch_size, self._bit_width = n_orig, batch_size, bit_width
        self._own_ids = False      # True when active_ids came out of this library's compaction (in range, listed once)

    @staticmethod
    def _compact(mask_source):
        """device spikes [n] -> (active_ids int32 [n], n_active int32 [1]) through ``be_compact_spikes``."""
        import ctypes
        from . import _array as A
        from ._lib import fn, check
        sp, sd = _nonzero_mask(A.to_device(mask_source)).contiguous(), A.BE_SPIKE_BOOL
        n = int(sp.shape[0])
        ids = torch.zeros(n, dtype=torch.int32, device=sp.device)
        cnt = torch.zeros(1, dtype=torch.int32, device=sp.device)
        if n:
            f = fn('be_compact_spikes', ctypes.c_int,
                   [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p])
            check(f(A.ptr(sp), sd, n, A.ptr(ids), A.ptr(cnt), A.stream_ptr()), 'be_compact_spikes')
        return ids, cnt

    @classmethod
    def _build(cls, x, compact_1d: bool, pack_1d: bool = True):
        from . import _array as A
        as_np = not isinstance(x, torch.Tensor)
        t = A.to_device(x)
        conv = (lambda a: a.cpu().numpy()) if as_np else (lambda a: a)
        if t.ndim == 1:
            n = int(t.shape[0])
            packed = bitpack(t, 0) if pack_1d else torch.zeros(0, dtype=torch.int32, device=t.device)
            if compact_1d:
                ids, cnt = cls._compact(t)
            else:
                ids = torch.zeros(n, dtype=torch.int32, device=t.device)
                cnt = torch.zeros(1, dtype=torch.int32, device=t.device)
            pk = conv(packed)
            obj = cls(pk.view(np.uint32) if as_np else pk, conv(ids), conv(cnt), x if not as_np else np.asarray(x),
                      n_orig=n, batch_size=None, bit_width=32)
            obj._own_ids = compact_1d and not as_np
            return obj
        if t.ndim == 2:
            packed = bitpack(t, 1)
            ids, cnt = cls._compact(_nonzero_mask(t).any(dim=1))
            pk = conv(packed)
            return cls(pk.view(np.uint32) if as_np else pk, conv(ids), conv(cnt), x if not as_np else np.asarray(x),
                       n_orig=int(t.shape[0]), batch_size=int(t.shape[1]), bit_width=32)
        raise ValueError(f"CompactBinary only supports 1D and 2D arrays, got {t.ndim}D.")

    @classmethod
    def from_array(cls, x, bit_width=32):
        if bit_width != 32:
            raise ValueError(f"Only bit_width=32 is supported, got {bit_width}.")
        return cls._build(x, compact_1d=True)

    @classmethod
    def from_array_light(cls, x, bit_width=32):
        """Like :meth:`from_array` but a 1-D input skips the compaction (zeros in ``active_ids`` / ``n_active``)."""
        if bit_width != 32:
            raise ValueError(f"Only bit_width=32 is supported, got {bit_width}.")
        return cls._build(x, compact_1d=False)

    @classmethod
    def from_packed(cls, packed, active_ids, n_active, value, n_orig, batch_size=None, bit_width=32):
        return cls(packed, active_ids, n_active, value, n_orig=n_orig, batch_size=batch_size, bit_width=bit_width)

    @classmethod
    def compacy_only_vector(cls, x):      # (sic) the reference spells it this way, compact_binary.py:230
        """1-D compact-only encoding: ``packed`` is a zero-length sentinel."""
        nd = x.ndim if hasattr(x, 'ndim') else np.asarray(x).ndim
        if nd != 1:
            raise ValueError(f"CompactBinary.compacy_only_vector only supports 1D arrays, got {nd}D.")
        return cls._build(x, compact_1d=True, pack_1d=False)

    packed = property(lambda self: self._packed)
    active_ids = property(lambda self: self._active_ids)
    n_active = property(lambda self: self._n_active)
    value = property(lambda self: self._value)
    n_orig = property(lambda self: self._n_orig)
    batch_size = property(lambda self: self._batch_size)
    bit_width = property(lambda self: self._bit_width)
    dtype = property(lambda self: self._value.dtype)

    @property
    def shape(self):
        return (self._n_orig,) if self._batch_size is None else (self._n_orig, self._batch_size)

    @property
    def ndim(self):
        return 1 if self._batch_size is None else 2

    @property
    def size(self):
        return self._n_orig if self._batch_size is None else self._n_orig * self._batch_size

    def to_dense(self):
        return self._value

    def _packed_operand(self):
        from . import _array as A
        if self.ndim != 1 or self._packed is None or len(self._packed) == 0 or _is_float_payload(self._value):
            return None
        w = self._packed
        if not isinstance(w, torch.Tensor):
            return A.PackedSpikes(A.to_device(np.ascontiguousarray(w).view(np.int32)), self._n_orig, numpy_result=True)
        return A.PackedSpikes(w, self._n_orig)

    def _ids_operand(self):
        """The compacted list as a kernel operand (scatter direction): only for lists this library produced on the
        device — user-supplied lists (``from_packed``) are not trusted to be in range."""
        from . import _array as A
        if self.ndim != 1 or not self._own_ids or _is_float_payload(self._value):
            return None
        return A.ActiveIds(self._active_ids, self._n_active, self._n_orig)

    def __matmul__(self, oc):
        if _is_known_type(oc):
            return BinaryArray(self._value) @ oc
        return oc.__rmatmul__(self)

    def __rmatmul__(self, oc):
        if _is_known_type(oc):
            return oc @ BinaryArray(self._value)
        return oc.__matmul__(self)

    def __repr__(self):
        return (f"CompactBinary(n_orig={self._n_orig}, batch_size={self._batch_size}, bit_width={self._bit_width}, "
                f"dtype={self.dtype})")


def is_event(x) -> bool:
    """Operands the sparse containers accept on the event-driven path."""
    return isinstance(x, (BinaryArray, BitPackedBinary, CompactBinary))


def event_operand(x, allow_packed: bool = True, scatter: bool = False):
    """The array a kernel call receives for event container ``x``: the compacted id list (scatter ops, 1-D
    ``CompactBinary`` built on the device), packed words for 1-D bit-packed containers (when the op takes them),
    otherwise the plain value."""
    if allow_packed and scatter and isinstance(x, CompactBinary):
        p = x._ids_operand()
        if p is not None:
            return p
    if allow_packed and isinstance(x, (BitPackedBinary, CompactBinary)):
        p = x._packed_operand()
        if p is not None:
            return p
    return x.value
