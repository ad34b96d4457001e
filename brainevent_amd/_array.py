"""Array plumbing: device buffers are torch ROCm tensors; numpy in -> numpy out.

PyTorch is used only for device memory, streams and (in ``_dist``) process groups.
"""
import ctypes
from typing import Any, Tuple

import numpy as np
import torch

from ._lib import require_device

BE_F32, BE_F64, BE_F16, BE_BF16 = 0, 1, 2, 3
BE_SPIKE_BOOL, BE_SPIKE_FLOAT, BE_SPIKE_BITS, BE_SPIKE_IDS = 0, 1, 2, 3

_W_CODE = {torch.float32: BE_F32, torch.float64: BE_F64, torch.float16: BE_F16, torch.bfloat16: BE_BF16}
_W_SUFFIX = {torch.float32: 'f32', torch.float64: 'f64', torch.float16: 'f16', torch.bfloat16: 'bf16'}


def is_array(x: Any) -> bool:
    return isinstance(x, (np.ndarray, torch.Tensor, np.generic, list, tuple, int, float, bool))


class PackedSpikes:
    """1-D event vector held bit-packed on the device: ``bits`` is ``uint32[ceil(n/32)]`` (stored as an int32
    tensor), bit ``i % 32`` of word ``i // 32`` (the layout of the reference's ``bitpack``,
    ``brainevent/_event/bitpack_binary.py:32-75``).  Quacks like a 1-D bool array for the shape validators."""
    __slots__ = ('bits', 'n', 'numpy_result')

    def __init__(self, bits: torch.Tensor, n: int, numpy_result: bool = False):
        assert bits.ndim == 1 and bits.dtype in (torch.int32, torch.uint32), "bits must be a 1-D 32-bit word tensor"
        assert bits.numel() >= (int(n) + 31) // 32, "bits too short for n"
        self.bits, self.n, self.numpy_result = bits, int(n), bool(numpy_result)

    shape = property(lambda self: (self.n,))
    ndim = property(lambda self: 1)
    dtype = property(lambda self: torch.bool)
    size = property(lambda self: self.n)

    def __len__(self) -> int:
        return self.n


class _SpikeIdsStruct(ctypes.Structure):      # be_spike_ids_t of include/brainevent_amd.h
    _fields_ = [('active_ids', ctypes.c_void_p), ('n_active', ctypes.c_void_p)]


class ActiveIds:
    """1-D event vector given as the device list of its active positions: the first ``count[0]`` entries of ``ids``
    (int32, each ``< n``, listed once) — C ABI code ``BE_SPIKE_IDS``; scatter entry points only.  ``data_ptr()`` is the
    host address of the ``be_spike_ids_t`` the C side reads."""
    __slots__ = ('ids', 'count', 'n', 'numpy_result', '_c')

    def __init__(self, ids: torch.Tensor, count: torch.Tensor, n: int, numpy_result: bool = False):
        assert ids.dtype == torch.int32 and count.dtype == torch.int32 and ids.is_cuda and count.is_cuda
        self.ids, self.count, self.n, self.numpy_result = ids, count, int(n), bool(numpy_result)
        self._c = _SpikeIdsStruct(ids.data_ptr(), count.data_ptr())

    shape = property(lambda self: (self.n,))
    ndim = property(lambda self: 1)
    dtype = property(lambda self: torch.bool)
    size = property(lambda self: self.n)

    def data_ptr(self) -> int:
        return ctypes.addressof(self._c)

    def reshape(self, *shape):
        return _IdsRow(self)


class _IdsRow:
    """``ActiveIds`` seen as a one-row batch (what the batched launchers take)."""
    __slots__ = ('src',)

    def __init__(self, src: ActiveIds):
        self.src = src

    shape = property(lambda self: (1, self.src.n))
    ndim = property(lambda self: 2)

    def data_ptr(self) -> int:
        return self.src.data_ptr()


def wants_numpy(*xs) -> bool:
    """Result type follows the inputs: torch tensor if any operand is one, else numpy."""
    return not any(isinstance(x, torch.Tensor) or (isinstance(x, (PackedSpikes, ActiveIds)) and not x.numpy_result) for x in xs)


_DEVICES = {}


def device() -> torch.device:
    require_device()
    i = torch.cuda.current_device()
    d = _DEVICES.get(i)
    if d is None:
        d = _DEVICES[i] = torch.device('cuda', i)
    return d


def to_device(x, dtype=None) -> torch.Tensor:
    """Contiguous tensor on the current HIP device (zero-copy for tensors already there)."""
    dev = device()
    if isinstance(x, torch.Tensor):
        t = x
    else:
        a = np.asarray(x)
        if a.dtype == np.float64 and dtype is None and not isinstance(x, (np.ndarray, np.generic)):
            a = a.astype(np.float32)   # python floats default to f32 like jnp.asarray
        t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    if t.device != dev:
        t = t.to(dev)
    return t.contiguous()


def to_result(t: torch.Tensor, as_numpy: bool):
    if as_numpy:
        if t.dtype == torch.bfloat16:
            return t.float().cpu().numpy()
        return t.cpu().numpy()
    return t


def wcode(t: torch.Tensor) -> int:
    try:
        return _W_CODE[t.dtype]
    except KeyError:
        raise AssertionError('Weights must be a floating-point type.')


def wsuffix(t: torch.Tensor) -> str:
    return _W_SUFFIX[t.dtype]


def spikes_to_device(v) -> Tuple[torch.Tensor, int]:
    """Event buffer + spike dtype code.

    bool / int8 / uint8 are 1-byte "active when != 0" buffers; float32 is "active when > 0";
    other float widths are thresholded to bool first, other integers are cast to bool
    (reference ``brainevent/_dense/binary.py:162-163``, ``brainevent/_fcn/binary.py:285``).
    """
    if isinstance(v, PackedSpikes):
        return to_device(v.bits), BE_SPIKE_BITS
    if isinstance(v, ActiveIds):
        return v, BE_SPIKE_IDS
    t = to_device(v)
    if t.dtype in (torch.bool, torch.uint8, torch.int8):
        return t, BE_SPIKE_BOOL
    if t.dtype == torch.float32:
        return t, BE_SPIKE_FLOAT
    if t.dtype.is_floating_point:
        return (t > 0), BE_SPIKE_BOOL
    return (t != 0), BE_SPIKE_BOOL


def spikes_batch_major(M) -> Tuple[torch.Tensor, int]:
    """The matrix operand ``M [k, n]`` of a batched op as the kernels take it: event rows ``[n, k]``, contiguous, + dtype code.

    ``BinaryArray(S [n, k]) @ X`` reaches the ops as the view ``S.T`` (the reference's operand convention,
    ``_event/binary.py:209-212``): its transpose already is the batch-major buffer, so nothing is copied — making the view
    contiguous first and transposing it back cost two copies of the spike matrix per call (C5: 13.6 us of a 0.41 ms step)."""
    if isinstance(M, torch.Tensor) and M.ndim == 2 and M.device == device() and M.T.is_contiguous():
        return spikes_to_device(M.T)
    s, sd = spikes_to_device(M)
    return s.T.contiguous(), sd


def ptr(t) -> ctypes.c_void_p:
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)     # the raw hipStream_t without a Stream object


def stream_ptr() -> ctypes.c_void_p:
    """``hipStream_t`` the calls are issued on: torch's current stream (honours stream contexts and graph capture)."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def workspace(nbytes: int) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device())
