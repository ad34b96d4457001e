"""Just-in-time connectivity matrices: ``JITC{Scalar,Uniform,Normal}{R,C}`` and ``binary_jit{s,u,n}{mv,mm}``.

Reference surface mirrored (read as text):
  * scalar : ``brainevent/_jit_scalar/main.py:89-252`` (constructor), ``:885-1065`` (``R`` dispatch), ``:1069+`` (``C``),
             ``brainevent/_jit_scalar/binary.py:44-167`` (``binary_jitsmv``), ``:171-286`` (``binary_jitsmm``),
             ``:688-795`` / ``:1247+`` (``*_p_call`` validation), ``:289-421`` / ``:833-957`` (CPU semantics);
  * uniform: ``brainevent/_jit_uniform/main.py:78-190``, ``brainevent/_jit_uniform/binary.py:44-289``, ``:292-415``;
  * normal : ``brainevent/_jit_normal/main.py:78-190``, ``brainevent/_jit_normal/binary.py:44-300``, ``:307-410``;
  * ``seed`` / ``clen``: ``brainevent/_data.py:1181-1245`` (``clen = ceil(2 / prob)`` as int32).

``corder`` alone selects the kernel (``True`` = gather over output rows, ``False`` = scatter over the active
input rows); ``transpose`` only fixes which side of ``shape`` is the output.  ``prob == 0`` yields zeros
(the reference's binary ops leave it undefined; its golden model and float twins return zeros).
The matrix drawn by the ``mv`` ops (lane stride 32) differs from the one drawn by the ``mm`` ops (stride 4),
exactly as in the reference (``brainevent/_misc.py:32-38``).
"""
import ctypes
import math
from typing import Dict, Optional

import numpy as np
import torch

from . import _array as A
from ._data import DataRepresentation
from ._event import BinaryArray, is_event, event_operand
from ._lib import check, fn
from ._op import OpKernel

__all__ = [
    'JITCScalarR', 'JITCScalarC', 'JITCUniformR', 'JITCUniformC', 'JITCNormalR', 'JITCNormalC',
    'binary_jitsmv', 'binary_jitsmm', 'binary_jitumv', 'binary_jitumm', 'binary_jitnmv', 'binary_jitnmm',
    'binary_jitsmv_p', 'binary_jitsmm_p', 'binary_jitumv_p', 'binary_jitumm_p', 'binary_jitnmv_p', 'binary_jitnmm_p',
    'binary_jitsmv_p_call', 'binary_jitsmm_p_call', 'binary_jitumv_p_call', 'binary_jitumm_p_call',
    'binary_jitnmv_p_call', 'binary_jitnmm_p_call', 'JITCScatterShard', 'JITCGatherShard', 'jit_scatter_class_columns',
    'jit_edge_weights',
]

c_i64, c_int, c_vp, c_dbl, c_u32 = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_double, ctypes.c_uint32
_MV_ARGS = [c_dbl, c_dbl, c_i64, c_u32, c_vp, c_int, c_vp, c_i64, c_i64, c_i64, c_int, c_vp, c_i64, c_vp]
_MM_ARGS = [c_dbl, c_dbl, c_i64, c_u32, c_vp, c_int, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp]
_FAMILY = {'s': 0, 'u': 1, 'n': 2}


def _initialize_seed(seed=None) -> int:
    """``int32 (1,)`` seed of the reference (``_data.py:1181-1209``), kept as a python int (low 32 bits key the RNG)."""
    if seed is None:
        seed = int(np.random.randint(0, int(1e8)))
    if isinstance(seed, torch.Tensor):
        seed = seed.reshape(-1)[0].item()
    return int(np.asarray(seed).reshape(-1)[0])


def _initialize_conn_length(prob) -> int:
    """``ceil(2 / prob)`` as int32 (``_data.py:1212-1245``); 0 stands for ``prob == 0``."""
    prob = float(np.asarray(prob).reshape(-1)[0]) if not isinstance(prob, torch.Tensor) else float(prob.reshape(-1)[0].item())
    if prob == 0.0:
        return 0
    return int(min(math.ceil(2.0 / prob), np.iinfo(np.int32).max))


def _scalar(x) -> float:
    if isinstance(x, torch.Tensor):
        return float(x.reshape(-1)[0].item())
    return float(np.asarray(x).reshape(-1)[0])


def _weight_dtype(*ws) -> torch.dtype:
    """Output dtype = dtype of the weight parameters (python floats act as f32, like ``jnp.asarray``)."""
    dts = []
    for w in ws:
        if isinstance(w, torch.Tensor):
            dts.append(w.dtype)
        elif isinstance(w, (np.ndarray, np.generic)):
            dts.append(torch.from_numpy(np.zeros(1, dtype=np.asarray(w).dtype)).dtype)
    if not dts:
        return torch.float32
    dt = dts[0]
    for d in dts[1:]:
        dt = torch.promote_types(dt, d)
    assert dt.is_floating_point, 'Weights must be a floating-point type.'
    return dt


#: bits the smallest non-zero operand value must keep at the fixed-point exponent of the float-operand JITC scatter (below: float atomics)
JIT_FLOAT_MIN_BITS = 24


def _fixed_scale_exp(wmax: float, n_rows: int) -> int:
    e = math.frexp(wmax)[1] if wmax > 0 else 0
    s = 62 - e - max(1, int(math.ceil(math.log2(n_rows + 1))))
    return max(-90, min(150, s))


def _jit_params(family: str, a, b):
    """(w0, w1, |w| bound) of the C ABI for each family."""
    if family == 's':
        w = _scalar(a)
        return w, 0.0, abs(w)
    if family == 'u':
        lo, hi = _scalar(a), _scalar(b)
        return lo, hi - lo, max(abs(lo), abs(hi))
    loc, scale = _scalar(a), _scalar(b)
    return loc, scale, abs(loc) + 6.5 * abs(scale)     # |normal01| <= 6.37 after the 1e-10 clamp


# Scatter workspaces are kept and ARMED (be_jit_scatter_workspace_arm): their spike counters are zeroed once and every later call
# skips the zeroing launch.  One workspace per (device, stream, layout, size) — the layout tag is ('mv',) or ('mm', batch columns):
# two batch shapes of equal byte size place their counter regions differently, and an armed workspace is only clean where ITS
# layout re-arms it — a few at most (least recently used first out: disarmed and released); a call that raises drops its workspace.
# Gather workspaces have no counters and stay per-call allocations.
# While the stream is CAPTURING the cache is not used at all: a graph records raw pointers, so a captured call takes a per-call,
# unarmed workspace from the graph's private pool (it lives as long as the graph; the call zeroes its counters itself, as every
# call did before workspaces were armed) — an eviction here can never pull memory from under a replay, and no eager call can run
# on a workspace a replay is using.
_ARMED_MAX = 8
_armed: 'Dict[tuple, torch.Tensor]' = {}


def _armed_scatter_workspace(nbytes: int, layout: tuple = ('mv',)) -> torch.Tensor:
    if torch.cuda.is_current_stream_capturing():
        return A.workspace(nbytes)
    st = A.stream_ptr()
    key = (A.device().index, int(getattr(st, 'value', st) or 0), tuple(layout), int(nbytes))
    ws = _armed.pop(key, None)
    if ws is None:
        while len(_armed) >= _ARMED_MAX:
            _drop_armed(next(iter(_armed)))
        ws = A.workspace(nbytes)
        check(fn('be_jit_scatter_workspace_arm', c_int, [c_vp, c_i64, c_vp])(A.ptr(ws), ws.numel(), A.stream_ptr()),
              'be_jit_scatter_workspace_arm')
    _armed[key] = ws                      # (re-inserted: most recently used last)
    return ws


def _drop_armed(key) -> None:
    ws = _armed.pop(key, None)
    if ws is not None:
        fn('be_jit_scatter_workspace_disarm', c_int, [c_vp])(A.ptr(ws))


def _drop_armed_tensor(ws: torch.Tensor) -> None:
    for k, v in list(_armed.items()):
        if v is ws:
            _drop_armed(k)


def _jitmv_hip(family, a, b, clen, vector, seed, *, shape, transpose, corder, out_dtype):
    spikes, sd = A.spikes_to_device(vector)
    in_len = int(shape[0] if transpose else shape[1])
    out_len = int(shape[1] if transpose else shape[0])
    out = torch.empty(out_len, dtype=out_dtype, device=A.device())
    if out_len == 0:
        return out
    w0, w1, wmax = _jit_params(family, a, b)
    gather = 1 if corder else 0
    f_ws = fn('be_binary_jitmv_workspace_bytes', c_i64, [c_i64, c_i64, c_i64, c_int])
    nbytes = f_ws(int(shape[1]), in_len, out_len, gather)
    ws = A.workspace(nbytes) if gather else _armed_scatter_workspace(nbytes)
    name = f"be_binary_jit{family}mv_{'notrans' if corder else 'trans'}_{A.wsuffix(out)}"
    f = fn(name, c_int, _MV_ARGS)
    try:
        check(f(w0, w1, int(clen), seed & 0xFFFFFFFF, A.ptr(spikes), sd, A.ptr(out), int(shape[1]), in_len, out_len,
                _fixed_scale_exp(wmax, in_len), A.ptr(ws), ws.numel(), A.stream_ptr()), name)
    except Exception:
        _drop_armed_tensor(ws)
        raise
    return out


def _jitmm_hip(family, a, b, clen, B, seed, *, shape, transpose, corder, out_dtype):
    spikes_bm, sd = A.spikes_batch_major(B)
    n = int(spikes_bm.shape[0])
    in_len = int(shape[0] if transpose else shape[1])
    out_len = int(shape[1] if transpose else shape[0])
    out_bm = torch.empty((n, out_len), dtype=out_dtype, device=A.device())
    if out_len == 0 or n == 0:
        return out_bm.T
    w0, w1, _ = _jit_params(family, a, b)
    f_ws = fn('be_binary_jitmm_workspace_bytes', c_i64, [c_i64, c_i64, c_i64, c_i64, c_int])
    nbytes = f_ws(int(shape[1]), in_len, out_len, n, 1 if corder else 0)
    ws = A.workspace(nbytes) if corder else _armed_scatter_workspace(nbytes, ('mm', n))
    name = f"be_binary_jit{family}mm_{'notrans' if corder else 'trans'}_{A.wsuffix(out_bm)}"
    f = fn(name, c_int, _MM_ARGS)
    try:
        check(f(w0, w1, int(clen), seed & 0xFFFFFFFF, A.ptr(spikes_bm), sd, A.ptr(out_bm), int(shape[1]), in_len,
                out_len, n, A.ptr(ws), ws.numel(), A.stream_ptr()), name)
    except Exception:
        _drop_armed_tensor(ws)
        raise
    return out_bm.T


def _make_ops(family: str, label: str):
    mv_p = OpKernel(f'binary_jit{family}mv')
    mm_p = OpKernel(f'binary_jit{family}mm')

    def mv_hip(a, b, clen, vector, seed, *, shape, transpose, corder, out_dtype):
        return _jitmv_hip(family, a, b, clen, vector, seed, shape=shape, transpose=transpose, corder=corder,
                          out_dtype=out_dtype)

    def mm_hip(a, b, clen, B, seed, *, shape, transpose, corder, out_dtype):
        return _jitmm_hip(family, a, b, clen, B, seed, shape=shape, transpose=transpose, corder=corder,
                          out_dtype=out_dtype)

    mv_p.def_kernel('hip', 'gpu', mv_hip, asdefault=True)
    mm_p.def_kernel('hip', 'gpu', mm_hip, asdefault=True)
    mv_p.def_tags(f'jit_{label}', 'binary')
    mm_p.def_tags(f'jit_{label}', 'binary')
    return mv_p, mm_p


binary_jitsmv_p, binary_jitsmm_p = _make_ops('s', 'scalar')
binary_jitumv_p, binary_jitumm_p = _make_ops('u', 'uniform')
binary_jitnmv_p, binary_jitnmm_p = _make_ops('n', 'normal')


def _check_mv(vector, shape, transpose):
    assert len(shape) == 2, "The matrix shape should be a tuple of two integers."
    assert vector.ndim == 1, f"The vector should be a 1D array, but got {vector.ndim}D."
    if transpose:
        assert shape[0] == len(vector), f"The matrix shape and vector length do not match. {tuple(vector.shape)} @ {shape}"
    else:
        assert shape[1] == len(vector), f"The matrix shape and vector length do not match. {shape} @ {tuple(vector.shape)}"


def _check_mm(B, shape, transpose):
    assert len(shape) == 2, "The matrix shape should be a tuple of two integers."
    assert B.ndim == 2, "The input matrix B should be a 2D array."
    if transpose:
        assert shape[0] == B.shape[0], f"The matrix shape and B shape do not match. {tuple(B.shape)} @ {shape}"
    else:
        assert shape[1] == B.shape[0], f"The matrix shape and B shape do not match. {shape} @ {tuple(B.shape)}"


def _arr(x):
    return x if isinstance(x, torch.Tensor) else np.asarray(x)


# ---- scalar ------------------------------------------------------------------------------------------
def binary_jitsmv_p_call(weight, clen, vector, seed, *, shape, transpose, corder, backend=None):
    _check_mv(vector, shape, transpose)
    return [binary_jitsmv_p(weight, None, clen, vector, seed, shape=tuple(shape), transpose=transpose, corder=corder,
                            out_dtype=_weight_dtype(weight), backend=backend)]


binary_jitsmv_p.def_call(binary_jitsmv_p_call)


def binary_jitsmm_p_call(weight, clen, B, seed, *, shape, transpose, corder, backend=None):
    _check_mm(B, shape, transpose)
    return [binary_jitsmm_p(weight, None, clen, B, seed, shape=tuple(shape), transpose=transpose, corder=corder,
                            out_dtype=_weight_dtype(weight), backend=backend)]


binary_jitsmm_p.def_call(binary_jitsmm_p_call)


def binary_jitsmv(weight, prob, vector, seed: Optional[int] = None, *, shape, transpose: bool = False,
                  corder: bool = True, backend: Optional[str] = None):
    """``y = M @ v`` / ``y = M.T @ v`` with ``M`` drawn on the fly (connection probability ``prob``, constant
    ``weight``); same signature as ``brainevent/_jit_scalar/binary.py:44-54``."""
    as_np = A.wants_numpy(weight, vector)
    r = binary_jitsmv_p_call(weight, _initialize_conn_length(prob), _arr(vector), _initialize_seed(seed), shape=shape,
                             transpose=transpose, corder=corder, backend=backend)[0]
    return A.to_result(r, as_np)


def binary_jitsmm(weight, prob, B, seed: Optional[int] = None, *, shape, transpose: bool = False, corder: bool = True,
                  backend: Optional[str] = None):
    """``Y = M @ B`` / ``Y = M.T @ B`` (``brainevent/_jit_scalar/binary.py:171-286``); the mm walk draws its own matrix."""
    as_np = A.wants_numpy(weight, B)
    r = binary_jitsmm_p_call(weight, _initialize_conn_length(prob), _arr(B), _initialize_seed(seed), shape=shape,
                             transpose=transpose, corder=corder, backend=backend)[0]
    return A.to_result(r, as_np)


# ---- uniform -----------------------------------------------------------------------------------------
def binary_jitumv_p_call(w_low, w_high, clen, vector, seed, *, shape, transpose, corder, backend=None):
    _check_mv(vector, shape, transpose)
    return [binary_jitumv_p(w_low, w_high, clen, vector, seed, shape=tuple(shape), transpose=transpose, corder=corder,
                            out_dtype=_weight_dtype(w_low, w_high), backend=backend)]


binary_jitumv_p.def_call(binary_jitumv_p_call)


def binary_jitumm_p_call(w_low, w_high, clen, B, seed, *, shape, transpose, corder, backend=None):
    _check_mm(B, shape, transpose)
    return [binary_jitumm_p(w_low, w_high, clen, B, seed, shape=tuple(shape), transpose=transpose, corder=corder,
                            out_dtype=_weight_dtype(w_low, w_high), backend=backend)]


binary_jitumm_p.def_call(binary_jitumm_p_call)


def binary_jitumv(w_low, w_high, prob, vector, seed: Optional[int] = None, *, shape, transpose: bool = False,
                  corder: bool = True, backend: Optional[str] = None):
    """JIT connectivity with per-edge weights ``U(w_low, w_high)`` (``brainevent/_jit_uniform/binary.py:44-164``)."""
    as_np = A.wants_numpy(w_low, w_high, vector)
    r = binary_jitumv_p_call(w_low, w_high, _initialize_conn_length(prob), _arr(vector), _initialize_seed(seed),
                             shape=shape, transpose=transpose, corder=corder, backend=backend)[0]
    return A.to_result(r, as_np)


def binary_jitumm(w_low, w_high, prob, B, seed: Optional[int] = None, *, shape, transpose: bool = False,
                  corder: bool = True, backend: Optional[str] = None):
    as_np = A.wants_numpy(w_low, w_high, B)
    r = binary_jitumm_p_call(w_low, w_high, _initialize_conn_length(prob), _arr(B), _initialize_seed(seed), shape=shape,
                             transpose=transpose, corder=corder, backend=backend)[0]
    return A.to_result(r, as_np)


# ---- normal ------------------------------------------------------------------------------------------
def binary_jitnmv_p_call(w_loc, w_scale, clen, vector, seed, *, shape, transpose, corder, backend=None):
    _check_mv(vector, shape, transpose)
    return [binary_jitnmv_p(w_loc, w_scale, clen, vector, seed, shape=tuple(shape), transpose=transpose, corder=corder,
                            out_dtype=_weight_dtype(w_loc, w_scale), backend=backend)]


binary_jitnmv_p.def_call(binary_jitnmv_p_call)


def binary_jitnmm_p_call(w_loc, w_scale, clen, B, seed, *, shape, transpose, corder, backend=None):
    _check_mm(B, shape, transpose)
    return [binary_jitnmm_p(w_loc, w_scale, clen, B, seed, shape=tuple(shape), transpose=transpose, corder=corder,
                            out_dtype=_weight_dtype(w_loc, w_scale), backend=backend)]


binary_jitnmm_p.def_call(binary_jitnmm_p_call)


def binary_jitnmv(w_loc, w_scale, prob, vector, seed: Optional[int] = None, *, shape, transpose: bool = False,
                  corder: bool = True, backend: Optional[str] = None):
    """JIT connectivity with per-edge weights ``N(w_loc, w_scale)`` (``brainevent/_jit_normal/binary.py:44-174``)."""
    as_np = A.wants_numpy(w_loc, w_scale, vector)
    r = binary_jitnmv_p_call(w_loc, w_scale, _initialize_conn_length(prob), _arr(vector), _initialize_seed(seed),
                             shape=shape, transpose=transpose, corder=corder, backend=backend)[0]
    return A.to_result(r, as_np)


def binary_jitnmm(w_loc, w_scale, prob, B, seed: Optional[int] = None, *, shape, transpose: bool = False,
                  corder: bool = True, backend: Optional[str] = None):
    as_np = A.wants_numpy(w_loc, w_scale, B)
    r = binary_jitnmm_p_call(w_loc, w_scale, _initialize_conn_length(prob), _arr(B), _initialize_seed(seed), shape=shape,
                             transpose=transpose, corder=corder, backend=backend)[0]
    return A.to_result(r, as_np)


# =====================================================================================================
# float-operand twins (last clause): the same on-the-fly matrices against a dense vector / matrix
#   reference: brainevent/_jit_scalar/float.py (jitsmv :838-905, jitsmm :1331-1420), _jit_uniform/float.py, _jit_normal/float.py
# =====================================================================================================
def _jit_float_hip(family, a, b, clen, X, seed, *, shape, transpose, corder, out_dtype, mm: bool):
    """``X [in_len] | [in_len, n]`` -> ``[out_len] | [out_len, n]`` through ``be_jitmm_float`` (lane stride 32 for a vector
    operand, 4 for a matrix operand: different draws, as in the reference)."""
    x = A.to_device(X, dtype=out_dtype)
    in_len = int(shape[0] if transpose else shape[1])
    out_len = int(shape[1] if transpose else shape[0])
    vec = x.ndim == 1
    n = 1 if vec else int(x.shape[1])
    out = torch.empty((out_len,) if vec else (out_len, n), dtype=out_dtype, device=A.device())
    if out_len == 0 or n == 0:
        return out
    w0, w1, wmax = _jit_params(family, a, b)
    gather = 1 if corder else 0
    if not gather and out_dtype != torch.float64 and in_len > 0 and int(clen) > 0:
        # The scatter orientation through LDS fixed-point sums (the event-driven scatter's structure with the operand's value as
        # a per-row factor): its exponent needs the operand's largest magnitude — one reduction and a host read per call, against
        # float atomics at 21 G/s otherwise (C3 shape: 757 ms).  f64 keeps the atomic kernel (the factor is formed in f32 here).
        # One device-to-host read per call (it synchronises: this op cannot be captured in a HIP graph — the atomic kernel below
        # can).  Besides the largest magnitude, the SMALLEST non-zero one: the single global exponent e leaves a product
        # |w x| * 2^e integer bits, and an operand of wide dynamic range would lose its small addends (at in_len = 1e6 about 42 bits
        # remain below the largest product: values 2^-40 of max|x| keep 2).  The fixed-point sums are taken only when the smallest
        # non-zero |x| keeps JIT_FLOAT_MIN_BITS bits at e against the weight bound; otherwise the float-atomic kernel, which is
        # what the reference's own GPU path does.
        xa = x.abs()
        xmax, xmin = (float(t) for t in torch.stack([xa.max(), torch.where(xa > 0, xa, torch.full_like(xa, float('inf'))).min()])
                      .to(torch.float64).tolist())
        bound = wmax * xmax * 1.001
        if bound == 0.0:
            return out.zero_()
        e_fix = _fixed_scale_exp(bound, in_len) if math.isfinite(bound) else 0
        resolves = math.isfinite(bound) and math.isfinite(xmin) and wmax * xmin >= math.ldexp(1.0, JIT_FLOAT_MIN_BITS - e_fix)
        if resolves:
            x_bm = x.reshape(1, -1) if vec else x.T.contiguous()
            out_bm = torch.empty((n, out_len), dtype=out_dtype, device=A.device())
            stride = 4 if mm else 32
            ws = A.workspace(fn('be_jitmm_float_scatter_workspace_bytes', c_i64, [c_i64, c_i64, c_i64, c_int])(int(shape[1]), out_len,
                                                                                                             n, stride))
            f = fn('be_jitmm_float_scatter', c_int, [c_int, c_dbl, c_dbl, c_int, c_i64, c_u32, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64,
                                                     c_int, c_int, c_vp, c_i64, c_vp])
            check(f(_FAMILY[family], w0, w1, A.wcode(out), int(clen), seed & 0xFFFFFFFF, A.ptr(x_bm), A.ptr(out_bm), int(shape[1]),
                    in_len, out_len, n, stride, e_fix, A.ptr(ws), ws.numel(), A.stream_ptr()),
                  'be_jitmm_float_scatter')
            return out_bm[0] if vec else out_bm.T
    f_ws = fn('be_jitmm_float_workspace_bytes', c_i64, [c_i64, c_i64, c_i64, c_i64, c_int, c_int])
    ws = A.workspace(f_ws(int(shape[1]), in_len, out_len, n, gather, A.wcode(out)))
    f = fn('be_jitmm_float', c_int, [c_int, c_dbl, c_dbl, c_int, c_i64, c_u32, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_int, c_int,
                                     c_vp, c_i64, c_vp])
    check(f(_FAMILY[family], w0, w1, A.wcode(out), int(clen), seed & 0xFFFFFFFF, A.ptr(x), A.ptr(out), int(shape[1]), in_len,
            out_len, n, 4 if mm else 32, gather, A.ptr(ws), ws.numel(), A.stream_ptr()), 'be_jitmm_float')
    return out


def _make_float_ops(family: str, label: str):
    mv_p = OpKernel(f'jit{family}mv')
    mm_p = OpKernel(f'jit{family}mm')

    def mv_hip(a, b, clen, vector, seed, *, shape, transpose, corder, out_dtype):
        return _jit_float_hip(family, a, b, clen, vector, seed, shape=shape, transpose=transpose, corder=corder,
                              out_dtype=out_dtype, mm=False)

    def mm_hip(a, b, clen, B, seed, *, shape, transpose, corder, out_dtype):
        return _jit_float_hip(family, a, b, clen, B, seed, shape=shape, transpose=transpose, corder=corder,
                              out_dtype=out_dtype, mm=True)

    mv_p.def_kernel('hip', 'gpu', mv_hip, asdefault=True)
    mm_p.def_kernel('hip', 'gpu', mm_hip, asdefault=True)
    mv_p.def_tags(f'jit_{label}', 'float')
    mm_p.def_tags(f'jit_{label}', 'float')
    return mv_p, mm_p


jitsmv_p, jitsmm_p = _make_float_ops('s', 'scalar')
jitumv_p, jitumm_p = _make_float_ops('u', 'uniform')
jitnmv_p, jitnmm_p = _make_float_ops('n', 'normal')


def _float_call(mv: bool, op, a, b, clen, X, seed, *, shape, transpose, corder, backend=None):
    (_check_mv if mv else _check_mm)(X, shape, transpose)
    ws_ = (a,) if b is None else (a, b)
    return [op(a, b, clen, X, seed, shape=tuple(shape), transpose=transpose, corder=corder, out_dtype=_weight_dtype(*ws_),
               backend=backend)]


def jitsmv_p_call(weight, clen, vector, seed, *, shape, transpose, corder, backend=None):
    return _float_call(True, jitsmv_p, weight, None, clen, vector, seed, shape=shape, transpose=transpose, corder=corder, backend=backend)


def jitsmm_p_call(weight, clen, B, seed, *, shape, transpose, corder, backend=None):
    return _float_call(False, jitsmm_p, weight, None, clen, B, seed, shape=shape, transpose=transpose, corder=corder, backend=backend)


def jitumv_p_call(w_low, w_high, clen, vector, seed, *, shape, transpose, corder, backend=None):
    return _float_call(True, jitumv_p, w_low, w_high, clen, vector, seed, shape=shape, transpose=transpose, corder=corder, backend=backend)


def jitumm_p_call(w_low, w_high, clen, B, seed, *, shape, transpose, corder, backend=None):
    return _float_call(False, jitumm_p, w_low, w_high, clen, B, seed, shape=shape, transpose=transpose, corder=corder, backend=backend)


def jitnmv_p_call(w_loc, w_scale, clen, vector, seed, *, shape, transpose, corder, backend=None):
    return _float_call(True, jitnmv_p, w_loc, w_scale, clen, vector, seed, shape=shape, transpose=transpose, corder=corder, backend=backend)


def jitnmm_p_call(w_loc, w_scale, clen, B, seed, *, shape, transpose, corder, backend=None):
    return _float_call(False, jitnmm_p, w_loc, w_scale, clen, B, seed, shape=shape, transpose=transpose, corder=corder, backend=backend)


for _op, _call in ((jitsmv_p, jitsmv_p_call), (jitsmm_p, jitsmm_p_call), (jitumv_p, jitumv_p_call), (jitumm_p, jitumm_p_call),
                   (jitnmv_p, jitnmv_p_call), (jitnmm_p, jitnmm_p_call)):
    _op.def_call(_call)


def jitsmv(weight, prob, vector, seed: Optional[int] = None, *, shape, transpose: bool = False, corder: bool = True,
           backend: Optional[str] = None):
    """``y = M @ v`` / ``y = M.T @ v`` with ``M`` drawn on the fly (constant ``weight``) and a dense ``v`` — every element counts
    (reference ``brainevent/_jit_scalar/float.py``; same draw as :func:`binary_jitsmv`)."""
    as_np = A.wants_numpy(weight, vector)
    return A.to_result(jitsmv_p_call(weight, _initialize_conn_length(prob), _arr(vector), _initialize_seed(seed), shape=shape,
                                     transpose=transpose, corder=corder, backend=backend)[0], as_np)


def jitsmm(weight, prob, B, seed: Optional[int] = None, *, shape, transpose: bool = False, corder: bool = True,
           backend: Optional[str] = None):
    """``Y = M @ B`` / ``Y = M.T @ B`` for a dense ``B`` (the mm walk draws its own matrix, as in the reference)."""
    as_np = A.wants_numpy(weight, B)
    return A.to_result(jitsmm_p_call(weight, _initialize_conn_length(prob), _arr(B), _initialize_seed(seed), shape=shape,
                                     transpose=transpose, corder=corder, backend=backend)[0], as_np)


def jitumv(w_low, w_high, prob, vector, seed: Optional[int] = None, *, shape, transpose: bool = False, corder: bool = True,
           backend: Optional[str] = None):
    as_np = A.wants_numpy(w_low, w_high, vector)
    return A.to_result(jitumv_p_call(w_low, w_high, _initialize_conn_length(prob), _arr(vector), _initialize_seed(seed), shape=shape,
                                     transpose=transpose, corder=corder, backend=backend)[0], as_np)


def jitumm(w_low, w_high, prob, B, seed: Optional[int] = None, *, shape, transpose: bool = False, corder: bool = True,
           backend: Optional[str] = None):
    as_np = A.wants_numpy(w_low, w_high, B)
    return A.to_result(jitumm_p_call(w_low, w_high, _initialize_conn_length(prob), _arr(B), _initialize_seed(seed), shape=shape,
                                     transpose=transpose, corder=corder, backend=backend)[0], as_np)


def jitnmv(w_loc, w_scale, prob, vector, seed: Optional[int] = None, *, shape, transpose: bool = False, corder: bool = True,
           backend: Optional[str] = None):
    as_np = A.wants_numpy(w_loc, w_scale, vector)
    return A.to_result(jitnmv_p_call(w_loc, w_scale, _initialize_conn_length(prob), _arr(vector), _initialize_seed(seed), shape=shape,
                                     transpose=transpose, corder=corder, backend=backend)[0], as_np)


def jitnmm(w_loc, w_scale, prob, B, seed: Optional[int] = None, *, shape, transpose: bool = False, corder: bool = True,
           backend: Optional[str] = None):
    as_np = A.wants_numpy(w_loc, w_scale, B)
    return A.to_result(jitnmm_p_call(w_loc, w_scale, _initialize_conn_length(prob), _arr(B), _initialize_seed(seed), shape=shape,
                                     transpose=transpose, corder=corder, backend=backend)[0], as_np)


# =====================================================================================================
# containers
# =====================================================================================================
def jit_edge_weights(family: str, a, b, seed, rows, cols):
    """f32 weights the edges ``(rows[i], cols[i])`` carry — RNG-orientation coordinates — evaluated by the device hashes the
    products use (``be_jit_edge_weights``; reference ``brainevent/_numba_random.py:424-486``).  ``family``: ``'s'`` scalar
    (``a`` = weight), ``'u'`` uniform (``a, b`` = low, high), ``'n'`` normal (``a, b`` = loc, scale)."""
    as_np = A.wants_numpy(rows, cols)
    r, c = A.to_device(rows, torch.int32).reshape(-1), A.to_device(cols, torch.int32).reshape(-1)
    assert r.numel() == c.numel(), "rows and cols must have the same length"
    w0, w1, _ = _jit_params(family, a, b)
    mode = _FAMILY[family]
    out = torch.empty(r.numel(), dtype=torch.float32, device=r.device)
    f = fn('be_jit_edge_weights', c_int, [c_int, c_dbl, c_dbl, c_u32, c_vp, c_vp, c_i64, c_vp, c_vp])
    check(f(mode, w0, w1, _initialize_seed(seed) & 0xFFFFFFFF, A.ptr(r), A.ptr(c), r.numel(), A.ptr(out), A.stream_ptr()),
          'be_jit_edge_weights')
    return A.to_result(out, as_np)


def _validate_prob(prob) -> float:
    p = np.asarray(prob.cpu() if isinstance(prob, torch.Tensor) else prob)
    if p.size != 1:
        raise ValueError(f"prob must be a scalar, but got shape {p.shape}.")
    p = float(p.item())
    if not np.isfinite(p):
        raise ValueError(f"prob must be finite, but got {p}.")
    if not (0. <= p <= 1.):
        raise ValueError(f"prob must be in [0, 1], but got {p}.")
    return p


class JITCMatrix(DataRepresentation):
    """Common base of the six JIT-connectivity containers."""
    _family = 's'
    _is_row = True          # R classes: logical orientation == generator orientation; C classes: transposed

    def __init__(self, params, *, shape, corder: bool = False, backend: Optional[str] = None,
                 buffers: Optional[Dict] = None):
        *weights, prob, seed = params
        self._weights = tuple(weights)
        self.prob = prob
        _validate_prob(prob)
        self.seed = _initialize_seed(seed)
        self.shape = (int(shape[0]), int(shape[1]))
        self.corder = bool(corder)
        self.backend = backend
        self._init_buffers(buffers)

    @property
    def dtype(self):
        return _weight_dtype(*self._weights)

    ndim = property(lambda self: 2)

    def __repr__(self):
        return (f"{type(self).__name__}(shape={self.shape}, weights={self._weights}, prob={self.prob}, "
                f"seed={self.seed}, corder={self.corder}, backend={self.backend})")

    # -- family hooks ------------------------------------------------------------------------------
    def _mv(self, v, *, shape, transpose, corder):
        f = {'s': binary_jitsmv_p_call, 'u': binary_jitumv_p_call, 'n': binary_jitnmv_p_call}[self._family]
        args = self._weights if self._family != 's' else (self._weights[0],)
        return f(*args, _initialize_conn_length(self.prob), v, self.seed, shape=shape, transpose=transpose,
                 corder=corder, backend=self.backend)[0]

    def _mm(self, B, *, shape, transpose, corder):
        f = {'s': binary_jitsmm_p_call, 'u': binary_jitumm_p_call, 'n': binary_jitnmm_p_call}[self._family]
        args = self._weights if self._family != 's' else (self._weights[0],)
        return f(*args, _initialize_conn_length(self.prob), B, self.seed, shape=shape, transpose=transpose,
                 corder=corder, backend=self.backend)[0]

    def _out(self, r, v):
        as_np = A.wants_numpy(v, *self._weights)
        return A.to_result(r, as_np)

    # -- materialised twin: trades memory for the per-step walk ------------------------------------------------------
    def prepare(self, matrix_mode: str = 'mv', *, force: bool = False):
        """Materialise the drawn connectivity once (:meth:`materialize`: the count / fill kernels, CSR or CSC of ``self.shape``)
        and let the products of that mode (``'mv'``: vector operands, ``'mm'``: matrix operands — different draws, as in the
        reference) run on the stored matrix from then on: event-driven in BOTH directions (planned / binned scatter, and the
        mirror for the other direction), where the on-the-fly walk has to regenerate every row of the unfavourable orientation
        on every call — the reference's default object, ``spk @ JITCScalarR(corder=False)``, walks all 1.6e10 edges of the
        C3 matrix per step (9.9 ms) against ~0.1 ms on the stored matrix (reference: brainevent/_jit_scalar/main.py:990-1007,
        ``corder=False`` -> the gather kernel).  Same numbers: counts x weight for the scalar family (exact), fixed-point sums
        of the same per-edge weights otherwise (1e-6).  Skipped with a warning when the stored matrix would not fit half of the
        free device memory (``force=True`` builds it anyway); f32 weight dtypes only (the stored weights are f32)."""
        if matrix_mode not in ('mv', 'mm'):
            raise ValueError(f"matrix_mode must be 'mv' or 'mm', got {matrix_mode!r}.")
        key = 'materialized_' + matrix_mode
        if self.buffers.get(key) is not None:
            return self
        if self.dtype != torch.float32:
            raise ValueError(f"prepare(): the stored matrix carries f32 weights; this matrix computes in {self.dtype}.")
        from . import _csr as C
        est = float(self.shape[0]) * float(self.shape[1]) * float(self.prob) * 1.05
        need = est * (4 + (0 if self._family == 's' else 4)) * 1.6            # raw arrays + a plan / mirror beside them
        if not force and need > 0.5 * C._free_device_bytes():
            import warnings
            warnings.warn(f"brainevent_amd: the stored form of this {type(self).__name__} ({need / 2**30:.0f} GiB with its workspaces) "
                          f"does not fit beside what is resident; its products stay on the fly (prepare(force=True) overrides).")
            return self
        self.buffers[key] = self.materialize(matrix_mode)
        return self

    def _stored(self, ndim: int):
        """The materialised matrix serving operands of this rank (``None``: on the fly)."""
        return self.buffers.get('materialized_mv' if ndim == 1 else 'materialized_mm')

    # -- dispatch (reference _jit_scalar/main.py:885-1065 for R, :1069+ for C) ---------------------
    def _dense_operand(self, other, left: bool):
        """A plain (non-event) array against the on-the-fly matrix: the float twins, same (shape, transpose, corder) mapping as
        the event-driven products below."""
        x = other if isinstance(other, torch.Tensor) else np.asarray(other)
        if left:
            shape, transpose, corder = (self.shape, True, not self.corder) if self._is_row else (self.shape[::-1], False, not self.corder)
        else:
            shape, transpose, corder = (self.shape, False, self.corder) if self._is_row else (self.shape[::-1], True, self.corder)
        mvf = {'s': jitsmv_p_call, 'u': jitumv_p_call, 'n': jitnmv_p_call}[self._family]
        mmf = {'s': jitsmm_p_call, 'u': jitumm_p_call, 'n': jitnmm_p_call}[self._family]
        args = self._weights if self._family != 's' else (self._weights[0],)
        clen = _initialize_conn_length(self.prob)
        if x.ndim == 1:
            r = mvf(*args, clen, x, self.seed, shape=shape, transpose=transpose, corder=corder, backend=self.backend)[0]
        elif x.ndim == 2:
            r = mmf(*args, clen, x.T if left else x, self.seed, shape=shape, transpose=transpose, corder=corder, backend=self.backend)[0]
            r = r.T if left else r
        else:
            raise NotImplementedError(f"matmul with object of shape {tuple(x.shape)}")
        return self._out(r, x)

    def __matmul__(self, other):
        if not is_event(other):
            return self._dense_operand(other, left=False)
        S = self._stored(other.ndim)
        if S is not None:                  # prepare(): the stored matrix, event-driven through its own workspaces
            return S @ other
        v = event_operand(other)           # (1-D bit-packed containers: their words, BE_SPIKE_BITS — no unpack launch)
        if self._is_row:
            shape, transpose, corder = self.shape, False, self.corder
        else:
            shape, transpose, corder = self.shape[::-1], True, self.corder
        if v.ndim == 1:
            return self._out(self._mv(v, shape=shape, transpose=transpose, corder=corder), v)
        if v.ndim == 2:
            return self._out(self._mm(v, shape=shape, transpose=transpose, corder=corder), v)
        raise NotImplementedError(f"matmul with object of shape {v.shape}")

    def __rmatmul__(self, other):
        if not is_event(other):
            return self._dense_operand(other, left=True)
        S = self._stored(other.ndim)
        if S is not None:
            return other @ S
        v = event_operand(other)           # (1-D bit-packed containers: their words, BE_SPIKE_BITS — no unpack launch)
        if self._is_row:
            shape, transpose, corder = self.shape, True, not self.corder
        else:
            shape, transpose, corder = self.shape[::-1], False, not self.corder
        if v.ndim == 1:
            return self._out(self._mv(v, shape=shape, transpose=transpose, corder=corder), v)
        if v.ndim == 2:
            return self._out(self._mm(v.T, shape=shape, transpose=transpose, corder=corder).T, v)
        raise NotImplementedError(f"matmul with object of shape {v.shape}")

    def _params(self):
        return (*self._weights, self.prob, self.seed)

    def scatter_shard(self, world: int, rank: int) -> 'JITCScatterShard':
        """This rank's share of ``events @ self`` for a multi-GPU run (nothing is stored; see :class:`JITCScatterShard`)."""
        return JITCScatterShard(self, world, rank)

    def gather_shard(self, world: int, rank: int, side: str = 'left') -> 'JITCGatherShard':
        """This rank's output rows of the product whose orientation is the gather kernel (see :class:`JITCGatherShard`)."""
        return JITCGatherShard(self, world, rank, side)

    # -- materialisation (reference: ``mat.mv.tocsr()`` / ``mat.mm.tocsr()``, ``_jit_scalar/main.py`` mode views) ----
    # -- materialisation views: ``mat.mv`` / ``mat.mm`` (reference ``_jit_scalar/main.py:40-110``, ``:404-413``) ----------
    @property
    def mv(self) -> '_JITCModeView':
        """Materialisation view of the matrix ``mat @ vector`` uses (lane stride 32)."""
        return _JITCModeView(self, 'mv')

    @property
    def mm(self) -> '_JITCModeView':
        """Materialisation view of the matrix ``mat @ matrix`` uses (lane stride 4) — a different draw, as in the reference."""
        return _JITCModeView(self, 'mm')

    def todense(self, matrix_mode: Optional[str] = None):
        """Ambiguous without a mode (the mv and mm kernels draw different matrices): use ``mat.mv.todense()`` or
        ``mat.mm.todense()`` — or pass ``matrix_mode``."""
        if matrix_mode is None:
            raise ValueError("todense() is ambiguous for a JIT-connectivity matrix: use mat.mv.todense() or mat.mm.todense().")
        return self.materialize(matrix_mode).todense()

    def tocsr(self, matrix_mode: str = 'mv'):
        """The drawn connectivity as a :class:`CSR` of ``self.shape`` (reference ``mat.mv.tocsr()`` / ``mat.mm.tocsr()``).
        When the walk owners are the logical columns the native form is column-major and this re-encodes it (a device
        sort: sizes up to 2^31 entries); :meth:`materialize` returns the native form without that step."""
        return self.materialize(matrix_mode).tocsr()

    def tocsc(self, matrix_mode: str = 'mv'):
        """The drawn connectivity as a :class:`CSC` of ``self.shape``."""
        return self.materialize(matrix_mode).tocsc()

    def _owner_counts(self, matrix_mode: str = 'mv'):
        """The count pass of :meth:`materialize` (``be_jitc_csr_count``): stored entries per generator row — the rows of
        ``self.shape`` when the walk owners are the logical rows (CSR form), its columns otherwise (CSC form)."""
        if matrix_mode not in ('mv', 'mm'):
            raise ValueError(f"matrix_mode must be 'mv' or 'mm', got {matrix_mode!r}.")
        stride = 32 if matrix_mode == 'mv' else 4
        # orientation of ``M @ v`` for this class (see __matmul__)
        if self._is_row:
            gshape, transpose, corder = self.shape, False, self.corder
        else:
            gshape, transpose, corder = self.shape[::-1], True, self.corder
        in_len = gshape[0] if transpose else gshape[1]
        out_len = gshape[1] if transpose else gshape[0]
        n_rows, walk = (out_len, in_len) if corder else (in_len, out_len)
        dev = A.device()
        clen = _initialize_conn_length(self.prob)
        counts = torch.empty(max(n_rows, 1), dtype=torch.int32, device=dev)
        f_cnt = fn('be_jitc_csr_count', c_int, [c_i64, c_u32, c_i64, c_i64, c_i64, c_int, c_vp, c_vp])
        check(f_cnt(clen, self.seed & 0xFFFFFFFF, int(gshape[1]), n_rows, walk, stride, A.ptr(counts), A.stream_ptr()),
              'be_jitc_csr_count')
        return counts, (gshape, n_rows, walk, stride, clen, corder)

    def owner_counts(self, matrix_mode: str = 'mv') -> torch.Tensor:
        """Stored entries per walk owner (int32, on the device) without materialising the matrix: entries per ROW of ``self.shape``
        when ``materialize()`` would return a CSR, per COLUMN when it would return a CSC."""
        counts, (_, n_rows, *_rest) = self._owner_counts(matrix_mode)
        return counts[:n_rows]

    def materialize(self, matrix_mode: str = 'mv'):
        """Materialise the drawn connectivity on the device.  ``matrix_mode`` picks the matrix of the ``mv`` ops
        (lane stride 32) or of the ``mm`` ops (stride 4) — they differ, as in the reference.

        The generator matrix has the walk owners as rows; for a logical matrix it is the CSR form when the walk
        owners are the logical rows and the CSC form otherwise, so this returns a :class:`CSR` or a :class:`CSC`
        of ``self.shape`` (both multiply identically).  f32 weights.
        """
        from ._csr import CSR, CSC
        counts, (gshape, n_rows, walk, stride, clen, corder) = self._owner_counts(matrix_mode)
        dev = counts.device
        w0, w1, _ = _jit_params(self._family, *(self._weights + (None,))[:2])
        indptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=dev)
        torch.cumsum(counts[:n_rows].to(torch.int64), 0, out=indptr[1:])
        nnz = int(indptr[-1].item())
        indices = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
        mode = _FAMILY[self._family]
        weights = torch.empty(max(nnz, 1), dtype=torch.float32, device=dev) if mode else None
        f_fill = fn('be_jitc_csr_fill', c_int, [c_int, c_dbl, c_dbl, c_i64, c_u32, c_i64, c_i64, c_i64, c_int, c_vp, c_vp,
                                               c_vp, c_vp, c_vp])
        check(f_fill(mode, w0, w1, clen, self.seed & 0xFFFFFFFF, int(gshape[1]), n_rows, walk, stride, A.ptr(indptr),
                     A.ptr(counts), A.ptr(indices), A.ptr(weights), A.stream_ptr()), 'be_jitc_csr_fill')
        data = weights[:nnz] if mode else torch.full((1,), float(w0), dtype=torch.float32, device=dev)
        # generator rows are the outputs of ``M @ v`` iff corder: then G is M row-wise (CSR), else column-wise (CSC)
        cls = CSR if corder else CSC
        out = cls._from_parts(data, indices[:nnz], indptr, shape=self.shape if self._is_row else self.shape,
                              numpy_result=not any(isinstance(w, torch.Tensor) for w in self._weights))
        return out

    def transpose(self, axes=None):
        assert axes is None, "transpose does not support axes argument."
        # (a stored twin — prepare() — belongs to this orientation: the transposed object materialises its own)
        return self._transposed_cls(self._params(), shape=self.shape[::-1], corder=not self.corder, backend=self.backend,
                                    buffers={k: v for k, v in self.buffers.items() if not str(k).startswith('materialized_')})

    T = property(lambda self: self.transpose())


def _unpack(first, rest, n):
    if all(r is None for r in rest):
        data = tuple(first)
    else:
        data = (first, *rest)
    assert len(data) == n, f"Expected {n} parameters, got {len(data)}."
    return data


class _ScalarInit(JITCMatrix):
    _family = 's'

    def __init__(self, weight, prob=None, seed=None, *, shape, corder: bool = False, backend: Optional[str] = None,
                 buffers: Optional[Dict] = None):
        super().__init__(_unpack(weight, (prob, seed), 3), shape=shape, corder=corder, backend=backend, buffers=buffers)

    weight = property(lambda self: self._weights[0])
    data = property(lambda self: self._weights[0])

    def with_data(self, data):
        return type(self)((data, self.prob, self.seed), shape=self.shape, corder=self.corder, backend=self.backend)


class _UniformInit(JITCMatrix):
    _family = 'u'

    def __init__(self, low, high=None, prob=None, seed=None, *, shape, corder: bool = False,
                 backend: Optional[str] = None, buffers: Optional[Dict] = None):
        data = _unpack(low, (high, prob, seed), 4)
        if _scalar(data[0]) > _scalar(data[1]):
            raise ValueError("wlow must be <= whigh element-wise.")
        super().__init__(data, shape=shape, corder=corder, backend=backend, buffers=buffers)

    wlow = property(lambda self: self._weights[0])
    whigh = property(lambda self: self._weights[1])


class _NormalInit(JITCMatrix):
    _family = 'n'

    def __init__(self, loc, scale=None, prob=None, seed=None, *, shape, corder: bool = False,
                 backend: Optional[str] = None, buffers: Optional[Dict] = None):
        super().__init__(_unpack(loc, (scale, prob, seed), 4), shape=shape, corder=corder, backend=backend,
                         buffers=buffers)

    wloc = property(lambda self: self._weights[0])
    wscale = property(lambda self: self._weights[1])


# the reference's names for the per-family bases (``_jit_scalar/main.py:190``, ``_jit_uniform/main.py:78``, ``_jit_normal/main.py:78``)
JITCScalarMatrix, JITCUniformMatrix, JITCNormalMatrix = _ScalarInit, _UniformInit, _NormalInit


class JITCScalarR(_ScalarInit):
    """Row-oriented homogeneous-weight JIT matrix (reference ``_jit_scalar/main.py:558``)."""
    _is_row = True


class JITCScalarC(_ScalarInit):
    """Column-oriented twin (reference ``_jit_scalar/main.py:1069``)."""
    _is_row = False


class JITCUniformR(_UniformInit):
    _is_row = True


class JITCUniformC(_UniformInit):
    _is_row = False


class JITCNormalR(_NormalInit):
    _is_row = True


class JITCNormalC(_NormalInit):
    _is_row = False


JITCScalarR._transposed_cls, JITCScalarC._transposed_cls = JITCScalarC, JITCScalarR
JITCUniformR._transposed_cls, JITCUniformC._transposed_cls = JITCUniformC, JITCUniformR
JITCNormalR._transposed_cls, JITCNormalC._transposed_cls = JITCNormalC, JITCNormalR


class _JITCModeView:
    """``todense`` / ``tocsr`` / ``tocsc`` of a JIT-connectivity matrix for a fixed ``matrix_mode``."""
    __slots__ = ('_mat', '_mode')

    def __init__(self, mat, mode):
        self._mat, self._mode = mat, mode

    def todense(self):
        return self._mat.materialize(self._mode).todense()

    def tocsr(self):
        return self._mat.tocsr(self._mode)

    def tocsc(self):
        return self._mat.tocsc(self._mode)


# =====================================================================================================
# multi-GPU partition of the scatter orientation ("JITC needs no storage — each GPU walks only its
# chunk range / its rows, since the stream is keyed by (seed, row, chunk, lane)")
# =====================================================================================================
def jit_scatter_class_columns(shape1: int, out_len: int, class_begin: int, class_end: int, stride: int = 32) -> np.ndarray:
    """Output columns touched by the walk classes ``[class_begin, class_end)`` (class = chunk * stride + lane):
    ``chunk_start + lane + stride * q`` inside the chunk.  Ascending int64."""
    chunk_size = max(1, (int(shape1) + 3) // 4)
    cols = []
    for c in range(int(class_begin), int(class_end)):
        chunk, lane = divmod(c, stride)
        cs = chunk * chunk_size
        ce = min(cs + chunk_size, int(out_len))
        if cs + lane < ce:
            cols.append(np.arange(cs + lane, ce, stride, dtype=np.int64))
    return np.sort(np.concatenate(cols)) if cols else np.zeros(0, np.int64)


class JITCScatterShard:
    """Rank ``rank`` of ``world``'s share of ``events @ M`` for a JIT-connectivity matrix in its scatter orientation.

    The (chunk, lane) walk classes are dealt to the ranks in contiguous ranges; a class owns its output columns
    outright, so the ranks' outputs are disjoint (no reduction) and nothing but the spike vector is exchanged.
    ``events @ shard`` returns the full-length output with this rank's columns filled and zeros elsewhere;
    ``owned_columns`` lists them.  Only 1-D events (the mv matrix, lane stride 32)."""

    def __init__(self, mat: JITCMatrix, world: int, rank: int):
        from ._dist import post_slice_bounds
        if mat._is_row:       # same (shape, transpose, corder) mapping as JITCMatrix.__rmatmul__
            shape, transpose, corder = mat.shape, True, not mat.corder
        else:
            shape, transpose, corder = mat.shape[::-1], False, not mat.corder
        if corder:
            raise ValueError("events @ M runs the gather kernel for this corder; only the scatter orientation shards by "
                             "walk class (shard the gather orientation by output rows instead).")
        self.mat, self.world, self.rank = mat, int(world), int(rank)
        self.in_len = int(shape[0] if transpose else shape[1])
        self.out_len = int(shape[1] if transpose else shape[0])
        self.shape1 = int(shape[1])                  # keys the chunk width, as in the ops (reference _misc.py:74-122)
        chunk_size = max(1, (self.shape1 + 3) // 4)
        self.n_classes = ((self.out_len + chunk_size - 1) // chunk_size) * 32
        self.class_begin, self.class_end = post_slice_bounds(self.n_classes, self.world, self.rank)
        self._cols = None

    @property
    def owned_columns(self) -> np.ndarray:
        if self._cols is None:
            self._cols = jit_scatter_class_columns(self.shape1, self.out_len, self.class_begin, self.class_end)
        return self._cols

    def __rmatmul__(self, other):
        if not is_event(other):
            raise NotImplementedError("only event operands are on the accelerated path.")
        v = event_operand(other)           # (1-D bit-packed containers: their words, BE_SPIKE_BITS — no unpack launch)
        if v.ndim != 1:
            raise NotImplementedError("JITCScatterShard takes 1-D events.")
        m = self.mat
        in_len, out_len = self.in_len, self.out_len
        assert v.shape[0] == in_len, f"vector length {v.shape[0]} != {in_len}"
        spikes, sd = A.spikes_to_device(v)
        out = torch.empty(out_len, dtype=m.dtype, device=A.device())
        if out_len == 0:
            return m._out(out, v)
        a = m._weights[0]
        b = m._weights[1] if m._family != 's' else 0.0
        w0, w1, wmax = _jit_params(m._family, a, b)
        f_ws = fn('be_binary_jitmv_workspace_bytes', c_i64, [c_i64, c_i64, c_i64, c_int])
        ws = _armed_scatter_workspace(f_ws(self.shape1, in_len, out_len, 0))
        f = fn('be_binary_jitmv_sharded', c_int,
               [c_int, c_dbl, c_dbl, c_int, c_i64, c_u32, c_vp, c_int, c_vp, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_vp,
                c_i64, c_vp])
        check(f(_FAMILY[m._family], w0, w1, A.wcode(out), _initialize_conn_length(m.prob), m.seed & 0xFFFFFFFF, A.ptr(spikes),
                sd, A.ptr(out), self.shape1, in_len, out_len, self.class_begin, self.class_end - self.class_begin,
                _fixed_scale_exp(wmax, in_len), A.ptr(ws), ws.numel(), A.stream_ptr()), 'be_binary_jitmv_sharded')
        return m._out(out, v)


class JITCGatherShard:
    """Rank ``rank`` of ``world``'s share of a JIT-connectivity product in its *gather* orientation, sharded by OUTPUT ROWS
    (open until round 4): the generator rows are the outputs there and a row's walk is keyed by
    ``(seed, row, chunk, lane)`` alone, so a rank that owns the output rows ``[lo, hi)`` (``post_slice_bounds``) computes
    exactly those outputs from the full spike vector — nothing stored, no reduction, the ranks' slices concatenate to the
    unsharded result bit for bit.  ``side='left'``: ``events @ M`` (``M.__rmatmul__``); ``side='right'``: ``M @ events``.
    Whichever side is asked for must run the gather kernel for this object's ``corder`` (the other orientation shards by
    walk class: :class:`JITCScatterShard`).  ``shard.apply(events)`` returns this rank's ``hi - lo`` outputs; 1-D events."""

    def __init__(self, mat: JITCMatrix, world: int, rank: int, side: str = 'left'):
        from ._dist import post_slice_bounds
        if side not in ('left', 'right'):
            raise ValueError("side must be 'left' (events @ M) or 'right' (M @ events).")
        left = side == 'left'
        if mat._is_row:       # the (shape, transpose, corder) mapping of JITCMatrix.__rmatmul__ / __matmul__
            shape, transpose, corder = (mat.shape, True, not mat.corder) if left else (mat.shape, False, mat.corder)
        else:
            shape, transpose, corder = (mat.shape[::-1], False, not mat.corder) if left else (mat.shape[::-1], True, mat.corder)
        if not corder:
            raise ValueError("this side of the product runs the scatter kernel for this corder: shard it by walk class "
                             "(JITCScatterShard / scatter_shard).")
        self.mat, self.world, self.rank, self.side = mat, int(world), int(rank), side
        self.in_len = int(shape[0] if transpose else shape[1])
        self.out_len = int(shape[1] if transpose else shape[0])
        self.shape1 = int(shape[1])
        self.lo, self.hi = post_slice_bounds(self.out_len, self.world, self.rank)

    def apply(self, other):
        if not is_event(other):
            raise NotImplementedError("only event operands are on the accelerated path.")
        v = event_operand(other)
        if v.ndim != 1:
            raise NotImplementedError("JITCGatherShard takes 1-D events.")
        m = self.mat
        assert v.shape[0] == self.in_len, f"vector length {v.shape[0]} != {self.in_len}"
        spikes, sd = A.spikes_to_device(v)
        n_rows = self.hi - self.lo
        out = torch.empty(n_rows, dtype=m.dtype, device=A.device())
        if n_rows == 0:
            return m._out(out, v)
        a = m._weights[0]
        b = m._weights[1] if m._family != 's' else 0.0
        w0, w1, _ = _jit_params(m._family, a, b)
        f_ws = fn('be_binary_jitmv_workspace_bytes', c_i64, [c_i64, c_i64, c_i64, c_int])
        ws = A.workspace(f_ws(self.shape1, self.in_len, n_rows, 1))
        f = fn('be_binary_jitmv_rows', c_int,
               [c_int, c_dbl, c_dbl, c_int, c_i64, c_u32, c_vp, c_int, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp])
        check(f(_FAMILY[m._family], w0, w1, A.wcode(out), _initialize_conn_length(m.prob), m.seed & 0xFFFFFFFF, A.ptr(spikes), sd,
                A.ptr(out), self.shape1, self.in_len, self.lo, n_rows, A.ptr(ws), ws.numel(), A.stream_ptr()), 'be_binary_jitmv_rows')
        return m._out(out, v)

    def __rmatmul__(self, other):          # events @ shard
        if self.side != 'left':
            raise ValueError("this shard was cut for M @ events (side='right').")
        return self.apply(other)

    def __matmul__(self, other):           # shard @ events
        if self.side != 'right':
            raise ValueError("this shard was cut for events @ M (side='left').")
        return self.apply(other)
