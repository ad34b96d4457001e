"""ctypes binding of oracle/liboracle_c.so (TEST INFRASTRUCTURE ONLY — see oracle/README.md)."""
import ctypes
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
_LIB = None

_vp, _i64, _int = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int


def build():
    subprocess.run(['make', '-C', str(_DIR), '-s'], check=True)
    return _DIR / 'liboracle_c.so'


def lib():
    global _LIB
    if _LIB is None:
        path = _DIR / 'liboracle_c.so'
        if not path.exists():
            build()
        _LIB = ctypes.CDLL(str(path))
    return _LIB


def _p(a):
    return ctypes.c_void_p(a.ctypes.data)


def csrmv_f32(w, indices, indptr, v, shape, transpose):
    w = np.ascontiguousarray(w, dtype=np.float32).reshape(-1)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    v = np.ascontiguousarray(v)
    is_float = int(v.dtype == np.float32)
    if not is_float:
        v = v.astype(np.uint8)
    m, k = shape
    out = np.empty(k if transpose else m, dtype=np.float32)
    f = lib().oracle_csrmv_t_f32 if transpose else lib().oracle_csrmv_nt_f32
    f.argtypes = [_vp, _int, _vp, _vp, _vp, _int, _i64, _i64, _vp]
    f.restype = None
    f(_p(w), int(w.size == 1), _p(indices), _p(indptr), _p(v), is_float, m, k, _p(out))
    return out


def csrmv_t_f32_parallel(w, indices, indptr, v, shape, n_threads=8):
    """All-cores upper bound of the scatter (OpenMP + atomic adds); NOT the reference's algorithm, see oracle_c.c."""
    w = np.ascontiguousarray(w, dtype=np.float32).reshape(-1)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    v = np.ascontiguousarray(v)
    is_float = int(v.dtype == np.float32)
    if not is_float:
        v = v.astype(np.uint8)
    m, k = shape
    out = np.empty(k, dtype=np.float32)
    f = lib().oracle_csrmv_t_f32_parallel
    f.argtypes = [_vp, _int, _vp, _vp, _vp, _int, _i64, _i64, _vp, _int]
    f.restype = None
    f(_p(w), int(w.size == 1), _p(indices), _p(indptr), _p(v), is_float, m, k, _p(out), int(n_threads))
    return out


def densemv_f32(w, s, transpose):
    w = np.ascontiguousarray(w, dtype=np.float32)
    s = np.ascontiguousarray(s)
    is_float = int(s.dtype == np.float32)
    if not is_float:
        s = s.astype(np.uint8)
    out = np.empty(w.shape[1] if transpose else w.shape[0], dtype=np.float32)
    f = lib().oracle_densemv_f32
    f.argtypes = [_vp, _i64, _i64, _vp, _int, _int, _vp]
    f.restype = None
    f(_p(w), w.shape[0], w.shape[1], _p(s), is_float, int(transpose), _p(out))
    return out


def jitmv(mode, w0, w1, prob, vector, seed, *, shape, transpose, corder, stride=32):
    """binary_jit{s,u,n}mv in C: mode 's' | 'u' | 'n'; w1 = high for 'u' (span formed here), scale for 'n'."""
    import math
    v = np.asarray(vector)
    act = np.ascontiguousarray(((v != 0) if v.dtype.kind in 'biu' else (v > 0)).astype(np.uint8))
    in_len = shape[0] if transpose else shape[1]
    out_len = shape[1] if transpose else shape[0]
    clen = 0 if float(prob) == 0.0 else max(2, int(math.ceil(2.0 / float(prob))))
    m = {'s': 0, 'u': 1, 'n': 2}[mode]
    a = np.float32(w0)
    b = np.float32(np.float32(w1) - np.float32(w0)) if mode == 'u' else np.float32(w1)
    out = np.empty(out_len, dtype=np.float64)
    f = lib().oracle_jitmv
    f.argtypes = [_int, ctypes.c_float, ctypes.c_float, _i64, ctypes.c_uint32, _vp, _i64, _i64, _i64, _int, _int, _vp]
    f.restype = None
    f(m, a, b, clen, int(seed) & 0xFFFFFFFF, _p(act), int(shape[1]), in_len, out_len, int(bool(corder)), int(stride), _p(out))
    return out
