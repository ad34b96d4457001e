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
