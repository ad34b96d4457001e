"""Build and load ``libbrainevent_amd.so`` (the C ABI declared in ``include/brainevent_amd.h``).

Replaces the reference's run-time compile-and-register pipeline
(``brainevent/_op/kernix_pipeline.py:255-473``, ``kernix_runtime.py:146-317``) with an
ahead-of-time ``hipcc --offload-arch=gfx950`` build and a plain ``ctypes.CDLL``.

There is no CPU fallback: if the library is missing, :func:`lib` raises ``KernelLoadError``; if it
loads but no HIP device is visible, :func:`require_device` raises ``KernelNotAvailableError``.
"""
import ctypes
import os
import shutil
import subprocess
import threading
from pathlib import Path
from typing import List, Optional

from ._error import (KernelCompilationError, KernelExecutionError, KernelLoadError, KernelNotAvailableError,
                     KernelToolchainError)

PKG_DIR = Path(__file__).resolve().parent
CSRC_DIR = PKG_DIR / 'csrc'
LIB_DIR = PKG_DIR / 'lib'
LIB_NAME = 'libbrainevent_amd.so'
HEADER = PKG_DIR.parent / 'include' / 'brainevent_amd.h'
ARCH = 'gfx950'

_lock = threading.Lock()
_lib: Optional[ctypes.CDLL] = None


def lib_path() -> Path:
    override = os.environ.get('BE_HIP_LIB')
    return Path(override) if override else LIB_DIR / LIB_NAME


def sources() -> List[Path]:
    return sorted(CSRC_DIR.glob('*.hip'))


def hipcc_path() -> str:
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and Path(cand).exists():
            return cand
    raise KernelToolchainError("hipcc not found (set HIPCC or put /opt/rocm/bin on PATH).")


def needs_build() -> bool:
    out = lib_path()
    if not out.exists():
        return True
    t = out.stat().st_mtime
    deps = sources() + sorted(CSRC_DIR.glob('*.h')) + [HEADER]
    return any(p.stat().st_mtime > t for p in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile every ``csrc/*.hip`` for gfx950 into one shared library (in-tree)."""
    out = LIB_DIR / LIB_NAME
    if not force and not needs_build():
        return out
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    objs = []
    obj_dir = LIB_DIR / 'obj'
    obj_dir.mkdir(exist_ok=True)
    flags = [f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics',
             '-Wno-unused-result', f'-I{HEADER.parent}'] + os.environ.get('BE_HIPCC_FLAGS', '').split()
    hipcc = hipcc_path()
    hdr_t = max([p.stat().st_mtime for p in CSRC_DIR.glob('*.h')] + [HEADER.stat().st_mtime])
    procs = []
    for src in sources():
        obj = obj_dir / (src.stem + '.o')
        objs.append(obj)
        if not force and obj.exists() and obj.stat().st_mtime > max(src.stat().st_mtime, hdr_t):
            continue
        cmd = [hipcc, *flags, '-c', str(src), '-o', str(obj)]
        if verbose:
            print(' '.join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        log, _ = p.communicate()
        if p.returncode != 0:
            raise KernelCompilationError(f"hipcc failed on {src.name}:\n{log}")
    cmd = [hipcc, f'--offload-arch={ARCH}', '-shared', '-fPIC', *map(str, objs), '-ldl', '-o', str(out)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise KernelCompilationError(f"link failed:\n{r.stdout}")
    return out


def lib() -> ctypes.CDLL:
    """The loaded library (loads on first use; never builds implicitly)."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                path = lib_path()
                if not path.exists():
                    raise KernelLoadError(
                        f"{path} not found. Build it first: `python -c 'import __graft_entry__ as g; g.build()'` "
                        f"or `python -m brainevent_amd._lib`. There is no CPU fallback.")
                try:
                    handle = ctypes.CDLL(str(path))
                except OSError as e:
                    raise KernelLoadError(f"cannot load {path}: {e}") from e
                handle.be_last_error.restype = ctypes.c_char_p
                handle.be_build_arch.restype = ctypes.c_char_p
                _lib = handle
    return _lib


def check(rc: int, what: str = '') -> None:
    if rc < 0:
        msg = lib().be_last_error()
        raise KernelExecutionError(f"{what or 'brainevent_amd'}: status {rc}: {msg.decode() if msg else '?'}")


_device_ok: Optional[bool] = None


def require_device() -> None:
    """Raise unless a HIP device is visible (the hot path has no CPU implementation)."""
    global _device_ok
    if _device_ok is None:
        import torch
        _device_ok = bool(torch.cuda.is_available())
    if not _device_ok:
        raise KernelNotAvailableError(
            "no HIP device visible: brainevent_amd runs its operators on MI355X only "
            "(the CPU oracle under oracle/ is test infrastructure, not a backend).")


_fn_cache = {}


def fn(name: str, restype=ctypes.c_int, argtypes=None):
    """Symbol ``name`` with its prototype set (cached: the hot path calls this every step)."""
    f = _fn_cache.get(name)
    if f is None:
        try:
            f = getattr(lib(), name)
        except AttributeError as e:
            raise KernelLoadError(f"{lib_path()} does not export {name}") from e
        f.restype = restype
        if argtypes is not None:
            f.argtypes = argtypes
        _fn_cache[name] = f
    return f


if __name__ == '__main__':
    print(build(force=True, verbose=True))
