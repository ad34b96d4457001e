"""Operator object with a per-platform backend table.

Keeps the *contract* of the reference's ``XLACustomKernel`` (``brainevent/_op/main.py:96-214``,
backend resolution ``:496-556``, ``def_kernel`` ``:343-416``, ``set_default`` ``:801-864``,
``available_backends`` ``:1205``, ``def_tags`` ``:1152``) without any JAX machinery:

    per-call ``backend=``  >  ``config.set_backend('gpu', …)``  >  per-operator default  >  first registered

An unknown per-call backend raises ``KernelFallbackExhaustedError``; an unknown *global* backend
only warns and falls through to the operator default, as the reference does.
"""
import warnings
from typing import Callable, Dict, Optional, Set

from . import config
from ._error import BenchmarkDataFnNotProvidedError, KernelFallbackExhaustedError, KernelNotAvailableError
from ._registry import register_primitive

PLATFORM = 'gpu'


class OpKernel:
    def __init__(self, name: str):
        self.name = name
        self._kernels: Dict[str, Dict[str, Callable]] = {}
        self._defaults: Dict[str, str] = {}
        self.tags: Set[str] = set()
        self._call_fn: Optional[Callable] = None
        self._benchmark_data_fn: Optional[Callable] = None
        register_primitive(name, self)

    # -- registration -------------------------------------------------------------------------
    def def_kernel(self, backend: str, platform: str, fn: Callable, asdefault: bool = False):
        table = self._kernels.setdefault(platform, {})
        table[backend] = fn
        if asdefault or platform not in self._defaults:
            self._defaults[platform] = backend
        return fn

    def def_tags(self, *tags: str):
        self.tags.update(tags)

    def set_default(self, platform: str, backend: str):
        if backend not in self._kernels.get(platform, {}):
            raise KernelFallbackExhaustedError(
                f"{self.name}: backend {backend!r} is not registered for platform {platform!r}; "
                f"available: {self.available_backends(platform)}")
        self._defaults[platform] = backend

    def available_backends(self, platform: str = PLATFORM):
        return list(self._kernels.get(platform, {}))

    # -- dispatch -----------------------------------------------------------------------------
    def resolve(self, backend: Optional[str] = None, platform: str = PLATFORM) -> Callable:
        table = self._kernels.get(platform, {})
        if not table:
            raise KernelNotAvailableError(f"{self.name}: no kernel registered for platform {platform!r}.")
        if backend is not None:
            if backend not in table:
                raise KernelFallbackExhaustedError(
                    f"{self.name}: backend {backend!r} is not available on {platform!r}; "
                    f"available: {list(table)}")
            return table[backend]
        g = config.get_backend(platform)
        if g is not None:
            if g in table:
                return table[g]
            warnings.warn(f"{self.name}: global backend {g!r} is not registered for {platform!r}; "
                          f"using {self._defaults[platform]!r}.", stacklevel=3)
        return table[self._defaults[platform]]

    def __call__(self, *args, backend: Optional[str] = None, **kwargs):
        return self.resolve(backend)(*args, **kwargs)

    # -- call function and benchmark harness (reference ``_op/main.py:1084-1150`` def_call / call,
    #    ``:1237-1439`` benchmark: warm-up, then timed runs around a device synchronisation) --------------
    def def_call(self, fn: Callable):
        """Associate the user-facing call function (e.g. ``binary_csrmv_p_call``) with this operator."""
        self._call_fn = fn

    def call(self, *args, **kwargs):
        if self._call_fn is None:
            raise ValueError(f"No call function registered for '{self.name}'. "
                             "Use def_call() to register one before calling.")
        return self._call_fn(*args, **kwargs)

    def def_benchmark_data(self, fn: Callable):
        """``fn(platform=...)`` -> iterable of ``(config_name, args, kwargs)`` for :meth:`benchmark`."""
        self._benchmark_data_fn = fn

    def benchmark(self, *, platform: str = PLATFORM, n_warmup: int = 5, n_runs: int = 20, n_batch_per_run: int = 1,
                  compare_results: bool = True, rtol: float = 1e-3, atol: float = 1e-3, verbose: bool = False,
                  catch_errors: bool = True, backends: Optional[list] = None):
        """Time the call function on every registered backend over the configured data; returns a list of records
        ``{name, backend, mean_ms, std_ms, min_ms, success, error}`` (per-call times)."""
        import time
        import numpy as np
        import torch
        if self._benchmark_data_fn is None:
            raise BenchmarkDataFnNotProvidedError(f"benchmark() of '{self.name}' needs def_benchmark_data().")
        if self._call_fn is None:
            raise ValueError(f"benchmark() of '{self.name}' needs def_call().")
        records = []
        for cfg_name, args, kwargs in self._benchmark_data_fn(platform=platform):
            ref_out = None
            for backend in (backends or self.available_backends(platform)):
                rec = {'name': cfg_name, 'backend': backend, 'mean_ms': None, 'std_ms': None, 'min_ms': None,
                       'success': True, 'error': None}
                try:
                    for _ in range(n_warmup):
                        out = self._call_fn(*args, backend=backend, **kwargs)
                    torch.cuda.synchronize()
                    times = []
                    for _ in range(n_runs):
                        t0 = time.perf_counter()
                        for _ in range(n_batch_per_run):
                            out = self._call_fn(*args, backend=backend, **kwargs)
                        torch.cuda.synchronize()
                        times.append((time.perf_counter() - t0) / n_batch_per_run * 1e3)
                    rec.update(mean_ms=float(np.mean(times)), std_ms=float(np.std(times)), min_ms=float(np.min(times)))
                    first = out[0] if isinstance(out, (tuple, list)) else out
                    if compare_results and ref_out is not None and not torch.allclose(first.float(), ref_out.float(),
                                                                                      rtol=rtol, atol=atol):
                        warnings.warn(f"{self.name}[{cfg_name}]: backend {backend!r} disagrees with the first backend.")
                    if ref_out is None:
                        ref_out = first
                except Exception as e:      # noqa: BLE001 - mirrored from the reference's catch_errors switch
                    if not catch_errors:
                        raise
                    rec.update(success=False, error=repr(e))
                if verbose:
                    print(rec)
                records.append(rec)
        return records
