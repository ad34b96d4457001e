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
from ._error import KernelFallbackExhaustedError, KernelNotAvailableError
from ._registry import register_primitive

PLATFORM = 'gpu'


class OpKernel:
    def __init__(self, name: str):
        self.name = name
        self._kernels: Dict[str, Dict[str, Callable]] = {}
        self._defaults: Dict[str, str] = {}
        self.tags: Set[str] = set()
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
