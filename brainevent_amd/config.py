"""Global backend selection (reference ``brainevent/config.py:220-363``: ``set_backend`` /
``get_backend`` / ``clear_backends``).  The only platform here is ``'gpu'`` and its only backend is
``'hip'``; the numba / LFSR / nvcc knobs of the reference have no counterpart."""
from typing import Dict, Optional

_PLATFORMS = ('gpu',)
_backends: Dict[str, Optional[str]] = {}


def set_backend(platform: str, backend: Optional[str]):
    if platform not in _PLATFORMS:
        raise ValueError(f"platform must be one of {_PLATFORMS}, got {platform!r}.")
    if backend is None:
        _backends.pop(platform, None)
    else:
        if not isinstance(backend, str):
            raise TypeError(f"backend must be a string or None, got {type(backend).__name__}.")
        _backends[platform] = backend


def get_backend(platform: str) -> Optional[str]:
    if platform not in _PLATFORMS:
        raise ValueError(f"platform must be one of {_PLATFORMS}, got {platform!r}.")
    return _backends.get(platform)


def clear_backends():
    _backends.clear()
