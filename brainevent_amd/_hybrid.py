"""Workspace-sizing contract of the reference's hybrid CSR scheduler, kept API-compatible ("may be a no-op").

The reference sizes a task queue per matrix from four scheduler constants (``brainevent/_csr/hybrid_config.py:77-88``
``HybridConfig``, ``:298-324`` ``hybrid_task_capacity``).  This library has no task queue — its per-matrix workspace is the
:class:`~brainevent_amd.ScatterPlan` — but code written against the reference may still ask for the numbers, so the
dataclass, the resolver (defaults + ``BRAINEVENT_CSR_HYBRID_CONFIG`` JSON override) and the capacity formula are here.
Nothing in the kernels reads them."""
import functools
import json
import os
from dataclasses import dataclass

import numpy as np
import torch

__all__ = ['HybridConfig', 'get_hybrid_config', 'hybrid_task_capacity']

_ENV_OVERRIDE = 'BRAINEVENT_CSR_HYBRID_CONFIG'


@dataclass(frozen=True)
class HybridConfig:
    block_size: int = 256
    fixed_scatter_blocks: int = 2048
    tpr_threshold: int = 128
    task_nnz: int = 4096


@functools.lru_cache(maxsize=None)
def get_hybrid_config() -> HybridConfig:
    raw = os.environ.get(_ENV_OVERRIDE)
    if raw:
        d = json.loads(raw)
        return HybridConfig(**{k: int(d[k]) for k in ('block_size', 'fixed_scatter_blocks', 'tpr_threshold', 'task_nnz') if k in d})
    return HybridConfig()


def hybrid_task_capacity(indptr) -> int:
    """``sum over rows longer than tpr_threshold of ceil(len / task_nnz)`` (reference ``hybrid_config.py:298-324``)."""
    cfg = get_hybrid_config()
    ptr = indptr.detach().cpu().numpy() if isinstance(indptr, torch.Tensor) else np.asarray(indptr)
    ptr = ptr.astype(np.int64)
    if ptr.ndim != 1:
        raise ValueError(f"indptr must be one-dimensional, got shape={ptr.shape}.")
    if ptr.size == 0:
        raise ValueError("indptr must contain at least one element.")
    lens = np.diff(ptr)
    if np.any(lens < 0):
        raise ValueError("CSR row lengths must be non-negative.")
    cap = int(np.where(lens > cfg.tpr_threshold, (lens + cfg.task_nnz - 1) // cfg.task_nnz, 0).sum())
    if cap > np.iinfo(np.int32).max:
        raise ValueError("binary task capacity exceeds int32 range.")
    return cap
