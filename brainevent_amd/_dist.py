"""Multi-GPU partitioning of the spike-triggered scatter: one process per GPU, post-neuron slices, one
spike all-gather per time step (RCCL over xGMI through ``torch.distributed``; ``gloo`` on CPU for tests).

The reference has no distributed path at all; this is the partition the north star
names: rank ``g`` of ``G`` owns the output neurons ``[lo_g, hi_g)`` and stores only the synapses that land
there, with *local* column ids.  Each step every rank contributes the spikes of its own 1/G of the pre
population; one ``all_gather`` rebuilds the full spike vector everywhere; the scatter is local and the
outputs are disjoint (no reduction, no halo).

The exchange carries the spike vector bit-packed (1 bit per neuron: 125 KB instead of 1 MB at N = 1M) when
``packed=True``; xGMI is point-to-point, so the payload per link is what matters.
"""
from typing import Callable, Optional, Tuple

import numpy as np
import torch

__all__ = ['post_slice_bounds', 'pre_slice_bounds', 'shard_csr_by_post', 'shard_fixed_num_by_post', 'SpikeExchange',
           'DistributedScatter']


def post_slice_bounds(n_post: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced slice ``[lo, hi)`` of the post population owned by ``rank``."""
    base, rem = divmod(int(n_post), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


pre_slice_bounds = post_slice_bounds   # the pre population is cut the same way for the spike exchange


def shard_csr_by_post(data: torch.Tensor, indices: torch.Tensor, indptr: torch.Tensor, shape, world: int, rank: int):
    """CSR shard holding the columns ``[lo, hi)`` with local column ids.  Returns ``(data, indices, indptr, shape)``.

    Works on tensors of any device (one-off preprocessing): a column mask, a per-row count via the prefix sum of
    the mask, and a compaction.  ``data`` of one element (homogeneous weight) is passed through.
    """
    m, k = int(shape[0]), int(shape[1])
    lo, hi = post_slice_bounds(k, world, rank)
    keep = (indices >= lo) & (indices < hi)
    csum = torch.zeros(indices.numel() + 1, dtype=torch.int64, device=indices.device)
    torch.cumsum(keep.to(torch.int64), 0, out=csum[1:])
    new_indptr = csum[indptr.to(torch.int64)]
    new_indices = (indices[keep] - lo).to(torch.int32)
    new_data = data if data.numel() == 1 else data[keep]
    if new_indices.numel() <= np.iinfo(np.int32).max:
        new_indptr = new_indptr.to(torch.int32)
    return new_data, new_indices, new_indptr, (m, hi - lo)


def shard_fixed_num_by_post(data: torch.Tensor, indices: torch.Tensor, shape, world: int, rank: int):
    """Post-slice a ``FixedNumPerPre`` matrix.  The slice of a fixed-K row is ragged (about K/G entries), so the
    shard is a CSR matrix (SURVEY.md §7 "hard parts")."""
    n_pre, n_conn = int(indices.shape[0]), int(indices.shape[1])
    indptr = torch.arange(n_pre + 1, dtype=torch.int64, device=indices.device) * n_conn
    flat_data = data if data.numel() == 1 else data.reshape(-1)
    return shard_csr_by_post(flat_data, indices.reshape(-1), indptr, shape, world, rank)


def _pack_bits(spikes: torch.Tensor) -> torch.Tensor:
    """bool[n] -> uint8[ceil(n/8)], bit i%8 of byte i/8 (little-endian bit order)."""
    n = spikes.numel()
    pad = (-n) % 8
    s = spikes.to(torch.uint8)
    if pad:
        s = torch.cat([s, torch.zeros(pad, dtype=torch.uint8, device=s.device)])
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=s.device)
    return (s.view(-1, 8) * w).sum(dim=1, dtype=torch.int32).to(torch.uint8)


def _unpack_bits(packed: torch.Tensor, n: int) -> torch.Tensor:
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=packed.device)
    return ((packed.view(-1, 1) & w) != 0).view(-1)[:n]


class SpikeExchange:
    """All-gather of the per-rank spike slices into the full spike vector (the only collective of the path)."""

    def __init__(self, n_pre: int, group=None, packed: bool = False, device=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.n_pre = int(n_pre)
        self.packed = bool(packed)
        self.bounds = [pre_slice_bounds(n_pre, self.world, r) for r in range(self.world)]
        self.lo, self.hi = self.bounds[self.rank]
        sizes = [b[1] - b[0] for b in self.bounds]
        self.uniform = len(set(sizes)) == 1 and (not packed or sizes[0] % 8 == 0)
        self.device = device
        n_local = sizes[0]
        if self.uniform:
            per = n_local // 8 if packed else n_local
            self._full = torch.empty(per * self.world, dtype=torch.uint8, device=device)
        else:
            # ragged slices: every rank pads its payload to the largest slice (collectives need equal sizes)
            self._pad = max((s + 7) // 8 if packed else s for s in sizes)
            self._chunks = [torch.empty(self._pad, dtype=torch.uint8, device=device) for _ in sizes]

    def gather(self, local_spikes: torch.Tensor) -> torch.Tensor:
        """``local_spikes``: bool / uint8 ``[hi - lo]`` of this rank -> bool ``[n_pre]`` on every rank."""
        assert local_spikes.numel() == self.hi - self.lo
        if self.packed:
            payload = _pack_bits(local_spikes != 0)
        elif local_spikes.dtype == torch.bool:
            payload = local_spikes.view(torch.uint8)            # zero-copy: bool storage is one 0/1 byte per spike
        else:
            payload = (local_spikes != 0).to(torch.uint8)
        if self.uniform:
            self.dist.all_gather_into_tensor(self._full, payload.contiguous(), group=self.group)
            if not self.packed:
                return self._full.view(torch.bool)
            return _unpack_bits(self._full, self.n_pre)
        if payload.numel() < self._pad:
            payload = torch.cat([payload, torch.zeros(self._pad - payload.numel(), dtype=torch.uint8, device=payload.device)])
        self.dist.all_gather(self._chunks, payload.contiguous(), group=self.group)
        parts = [(_unpack_bits(c, b[1] - b[0]) if self.packed else c[:b[1] - b[0]].view(torch.bool))
                 for c, b in zip(self._chunks, self.bounds)]
        return torch.cat(parts)


class DistributedScatter:
    """``spikes @ M`` with ``M`` post-sliced over the ranks of a process group.

    ``matmul(full_spikes, shard) -> local_out`` defaults to the event-driven GPU product; tests inject a CPU
    checker there to exercise the partition + exchange logic under ``gloo``.
    """

    def __init__(self, shard, n_pre: int, group=None, packed: bool = False, device=None,
                 matmul: Optional[Callable] = None):
        self.shard = shard
        self.exchange = SpikeExchange(n_pre, group=group, packed=packed, device=device)
        self.matmul = matmul if matmul is not None else self._gpu_matmul

    @staticmethod
    def _gpu_matmul(full_spikes, shard):
        from ._event import BinaryArray
        return BinaryArray(full_spikes) @ shard

    def step(self, local_spikes: torch.Tensor):
        """One time step: exchange, then the local scatter.  Returns this rank's output slice."""
        return self.matmul(self.exchange.gather(local_spikes), self.shard)
