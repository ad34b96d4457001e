"""Multi-GPU partitioning of the spike-triggered scatter: one process per GPU, post-neuron slices, one
spike all-gather per time step (RCCL over xGMI through ``torch.distributed``; ``gloo`` on CPU for tests).

The reference has no distributed path at all; this is the partition the north star
names: rank ``g`` of ``G`` owns the output neurons ``[lo_g, hi_g)`` and stores only the synapses that land
there, with *local* column ids.  Each step every rank contributes the spikes of its own 1/G of the pre
population; one ``all_gather`` rebuilds the full spike vector everywhere; the scatter is local and the
outputs are disjoint (no reduction, no halo).

The exchange carries the spike vector bit-packed (1 bit per neuron: 125 KB instead of 1 MB at N = 1M) when
``packed=True``; xGMI is point-to-point, so the payload per link is what matters, and the kernels consume the
gathered words as they are (no unpack).
"""
from typing import Callable, Optional, Tuple

import numpy as np
import torch

__all__ = ['post_slice_bounds', 'pre_slice_bounds', 'word_aligned_bounds', 'shard_csr_by_post', 'shard_fixed_num_by_post', 'SpikeExchange',
           'NativeSpikeExchange', 'DistributedScatter']


def post_slice_bounds(n_post: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced slice ``[lo, hi)`` of the post population owned by ``rank``."""
    base, rem = divmod(int(n_post), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


pre_slice_bounds = post_slice_bounds   # the pre population is cut the same way for the spike exchange


def shard_csr_by_post(data: torch.Tensor, indices: torch.Tensor, indptr: torch.Tensor, shape, world: int, rank: int,
                      block_entries: int = 1 << 27):
    """CSR shard holding the columns ``[lo, hi)`` with local column ids.  Returns ``(data, indices, indptr, shape)``.

    Works on tensors of any device (one-off preprocessing) in row blocks of about ``block_entries`` entries, so that a
    matrix of 1e10 entries (C4: 40 GB of indices) is cut with about 1 GB of temporaries: pass 1 counts the kept entries
    per row (column mask + prefix sum read at the row boundaries), pass 2 compacts block by block into the
    preallocated shard.  ``data`` of one element (homogeneous weight) is passed through.
    """
    m, k = int(shape[0]), int(shape[1])
    lo, hi = post_slice_bounds(k, world, rank)
    homo = data.numel() == 1
    dev = indices.device
    ptr64 = indptr.to(torch.int64)
    ptr_host = ptr64.cpu().numpy()
    blocks, r = [], 0
    while r < m:
        r2 = int(np.searchsorted(ptr_host, ptr_host[r] + int(block_entries), side='right')) - 1
        r2 = min(m, max(r2, r + 1))
        blocks.append((r, r2))
        r = r2

    def kept(r0, r1):
        e0, e1 = int(ptr_host[r0]), int(ptr_host[r1])
        seg = indices[e0:e1]
        return e0, e1, seg, (seg >= lo) & (seg < hi)

    counts = torch.zeros(m, dtype=torch.int64, device=dev)
    for r0, r1 in blocks:
        e0, e1, seg, keep = kept(r0, r1)
        csum = torch.zeros(e1 - e0 + 1, dtype=torch.int64, device=dev)
        torch.cumsum(keep, 0, out=csum[1:])
        local = ptr64[r0:r1 + 1] - e0
        counts[r0:r1] = csum[local[1:]] - csum[local[:-1]]
        del csum
    new_indptr = torch.zeros(m + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=new_indptr[1:])
    out_host = new_indptr.cpu().numpy()
    total = int(out_host[-1])
    new_indices = torch.empty(total, dtype=torch.int32, device=dev)
    new_data = data if homo else torch.empty(total, dtype=data.dtype, device=dev)
    flat_data = None if homo else data.reshape(-1)
    for r0, r1 in blocks:
        e0, e1, seg, keep = kept(r0, r1)
        o0, o1 = int(out_host[r0]), int(out_host[r1])
        new_indices[o0:o1] = (seg[keep] - lo).to(torch.int32)
        if not homo:
            new_data[o0:o1] = flat_data[e0:e1][keep]
    if total <= np.iinfo(np.int32).max:
        new_indptr = new_indptr.to(torch.int32)
    return new_data, new_indices, new_indptr, (m, hi - lo)


def shard_fixed_num_by_post(data: torch.Tensor, indices: torch.Tensor, shape, world: int, rank: int):
    """Post-slice a ``FixedNumPerPre`` matrix.  The slice of a fixed-K row is ragged (about K/G entries), so the
    shard is a CSR matrix (SURVEY.md §7 "hard parts")."""
    n_pre, n_conn = int(indices.shape[0]), int(indices.shape[1])
    indptr = torch.arange(n_pre + 1, dtype=torch.int64, device=indices.device) * n_conn
    flat_data = data if data.numel() == 1 else data.reshape(-1)
    return shard_csr_by_post(flat_data, indices.reshape(-1), indptr, shape, world, rank)


def _active(spikes: torch.Tensor) -> torch.Tensor:
    """The one spike predicate of the path (``_array.spikes_to_device``; reference ``_csr/binary.py`` ``v > 0.`` for floats,
    ``include/cuda_common.h:120-131``): bool as is, floating point ``> 0``, integers ``!= 0``."""
    if spikes.dtype == torch.bool:
        return spikes
    return spikes > 0 if spikes.dtype.is_floating_point else spikes != 0


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


def word_aligned_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Slice ``[lo, hi)`` of an ``n``-long vector for ``rank`` when every rank owns the same whole number of 32-bit
    words (the last owners may hold fewer — or no — elements).  Used by the bit-packed exchange so the gathered words
    are the packed full vector with no re-alignment."""
    wpr = (((int(n) + 31) // 32) + world - 1) // world
    lo = min(int(n), rank * wpr * 32)
    return lo, min(int(n), (rank + 1) * wpr * 32)


class SpikeExchange:
    """All-gather of the per-rank spike slices into the full spike vector (the only collective of the path).

    ``packed=False``: one byte per neuron, balanced slices (``pre_slice_bounds``).
    ``packed=True`` : one bit per neuron, word-aligned slices (``word_aligned_bounds``).  On the GPU the local slice is
    packed by ``be_pack_spikes``, the words are gathered, and the result is handed to the kernels still packed
    (``BitPackedBinary.from_packed`` -> ``BE_SPIKE_BITS``): 1/8 of the bytes on every xGMI link and no unpack kernel.
    """

    def __init__(self, n_pre: int, group=None, packed: bool = False, device=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.n_pre = int(n_pre)
        self.packed = bool(packed)
        self.device = device
        bounds_of = word_aligned_bounds if self.packed else pre_slice_bounds
        self.bounds = [bounds_of(self.n_pre, self.world, r) for r in range(self.world)]
        self.lo, self.hi = self.bounds[self.rank]
        sizes = [b[1] - b[0] for b in self.bounds]
        if self.packed:
            self.words_per_rank = (((self.n_pre + 31) // 32) + self.world - 1) // self.world
            self._local_words = torch.zeros(max(self.words_per_rank, 1), dtype=torch.int32, device=device)
            self._full_words = torch.zeros(max(self.words_per_rank, 1) * self.world, dtype=torch.int32, device=device)
            self.uniform = True
        else:
            self.uniform = len(set(sizes)) == 1
            if self.uniform:
                self._full = torch.empty(sizes[0] * self.world, dtype=torch.uint8, device=device)
            else:
                # ragged slices: every rank pads its payload to the largest slice (collectives need equal sizes)
                self._pad = max(sizes)
                self._chunks = [torch.empty(self._pad, dtype=torch.uint8, device=device) for _ in sizes]

    # -- packed path ------------------------------------------------------------------------------------
    def _packed_local(self, local_spikes):
        """The slice's own words when the producer hands it over packed (``BitPackedBinary`` / ``PackedSpikes``), else None."""
        from . import _array as A
        from ._event import BitPackedBinary
        if isinstance(local_spikes, BitPackedBinary) and local_spikes.ndim == 1:
            local_spikes = local_spikes._packed_operand()
        if isinstance(local_spikes, A.PackedSpikes):
            assert local_spikes.n == self.hi - self.lo
            return local_spikes.bits if local_spikes.bits.dtype == torch.int32 else local_spikes.bits.view(torch.int32)
        return None

    def _pack_into(self, local_spikes, local_words: torch.Tensor) -> torch.Tensor:
        """The words to send: ``local_words`` filled from the spikes — or the producer's own words when they arrive packed and
        fill the slice (no pack launch, no copy)."""
        n_local = self.hi - self.lo
        words = self._packed_local(local_spikes)
        if words is not None:
            used = (n_local + 31) // 32
            if used == local_words.numel() and words.device == local_words.device:
                return words[:used]
            local_words.zero_()
            local_words[:used] = words[:used].to(local_words.device)
            return local_words
        assert local_spikes.numel() == n_local
        if local_spikes.is_cuda:
            import ctypes
            from . import _array as A
            from ._lib import fn, check
            sp, sd = A.spikes_to_device(local_spikes)
            if n_local:
                f = fn('be_pack_spikes', ctypes.c_int,
                       [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p])
                check(f(A.ptr(sp), sd, n_local, A.ptr(local_words), A.stream_ptr()), 'be_pack_spikes')
        else:           # CPU tensors (gloo tests): same words through tensor arithmetic
            by = _pack_bits(_active(local_spikes))
            buf = local_words.view(torch.uint8)
            buf.zero_()
            buf[:by.numel()] = by
        return local_words

    def _gather_words(self, local_spikes) -> torch.Tensor:
        send = self._pack_into(local_spikes, self._local_words)
        self.dist.all_gather_into_tensor(self._full_words, send, group=self.group)
        return self._full_words

    # -- pipelined exchange (packed mode): post step t+1's spikes, then work on step t ---------------------------
    def post(self, local_spikes: torch.Tensor):
        """Start the exchange of this rank's spikes and return a ticket for :meth:`wait_events`.

        The collective is issued with ``async_op=True``: it runs on the backend's own stream behind whatever is already
        queued on the current stream, so it overlaps with the work queued *after* this call — post the spikes of step
        ``t + 1``, then scatter step ``t``.  That is the schedule of a network whose synaptic delays are at least two
        steps (the spikes a step delivers were emitted before the previous step started).  Two buffers alternate: at most
        one ticket may be in flight while another is being consumed."""
        assert self.packed, "post()/wait_events() use the bit-packed exchange"
        if not hasattr(self, '_slots'):
            self._slots = [(self._local_words, self._full_words),
                           (torch.zeros_like(self._local_words), torch.zeros_like(self._full_words))]
            self._next = 0
        slot = self._next
        self._next ^= 1
        local_words, full_words = self._slots[slot]
        send = self._pack_into(local_spikes, local_words)
        work = self.dist.all_gather_into_tensor(full_words, send, group=self.group, async_op=True)
        return slot, work

    def wait_events(self, ticket):
        """The full spike vector of a posted exchange as an event container (the current stream waits for it)."""
        from ._event import BinaryArray, BitPackedBinary
        slot, work = ticket
        work.wait()
        words = self._slots[slot][1]
        if words.is_cuda:
            return BitPackedBinary.from_packed(words, self.n_pre)
        return BinaryArray(_unpack_bits(words.view(torch.uint8), self.n_pre))

    def gather_events(self, local_spikes: torch.Tensor):
        """This rank's spikes ``[hi - lo]`` -> the full spike vector as an event container usable as ``ev @ shard``."""
        from ._event import BinaryArray, BitPackedBinary
        if self.packed:
            words = self._gather_words(local_spikes)
            if words.is_cuda:
                return BitPackedBinary.from_packed(words, self.n_pre)
            return BinaryArray(_unpack_bits(words.view(torch.uint8), self.n_pre))
        return BinaryArray(self._gather_bytes(local_spikes))

    def gather(self, local_spikes: torch.Tensor) -> torch.Tensor:
        """``local_spikes``: bool / uint8 ``[hi - lo]`` of this rank -> bool ``[n_pre]`` on every rank."""
        return self.gather_events(local_spikes).value

    # -- one byte per neuron ---------------------------------------------------------------------------
    def _gather_bytes(self, local_spikes: torch.Tensor) -> torch.Tensor:
        if local_spikes.dtype == torch.bool:
            payload = local_spikes.view(torch.uint8)            # zero-copy: bool storage is one 0/1 byte per spike
        else:
            payload = _active(local_spikes).to(torch.uint8)
        if self.uniform:
            self.dist.all_gather_into_tensor(self._full, payload.contiguous(), group=self.group)
            return self._full.view(torch.bool)
        if payload.numel() < self._pad:
            payload = torch.cat([payload, torch.zeros(self._pad - payload.numel(), dtype=torch.uint8, device=payload.device)])
        self.dist.all_gather(self._chunks, payload.contiguous(), group=self.group)
        return torch.cat([c[:b[1] - b[0]].view(torch.bool) for c, b in zip(self._chunks, self.bounds)])


def _local_operand(local_spikes, n_local: int):
    """This rank's spikes as (device buffer, spike dtype code).  A packed-only ``BitPackedBinary`` / ``PackedSpikes`` of the
    slice — what a producer that emits words delivers (``lif_coba_step(..., spike_bits=...)``) — is handed over as words
    (``BE_SPIKE_BITS``: the exchange gathers them from where they lie, no pack launch); bits past ``n_local`` must be 0."""
    from . import _array as A
    from ._event import BitPackedBinary
    if isinstance(local_spikes, BitPackedBinary) and local_spikes.ndim == 1:
        pk = local_spikes._packed_operand()
        if pk is not None:
            local_spikes = pk
    if isinstance(local_spikes, A.PackedSpikes):
        assert local_spikes.n == n_local
        return A.to_device(local_spikes.bits), A.BE_SPIKE_BITS
    assert local_spikes.numel() == n_local
    return A.spikes_to_device(local_spikes)



def _device_view_i32(ptr: int, n: int, device) -> torch.Tensor:
    """int32 tensor over ``n`` words of device memory the library owns (``__cuda_array_interface__``; no copy, no ownership)."""
    class _Raw:
        __cuda_array_interface__ = {'shape': (int(n),), 'typestr': '<i4', 'data': (int(ptr), False), 'version': 2}
    return torch.as_tensor(_Raw(), device=device)


class NativeSpikeExchange:
    """The bit-packed exchange through the C ABI alone (``be_exchange_*``: the library loads RCCL itself; no
    ``torch.distributed`` on the data path) — what a binder that is not PyTorch uses.  Rank 0 calls :meth:`unique_id` and
    ships the bytes to the other processes by whatever means it has; every process then constructs the exchange (a
    collective: ``ncclCommInitRank``) on its current device.  Same slices as ``SpikeExchange(packed=True)``."""

    def __init__(self, n_pre: int, world: int, rank: int, unique_id: bytes, device=None):
        import ctypes
        from ._lib import fn, check
        self._ct = ctypes
        self.n_pre, self.world, self.rank = int(n_pre), int(world), int(rank)
        # GPU only (unlike SpikeExchange, which also serves gloo): the receive buffer is handed to ncclAllGather as it is
        device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        if device.type != 'cuda':
            raise ValueError(f"NativeSpikeExchange runs on a HIP device, not on {device} (use SpikeExchange for gloo / CPU)")
        self.device = device
        self._h = ctypes.c_void_p(0)
        buf = ctypes.create_string_buffer(bytes(unique_id), len(unique_id))
        f = fn('be_exchange_init', ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                                  ctypes.POINTER(ctypes.c_void_p)])
        check(f(buf, self.world, self.rank, self.n_pre, ctypes.byref(self._h)), 'be_exchange_init')
        lo, hi = ctypes.c_int64(0), ctypes.c_int64(0)
        f = fn('be_exchange_slice', ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int64),
                                                   ctypes.POINTER(ctypes.c_int64)])
        check(f(self._h, self.rank, ctypes.byref(lo), ctypes.byref(hi)), 'be_exchange_slice')
        try:
            self.lo, self.hi = int(lo.value), int(hi.value)
            if (self.lo, self.hi) != word_aligned_bounds(self.n_pre, self.world, self.rank):
                raise RuntimeError(f"be_exchange_slice disagrees with word_aligned_bounds: {(self.lo, self.hi)}")
            n_words = int(fn('be_exchange_full_words', ctypes.c_int64, [ctypes.c_void_p])(self._h))
            self._full_words = torch.zeros(max(n_words, 1), dtype=torch.int32, device=device)
        except Exception:
            self.close()                  # the communicator must not outlive a failed constructor
            raise

    @staticmethod
    def unique_id() -> bytes:
        import ctypes
        from ._lib import fn, check
        n = int(fn('be_exchange_unique_id_bytes', ctypes.c_int, [])())
        buf = ctypes.create_string_buffer(n)
        check(fn('be_exchange_get_unique_id', ctypes.c_int, [ctypes.c_void_p])(buf), 'be_exchange_get_unique_id')
        return buf.raw

    def gather_events(self, local_spikes: torch.Tensor):
        """This rank's spikes ``[hi - lo]`` -> the full spike vector as a bit-packed event container."""
        from . import _array as A
        from ._event import BitPackedBinary
        from ._lib import fn, check
        sp, sd = _local_operand(local_spikes, self.hi - self.lo)
        ct = self._ct
        f = fn('be_exchange_allgather_bits', ct.c_int, [ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_void_p])
        check(f(self._h, A.ptr(sp), sd, A.ptr(self._full_words), A.stream_ptr()), 'be_exchange_allgather_bits')
        return BitPackedBinary.from_packed(self._full_words, self.n_pre)

    def gather(self, local_spikes: torch.Tensor) -> torch.Tensor:
        return self.gather_events(local_spikes).value

    # -- pipelined exchange: post step t + 1's spikes on the library's own stream, then work on step t ------------------
    def post(self, local_spikes: torch.Tensor, ids: bool = False):
        """Start the exchange of this rank's spikes (``be_exchange_post``: the library's stream waits for the caller's, packs,
        gathers, records an event) and return a ticket for :meth:`wait_events` — same contract as ``SpikeExchange.post``:
        legitimate when synaptic delays are at least two steps; two buffers alternate, at most one ticket in flight while
        another is consumed.  ``ids=True`` (``be_exchange_post_ids``): the gathered words are also compacted into the list of
        active pre neurons on the exchange's stream; :meth:`wait_ids` hands that list to a scatter (``BE_SPIKE_IDS``)."""
        from . import _array as A
        from ._lib import fn, check
        slot = getattr(self, '_next', 0)
        self._next = slot ^ 1
        sp, sd = _local_operand(local_spikes, self.hi - self.lo)
        ct = self._ct
        name = 'be_exchange_post_ids' if ids else 'be_exchange_post'
        f = fn(name, ct.c_int, [ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_int, ct.c_void_p])
        check(f(self._h, A.ptr(sp), sd, slot, A.stream_ptr()), name)
        return slot, sp, bool(ids)           # (the spikes stay referenced until the ticket is consumed)

    def wait_ids(self, ticket):
        """Make the current stream wait for a ticket posted with ``ids=True``; returns ``(be_spike_ids_t, words pointer)`` — the
        struct (host memory holding two device pointers into the handle) is what a scatter entry point takes as its ``spikes``
        argument with ``BE_SPIKE_IDS``.  Valid until the slot is posted again."""
        from . import _array as A
        from ._lib import fn, check
        ct = self._ct
        if not hasattr(self, '_ids_structs'):
            class _Ids(ct.Structure):
                _fields_ = [('active_ids', ct.c_void_p), ('n_active', ct.c_void_p)]
            self._ids_structs = {0: _Ids(), 1: _Ids()}
        slot = ticket[0]
        st_ = self._ids_structs[slot]
        words = ct.c_void_p(0)
        f = fn('be_exchange_wait_ids', ct.c_int, [ct.c_void_p, ct.c_int, ct.c_void_p, ct.POINTER(ct.c_void_p), ct.c_void_p])
        check(f(self._h, slot, ct.byref(st_), ct.byref(words), A.stream_ptr()), 'be_exchange_wait_ids')
        return st_, words.value

    def wait_events(self, ticket):
        from . import _array as A
        from ._event import BitPackedBinary
        from ._lib import fn, check
        slot = ticket[0]
        ct = self._ct
        out = ct.c_void_p(0)
        f = fn('be_exchange_wait', ct.c_int, [ct.c_void_p, ct.c_int, ct.POINTER(ct.c_void_p), ct.c_void_p])
        check(f(self._h, slot, ct.byref(out), A.stream_ptr()), 'be_exchange_wait')
        if not hasattr(self, '_views'):
            self._views = {}
        view = self._views.get(slot)
        if view is None or view.data_ptr() != out.value:       # wrap the library's buffer once (no copy)
            n_words = self._full_words.numel()
            view = self._views[slot] = _device_view_i32(out.value, n_words, self._full_words.device)
        return BitPackedBinary.from_packed(view, self.n_pre)

    def release(self, ticket) -> None:
        """Tell the exchange that the consumer has queued its last read of the ticket's buffer (``be_exchange_release``).
        Only needed when the products that consume the events run on another stream than the one ``post`` was called on."""
        from . import _array as A
        from ._lib import fn, check
        ct = self._ct
        check(fn('be_exchange_release', ct.c_int, [ct.c_void_p, ct.c_int, ct.c_void_p])(self._h, ticket[0], A.stream_ptr()),
              'be_exchange_release')

    def close(self) -> None:
        if self._h:
            from ._lib import fn
            fn('be_exchange_destroy', self._ct.c_int, [self._ct.c_void_p])(self._h)
            self._h = self._ct.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RankStep:
    """One rank's time step — the exchange of the spikes, then ``events @ shard`` — with the host path cut to two C calls.

    ``exchange.gather_events(local) @ shard`` walks the reference's operator surface on every step (event container, ``@``
    dispatch, operand validation, workspace lookup): ≈ 40 µs of Python — more than the 35 µs of kernels a 1-of-8 post slice of
    the 1M x 1M problem runs, so that step was host-bound.  This object resolves once what does not change from step to step
    (function pointers, the shard's workspace and its constant arguments) and then issues ``be_exchange_allgather_bits`` and
    the planned / binned step directly.  Same results bit for bit; anything it does not cover — a shard without a fixed-point
    workspace, weights modified in place since the workspace was filled, an exchange that is not the library's own — takes
    the general path."""

    def __init__(self, exchange, shard):
        import ctypes as ct
        from . import _array as A, _csr as C
        from ._lib import fn
        self.exchange, self.shard = exchange, shard
        self._fast = None
        self._fast_ids = None
        self._ids_pays = False
        ws = shard._scatter_workspace() if hasattr(shard, '_scatter_workspace') else None
        if (not isinstance(exchange, NativeSpikeExchange) or not isinstance(shard, C.CSR)
                or not isinstance(ws, (C.ScatterPlan, C.BinnedScatter))):
            return
        vp, i64, ci = ct.c_void_p, ct.c_int64, ct.c_int
        data = shard.data
        self._gather = fn('be_exchange_allgather_bits', ci, [vp, vp, ci, vp, vp])
        self._words = exchange._full_words
        self._n_local = exchange.hi - exchange.lo
        self._ws_obj, self._data = ws, data
        self._arrays = (shard.indices, getattr(shard, 'indptr', None))
        m, k = ws.m, ws.k
        self._out_shape, self._out_dtype, self._dev = (k,), data.dtype, data.device
        if isinstance(ws, C.ScatterPlan):
            parts = ws.default_parts()
            wsp = ws.workspace(parts, 1)
            f = fn('be_binary_csrmm_t_plan', ci, [vp, ci, ci, vp, vp, vp, ci, vp, i64, i64, i64, ci, ci, ci, ci, ci, ci, vp, i64, vp])
            head = (A.ptr(data), int(ws.homo), A.wcode(data), A.ptr(ws.blob), A.ptr(ws.seg))
            tail = (m, k, 1, ws.slice_shift, ws.slice_width, ws.layout, ws.block_hint, parts)
            self._keep = (wsp,)
            self._fast = lambda out_ptr, words_ptr, st: f(*head, words_ptr, A.BE_SPIKE_BITS, out_ptr, *tail, ws.scale_exp, A.ptr(wsp),
                                                          wsp.numel(), st)
            # the same step over a caller's id list (a host be_spike_ids_t holding two device pointers): no spike-list build
            self._fast_ids = lambda out_ptr, ids_ptr, st: f(*head, ids_ptr, A.BE_SPIKE_IDS, out_ptr, *tail, ws.scale_exp, A.ptr(wsp),
                                                            wsp.numel(), st)
            self._what = 'be_binary_csrmm_t_plan'
        else:
            indices, indptr = shard.indices, getattr(shard, 'indptr', None)
            row_len = -1 if indptr is not None else int(indices.numel() // max(m, 1))
            is64 = int(indptr is not None and indptr.dtype == torch.int64)
            f = fn('be_binary_csrmv_t_binned', ci, [vp, ci, ci, vp, vp, ci, i64, vp, ci, vp, i64, i64, ci, i64, ci, vp, i64, vp])
            head = (A.ptr(data), ws.step_kind, A.wcode(data), A.ptr(indices), A.ptr(indptr), is64, row_len)
            self._keep = (indices, indptr)
            self._fast = lambda out_ptr, words_ptr, st: f(*head, words_ptr, A.BE_SPIKE_BITS, out_ptr, m, k, ws.slice_shift,
                                                          ws.bin_capacity, ws.scale_exp, A.ptr(ws.ws), ws.ws.numel(), st)
            self._fast_ids = lambda out_ptr, ids_ptr, st: f(*head, ids_ptr, A.BE_SPIKE_IDS, out_ptr, m, k, ws.slice_shift,
                                                            ws.bin_capacity, ws.scale_exp, A.ptr(ws.ws), ws.ws.numel(), st)
            self._ids_pays = True
            self._what = 'be_binary_csrmv_t_binned'

    def __call__(self, local_spikes):
        from . import _array as A
        from ._lib import check
        shard = self.shard
        # the fast path holds raw pointers of the arrays it was resolved from: it is taken only while the shard still IS those
        # arrays (a caller may rebind shard.data / indices / indptr to new tensors instead of updating them in place) and its
        # workspace is the same object and up to date with the weights
        if not self._usable():
            return self.exchange.gather_events(local_spikes) @ shard
        sp, sd = _local_operand(local_spikes, self._n_local)
        st = A.stream_ptr()
        ex = self.exchange
        check(self._gather(ex._h, A.ptr(sp), sd, A.ptr(self._words), st), 'be_exchange_allgather_bits')
        out = torch.empty(self._out_shape, dtype=self._out_dtype, device=self._dev)
        check(self._fast(A.ptr(out), A.ptr(self._words), st), self._what)
        return out

    # ---- the pipelined schedule (``--exchange-ahead 1``): the all-gather of step t + 1 runs on the exchange's own stream while
    #      this stream scatters step t.  Legitimate wherever the spikes a step delivers were emitted before the previous step
    #      began — synaptic delays of at least two steps; the reference has no distributed path to compare with.  Same two C
    #      calls per step as the sequential schedule (``be_exchange_post``, then ``be_exchange_wait`` + the planned / binned step).
    def _usable(self) -> bool:
        shard = self.shard
        return not (self._fast is None or shard.data is not self._data
                    or any(a is not b for a, b in zip(self._arrays, (shard.indices, getattr(shard, 'indptr', None))))
                    or self._ws_obj.is_stale(self._data) or shard.buffers.get('scatter_plan') is not self._ws_obj)

    def post(self, local_spikes, ids: Optional[bool] = None) -> None:
        """Queue the exchange of a LATER step's local spikes (at most two may be in flight: the exchange has two buffers).
        ``ids=True`` (native exchange + fast path only): the exchange's stream also compacts the gathered words into the list of
        active rows, and :meth:`step_posted` hands that list to the scatter — no spike-list build on the scattering stream.
        ``ids=None`` (default) decides by what was measured (one rank of eight, profiles/r06_rank_step_schedules.txt): a binned shard
        runs its compaction as a launch of its own on the scattering stream (6 us of the C4 post slice's 88) and gains from the
        list; a planned shard builds its list inside the accumulate kernel, which is faster than reading one from memory
        (C2 post slice: 39.0 us with the words, 40.9 with the list)."""
        if not hasattr(self, '_pending'):
            self._pending = []
        assert len(self._pending) < 2, "RankStep.post: two exchanges are already in flight (consume one with step_posted())"
        if ids is None:
            ids = self._ids_pays
        if ids and self._fast_ids is not None and isinstance(self.exchange, NativeSpikeExchange):
            self._pending.append(self.exchange.post(local_spikes, ids=True))
        else:
            self._pending.append(self.exchange.post(local_spikes))

    def step_posted(self):
        """The step whose exchange was posted first: wait for its gathered words on this stream, then ``events @ shard``."""
        from . import _array as A
        from ._lib import check, fn
        import ctypes as ct
        ticket = self._pending.pop(0)
        ex = self.exchange
        if not self._usable() or not isinstance(ex, NativeSpikeExchange):
            return ex.wait_events(ticket) @ self.shard
        st = A.stream_ptr()
        if len(ticket) > 2 and ticket[2]:          # posted with ids: the list the exchange's stream compacted
            ids_struct, _ = ex.wait_ids(ticket)
            out = torch.empty(self._out_shape, dtype=self._out_dtype, device=self._dev)
            check(self._fast_ids(A.ptr(out), ct.addressof(ids_struct), st), self._what)
            return out
        words = ct.c_void_p(0)
        check(fn('be_exchange_wait', ct.c_int, [ct.c_void_p, ct.c_int, ct.POINTER(ct.c_void_p), ct.c_void_p])(
            ex._h, ticket[0], ct.byref(words), st), 'be_exchange_wait')
        out = torch.empty(self._out_shape, dtype=self._out_dtype, device=self._dev)
        check(self._fast(A.ptr(out), words.value, st), self._what)
        return out

    def ahead(self, local_spikes_next, ids: Optional[bool] = None):
        """One step of the pipelined schedule: post step t + 1's exchange, consume step t's (``post`` step 0 first)."""
        self.post(local_spikes_next, ids=ids)
        return self.step_posted()

    def drain(self) -> None:
        """Wait (on this stream) for every exchange still in flight without consuming it — the end of a pipelined loop."""
        from . import _array as A
        ex = self.exchange
        while getattr(self, '_pending', None):
            t = self._pending.pop(0)
            ex.wait_events(t[:2]) if isinstance(ex, NativeSpikeExchange) else t[1].wait()

    # ---- the two halves of the sequential step on their own (bench.py --gpus N reports them per rank, so that a scaling run
    #      says where a rank's time went: exchange, scatter, or waiting)
    def exchange_only(self, local_spikes) -> None:
        from . import _array as A
        from ._lib import check
        if self._fast is None:
            self.exchange.gather_events(local_spikes)
            return
        sp, sd = _local_operand(local_spikes, self._n_local)
        check(self._gather(self.exchange._h, A.ptr(sp), sd, A.ptr(self._words), A.stream_ptr()), 'be_exchange_allgather_bits')

    def scatter_only(self):
        """The local product over the words of the LAST sequential exchange (``__call__`` / ``exchange_only``)."""
        from . import _array as A
        from ._lib import check
        assert self._fast is not None, "RankStep.scatter_only needs the fast path"
        out = torch.empty(self._out_shape, dtype=self._out_dtype, device=self._dev)
        check(self._fast(A.ptr(out), A.ptr(self._words), A.stream_ptr()), self._what)
        return out


    def check_status(self) -> None:
        """Status of the shard's binned workspace, if it has one (sticky give-up flag, conservation counters): synchronises."""
        ws = getattr(self, '_ws_obj', None)
        if hasattr(ws, 'check_status'):
            ws.check_status()

    def captured(self, static_local, warmup: int = 3):
        """This rank's step as ONE replayable HIP graph: exchange (``ncclAllGather`` on the caller's stream is capturable) +
        the planned / binned step, recorded once over ``static_local`` — a buffer (tensor or ``PackedSpikes`` words) the producer
        rewrites in place before every replay.  Returns a :class:`brainevent_amd._graph.GraphedStep`; its output tensor is
        static too.  Every rank of the group must capture and replay in step (the collective is part of the graph)."""
        from ._graph import GraphedStep
        if self._fast is None:
            raise ValueError("RankStep.captured: this step has no fast path (no fixed-point workspace / not the native exchange).")
        return GraphedStep(lambda: self(static_local), warmup=warmup)


class DistributedScatter:
    """``spikes @ M`` with ``M`` post-sliced over the ranks of a process group.

    The local product defaults to the event-driven GPU kernels (``events @ shard``); tests inject a CPU checker
    ``matmul(full_spikes, shard)`` there to exercise the partition + exchange logic under ``gloo``.
    """

    def __init__(self, shard, n_pre: int, group=None, packed: bool = False, device=None,
                 matmul: Optional[Callable] = None):
        self.shard = shard
        self.exchange = SpikeExchange(n_pre, group=group, packed=packed, device=device)
        self.matmul = matmul

    def step(self, local_spikes: torch.Tensor):
        """One time step: exchange, then the local scatter.  Returns this rank's output slice."""
        events = self.exchange.gather_events(local_spikes)
        if self.matmul is not None:
            return self.matmul(events.value, self.shard)
        return events @ self.shard
