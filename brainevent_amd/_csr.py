"""CSR / CSC containers and the event-driven CSR products ``binary_csrmv`` / ``binary_csrmm``.

Reference surface being mirrored (read as text, nothing imported):
  * ``brainevent/_csr/main.py:182-277`` (constructor contract), ``:1595-1776`` (``CSR.__matmul__`` /
    ``__rmatmul__``), ``:2509-2696`` (``CSC`` mirror of the same), ``:58-88`` / ``:148-161`` (the
    per-matrix binary workspace that is built on first use and cached in ``buffers``);
  * ``brainevent/_csr/binary.py:128-260`` (``binary_csrmv``), ``:264-384`` (``binary_csrmm``),
    ``:827-987`` / ``:1452-1607`` (``*_p_call`` validation), ``:387-489`` / ``:1029-1160`` (CPU semantics).

Semantics (``e(x)`` = ``x`` for bool, ``x > 0`` for float):
  transpose=True : ``y[j] = sum_{i: e(v[i])} A[i, j]``        (scatter over active rows)
  transpose=False: ``y[i] = sum_j A[i, j] * e(v[j])``          (gather)

The per-matrix workspace of the reference (a task queue sized by ``hybrid_task_capacity``) becomes
here a :class:`ScatterPlan` — the post-sliced row-segment layout consumed by the LDS-accumulating
scatter kernel (see ``csrc/be_csr_plan.hip``; the binned route is ``csrc/be_csr_binned.hip``).  ``workspace=None`` selects the preprocessing-free
"direct" kernel (global atomics).
"""
import ctypes
import os
import math
from typing import Callable, Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _array as A
from ._data import DataRepresentation
from ._error import MathError
from ._event import BinaryArray, is_event, event_operand
from ._lib import check, fn
from ._misc import _as_indptr, _as_int32_indices, _check_compressed_structure
from ._op import OpKernel

__all__ = ['CSR', 'CSC', 'ScatterPlan', 'BinnedScatter', 'Mirror', 'binary_csrmv', 'binary_csrmm', 'binary_csrmv_p',
           'binary_csrmm_p', 'binary_csrmv_p_call', 'binary_csrmm_p_call', 'binary_csrmv_indexed', 'binary_csrmm_indexed',
           'indexed_workspace', 'build_mirror_of', 'hybrid_task_capacity']

c_i64, c_int, c_vp = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p


# =====================================================================================================
# scatter plan (per-matrix workspace)
# =====================================================================================================
class ScatterPlan:
    """Post-sliced row segments of a CSR / fixed-number-connectivity matrix (device resident).

    Built once per matrix (structure + weights), reused by every ``spk @ matrix`` call.  The plan owns one workspace
    (active list, partial sums, a counter that every call leaves zeroed for the next): calls on one plan must be ordered
    on a stream (any single stream, or streams that wait on each other) — two streams running the same plan concurrently
    would share that workspace.  Fields:
    ``seg`` int32 view of ``{uint32 block start / 128 B, uint32 n4}`` per (row, slice), ``blob`` uint8 (the
    128-byte aligned blocks ``[f32 weights][uint16 local columns]``), ``slice_shift``, ``scale_exp``
    (fixed-point exponent).  See ``include/brainevent_amd.h`` for the exact layout.
    """

    #: default slice widths: hetero accumulators are 8 B (2^14 * 8 = 128 KiB of LDS), homo are 4 B
    HETERO_SHIFT = 14
    HOMO_SHIFT = 15
    #: a plan is refused when its smallest non-zero weight would be represented with fewer bits than this
    MIN_WEIGHT_BITS = 16

    #: block layouts (C ABI codes BE_PLAN_U16 / BE_PLAN_D8 / BE_PLAN_H8)
    LAYOUT_U16, LAYOUT_D8, LAYOUT_H8 = 0, 1, 2
    D8_MAX_ROW, D8_MAX_SLICES, D8_CAP, H8_CAP = 16384, 1024, 20000, 40000
    KEEP_ORDER_MAX_BYTES = 4 << 30      # the rows' column order (2 B per entry) stays with a d8 plan by default up to this size

    def __init__(self, m, k, homo, slice_shift, seg, blob, scale_exp, weight_dtype, slice_width=0, layout=0):
        self.m, self.k, self.homo = int(m), int(k), bool(homo)
        self.layout = int(layout)
        self.slice_shift = int(slice_shift)
        self.slice_width = int(slice_width) if slice_width else (1 << self.slice_shift)   # columns per slice
        self.n_slices = (self.k + self.slice_width - 1) // self.slice_width
        self.seg, self.blob = seg, blob
        self.scale_exp = int(scale_exp)
        self.weight_dtype = weight_dtype
        self.nnz = 0                      # stored entries (set by build): sizes the number of parts
        self.row_len = -1                 # fixed row length when the plan was built without an indptr
        self.split_f64 = False            # per-entry f64 weights stored as two f32 entries each
        self.stamp = None                 # weights_stamp() of the weights the blocks were filled from
        self.block_hint_override: Optional[int] = None
        self.items_hint: Optional[int] = None      # d8: average stored items (entries + escapes) per block, from the table
        self._ws: Dict = {}

    # -- sizing ---------------------------------------------------------------------------------
    @staticmethod
    def default_shift(k: int, homo: bool) -> int:
        cap = ScatterPlan.HOMO_SHIFT if homo else ScatterPlan.HETERO_SHIFT
        need = max(4, int(math.ceil(math.log2(max(int(k), 2)))))
        return min(cap, need)

    @staticmethod
    def balanced_width(k: int, slice_shift: int) -> int:
        """Slice width that fills the chip evenly: as many slices as the capacity ``2^slice_shift`` needs, rounded up
        so that ``n_slices * parts`` lands on a multiple of the 256 CUs (k = 1M, shift 14: 62 -> 64 slices of 15625,
        4 workgroups each = 256 equal workgroups instead of 244 full ones and 4 nearly empty)."""
        return ScatterPlan.balanced_width_cap(k, 1 << slice_shift)

    @staticmethod
    def balanced_width_cap(k: int, cap: int, short_blocks: bool = False) -> int:
        """:meth:`balanced_width` for an accumulator capacity of ``cap`` columns (the d8 layout is not tied to powers of
        two: 20000 eight-byte accumulators fill the LDS, k = 1M -> 51 slices of 19608 x 5 parts).  ``short_blocks``: the
        step pays per block, not per byte — as few slices as the capacity allows (a multiple of 8, one per XCD) instead of
        a count that fills every CU (N = 1.5M, K = 1000 weighted: 128 x 2 parts 82 us, 96 x 2 71 us; N = 2.5M: 256 x 1
        198 us, 160 x 1 158 us)."""
        n_min = (int(k) + cap - 1) // cap
        if n_min >= 256 or (short_blocks and n_min > 8):
            n = (n_min + 7) // 8 * 8
        else:
            parts = max(1, 256 // n_min)
            n = max(n_min, 256 // parts)
        return max(1, min(cap, (int(k) + n - 1) // n))

    @property
    def block_hint(self) -> int:
        """Average entries per (row, slice) block — for the d8 layout including its escape entries — (a speed hint for the
        step kernel: lanes per block, pre-gathered segment table; 0 = unknown)."""
        if self.block_hint_override is not None:
            return int(self.block_hint_override)
        if self.items_hint is not None:
            return self.items_hint
        if not self.nnz:
            return 0
        return max(1, min(1 << 20, int(round(self.nnz / max(1, self.m * self.n_slices)))))

    def default_parts(self) -> int:
        """Workgroups per slice.  One 1024-thread workgroup per CU at most (~256 in total), but not more than the matrix can
        feed: every part writes the slice's accumulators once and the reduce reads them back, so a 100k-column matrix cut
        into 51 parts moved 80 MB of partial sums per step for 1e5 updates (33-40 us per step; 24 us, the host's issue
        rate, with parts sized by the entries per slice; ``tools/exp_plan_vs_direct.py``)."""
        return self.parts_for(self.n_slices, self.nnz)

    @staticmethod
    def parts_for(n_slices: int, nnz: int) -> int:
        if n_slices == 1 and 0 < nnz <= (1 << 20):
            return 1          # small matrix: the single-launch kernel (k_plan_single) takes the whole step
        by_chip = 256 // max(n_slices, 1)
        by_work = -(-nnz // (max(n_slices, 1) << 17)) if nnz else by_chip
        return int(max(1, min(64, by_chip, max(1, by_work))))

    #: entries of one block that a wave handles in a single 64-lane pass (4 per lane with weights, 8 without); longer
    #: blocks go through a tail loop whose loads are not prefetched
    HETERO_PASS, HOMO_PASS = 256, 512
    #: auto_geometry thresholds (stored rows per workgroup-part; entries per block; slice width with small partial sums)
    D8_MIN_ROWS_PER_PART, H8_MIN_ROWS_PER_PART, H8_MIN_BLOCK = 20000, 32768, 192
    D8_SMALL_WIDTH, H8_SMALL_WIDTH = 8192, 16384
    #: widest average column gap (slice width / entries per block) the d8 layout is chosen for: a gap above 255 costs an
    #: escape entry, and blocks of 16-40 entries in slices 14000-20000 columns wide carry 1.4-4 escapes per entry
    #: (N = 350k ... 1M, K = 1000: d8 31 / 82 us against 26 / 55 us for uint16 columns decoded by 8 lanes per block;
    #: N = 200k, gap 200: 23 / 23 us)
    D8_MAX_GAP = 160

    #: slices a matrix is cut into when neither the LDS capacity nor the pass size asks for more, and the shortest
    #: average block that is worth it
    TARGET_SLICES, TARGET_MIN_BLOCK = 24, 32

    @classmethod
    def pass_sized_width(cls, k: int, cap: int, row: float, homo: bool, nnz: int = 0) -> int:
        """Balanced slice width for an accumulator capacity of ``cap`` columns.  More slices than the capacity needs when

        * an average (row, slice) block would not fit one 64-lane pass (``HETERO_PASS`` / ``HOMO_PASS`` entries), or
        * the output is small: every workgroup writes its slice of partial sums and the reduce reads them back, so few
          wide slices x many parts move ~32 MB per step whatever the work (N = 100k, K = 1000: 4 slices x 64 parts 33 us,
          32 x 8 19 us; K = 10000: 7 x 36 39 us, 51 x 5 24 us; the cost model in DESIGN.md puts the optimum at 22-31).  Up to ``TARGET_SLICES`` slices as long as an average
          block keeps ``TARGET_MIN_BLOCK`` entries (N = 350k, K = 1000: 11 slices 31 us, 20: 27, 40: 29, 80: 41).

        A matrix that the single-launch kernel takes whole (one slice, <= 1M entries) stays one slice."""
        if int(k) <= cap and 0 < nnz <= (1 << 20):
            return max(1, int(k))
        # fewer outputs than stored rows (a post slice of a multi-GPU partition: 1M x 125k) move the optimum to fewer, wider
        # slices: more active rows per output column make the per-block cost weigh more against the partial sums (cost
        # model in DESIGN.md: n^2 ~ k * acc / active rows).  Measured on the 1-of-8 shard of C2 (tools/exp_shard_geometry.py):
        # 25 slices 37 us weighted / 17.6 counted, 13 slices 28 / 15.8, 7-8 slices 27 / 20.5.
        m_rows = nnz / row if row > 0 and nnz > 0 else float(k)
        target = max(1, int(round(cls.TARGET_SLICES * min(1.0, int(k) / max(m_rows, 1.0)) ** (1.0 / 3.0))))
        n_need = max(1, int(math.ceil(row / (0.82 * (cls.HOMO_PASS if homo else cls.HETERO_PASS)))),
                     min(target, int(row // cls.TARGET_MIN_BLOCK)))
        cap_w = max(16, min(cap, -(-int(k) // n_need)))
        return cls.balanced_width_cap(k, cap_w, short_blocks=row / -(-int(k) // cap_w) < cls.TARGET_MIN_BLOCK)

    @classmethod
    def auto_geometry(cls, m: int, k: int, nnz: int, homo: bool, slice_shift: int, delta_ok: bool = True,
                      force: Optional[str] = None):
        """``(layout, slice_width)`` of a plan built without explicit choices (``force``: ``'u16'`` or ``'delta'``).

        Measured (``tools/exp_layouts.py``, FixedNumPerPre K = 1000 ... 16000, 1 % firing).  Geometry: blocks longer than
        one decode pass are slow in the sorted layouts (K = 16000 homo: 640 entries per block 108 us, 400 per block 71 us;
        hetero 444 -> 141 us, 200 -> 129 us) and wide slices mean large partial sums, so every layout gets pass-sized
        slices (:meth:`pass_sized_width`).  Layout, at equal geometry: ``d8`` beats ``u16`` by 10-14 % and ``h8`` by
        10-15 % when blocks are pass-sized (K = 10000, N = 100k ... 1M); with short rows the sorted layouts' serial decode
        (wave prefix sum + dependent adds) costs 2-5 us whenever a wave meets only a handful of blocks per step, and
        ``h8`` needs blocks long enough for the bytes to matter (N = 1M, K = 3000: 48 us u16, 54 h8).  Hence: ``d8`` when a
        workgroup-part holds >= 20000 stored rows or the slices are narrow (small partial sums) or there is one slice
        (the single-launch kernel; d8 reaches 20000 columns where u16 stops at 16384); ``h8`` when additionally an average
        block has >= 192 entries."""
        U16, delta = cls.LAYOUT_U16, (cls.LAYOUT_H8 if homo else cls.LAYOUT_D8)
        cap16 = 1 << slice_shift
        row = nnz / max(m, 1)
        w16 = cls.pass_sized_width(k, cap16, row, homo, nnz)
        if not delta_ok or force == 'u16':
            return U16, w16
        full = slice_shift >= (cls.HOMO_SHIFT if homo else cls.HETERO_SHIFT)
        cap = (cls.H8_CAP if homo else cls.D8_CAP) if full else cap16
        wd = cls.pass_sized_width(k, cap, row, homo, nnz)
        n_d = -(-int(k) // wd)
        if n_d > cls.D8_MAX_SLICES:
            return U16, w16
        if force == 'delta':
            return delta, wd
        if n_d == 1:
            return (delta, wd) if (not homo or k > cap16) else (U16, w16)
        busy = m / cls.parts_for(n_d, nnz)
        if homo:
            ok = row / n_d >= cls.H8_MIN_BLOCK and (busy >= cls.H8_MIN_ROWS_PER_PART or wd <= cls.H8_SMALL_WIDTH)
        else:
            ok = (busy >= cls.D8_MIN_ROWS_PER_PART or wd <= cls.D8_SMALL_WIDTH) and wd * n_d <= cls.D8_MAX_GAP * row
        return (delta, wd) if ok else (U16, w16)

    def nbytes(self) -> int:
        order = getattr(self, 'order', None)
        return self.seg.numel() * 4 + self.blob.numel() + (order.numel() * 2 if order is not None else 0)

    def workspace(self, parts: int, n_batch: int = 1) -> torch.Tensor:
        # with room for the pre-gathered segment table only when this plan's blocks are short enough for the step to use it
        # (up to 8 GiB at 100 slices x 10M rows); if that does not fit the device, the smaller workspace does: the step
        # then gathers from the plan's own table
        f = fn('be_binary_csrmm_t_plan_workspace_bytes_for', c_i64, [c_i64, c_i64, c_i64, c_int, c_int, c_int, c_int, c_int])
        args = (self.m, self.k, n_batch, self.slice_shift, self.slice_width, parts, int(self.homo))
        need, base = f(*args, int(self.block_hint)), f(*args, 0)
        key = (parts, n_batch, need > base)
        ws = self._ws.get(key)
        if ws is None:
            oom = getattr(torch, 'OutOfMemoryError', getattr(torch.cuda, 'OutOfMemoryError', RuntimeError))
            try:
                ws = A.workspace(need)
            except oom:
                if need == base:
                    raise
                torch.cuda.empty_cache()
                ws = A.workspace(base)
            ws[:4 * max(n_batch, 64)].zero_()   # spike counters: zero on entry, re-armed by every call
            # never evicted: a captured HIP graph keeps the raw pointer of the workspace its launches were recorded with,
            # so a workspace that has been handed out must outlive every later call with another (parts, n_batch)
            self._ws[key] = ws
        return ws

    # -- construction ---------------------------------------------------------------------------
    @classmethod
    def _choose_geometry(cls, m: int, k: int, nnz: int, homo: bool, max_row: Optional[int], slice_shift: Optional[int],
                         slice_width: Optional[int], layout: Optional[str]):
        """(layout code, slice_shift, slice_width, n_slices) of a plan: the sorted-delta layouts (d8 / h8) apply to rows the
        LDS sort holds (``max_row``; None = not asked for) and <= 1024 slices; without explicit choices ``auto_geometry``
        picks by the measured regime."""
        if layout not in (None, 'u16', 'd8', 'h8'):
            raise ValueError(f"layout must be 'u16', 'd8', 'h8' or None, got {layout!r}.")
        if layout == ('h8', 'd8')[homo]:
            raise ValueError("d8 is the layout of heterogeneous weights, h8 the one of a homogeneous weight.")
        if slice_shift is None:
            slice_shift = cls.default_shift(k, homo)
        d8_ok = layout != 'u16' and max_row is not None and max_row <= cls.D8_MAX_ROW
        delta_cap = cls.H8_CAP if homo else cls.D8_CAP
        if slice_width is None:
            lay, slice_width = cls.auto_geometry(m, k, nnz, homo, slice_shift, delta_ok=d8_ok,
                                                 force={'d8': 'delta', 'h8': 'delta', 'u16': 'u16'}.get(layout))
        else:
            lay = (cls.LAYOUT_H8 if homo else cls.LAYOUT_D8) if d8_ok else cls.LAYOUT_U16
            if lay != cls.LAYOUT_U16 and layout is None and slice_width <= (1 << slice_shift):
                lay = cls.auto_geometry(m, k, nnz, homo, slice_shift)[0]     # the width is given, the layout is not
        n_slices = (k + slice_width - 1) // slice_width
        if lay != cls.LAYOUT_U16 and n_slices > cls.D8_MAX_SLICES:
            lay = cls.LAYOUT_U16
        if layout in ('d8', 'h8') and lay == cls.LAYOUT_U16:
            raise ValueError("the d8 / h8 layouts need f32/f16/bf16 weights, rows of at most 16384 entries and at most 1024 "
                             "slices.")
        if not (0 < slice_width <= (1 << slice_shift) or (lay != cls.LAYOUT_U16 and 0 < slice_width <= delta_cap)):
            raise ValueError(f"slice_width {slice_width} exceeds the accumulator capacity of this layout.")
        return lay, slice_shift, slice_width, n_slices

    @classmethod
    def build(cls, weights: torch.Tensor, indices: torch.Tensor, indptr: Optional[torch.Tensor], *, shape,
              row_len: int = -1, slice_shift: Optional[int] = None, slice_width: Optional[int] = None,
              layout: Optional[str] = None, keep_order: Optional[bool] = None) -> 'ScatterPlan':
        """Build the plan on the device.  ``indptr=None`` + ``row_len`` describes fixed-length rows.  ``slice_shift``
        (accumulator capacity) and ``slice_width`` (columns per slice) default to the LDS-filling capacity and the
        balanced width; an explicit ``slice_shift`` alone means full-capacity slices.  ``layout``: ``'u16'`` (uint16
        local columns, 6 B per weighted entry / 2 B per counted one), ``'d8'`` (sorted columns as uint8 deltas, 5 B per
        entry: heterogeneous weights), ``'h8'`` (uint8 advance codes, 1 B per entry: one homogeneous weight) — both for
        rows of at most 16384 entries and at most 1024 slices — or ``None`` = the delta layout whenever it applies.

        The sorted layouts need every row in column order: the count pass stores that order (2 bytes per entry) and the fill
        reads it back instead of sorting again.  ``keep_order=True`` keeps it with the plan, which makes every later
        :meth:`refresh_weights` a gather-copy instead of a re-sort (weights updated in place — plasticity); ``False`` frees it
        after the fill; ``None`` (default) keeps it while it is small next to the plan (``KEEP_ORDER_MAX_BYTES``).  When the
        order does not fit beside the plan during the build it is simply not used (three sorts, as before)."""
        m, k = int(shape[0]), int(shape[1])
        weights = A.to_device(weights).reshape(-1)
        indices = A.to_device(indices).reshape(-1)
        assert indices.dtype == torch.int32
        homo = weights.numel() == 1
        out_dtype = weights.dtype
        if indptr is not None:
            indptr = A.to_device(indptr)
        split = weights.dtype == torch.float64 and not homo
        src_stamp = weights_stamp(weights)
        if split:       # per-entry f64 weights: every entry becomes two f32 entries (see _split_f64)
            weights, indices, indptr, row_len = _split_f64(weights, indices, indptr, row_len)
        dev = A.device()
        st = A.stream_ptr()
        is64 = int(indptr is not None and indptr.dtype == torch.int64)
        nnz = int(indices.numel())
        max_row = None
        if layout != 'u16':
            max_row = int(row_len) if indptr is None else (int((indptr[1:] - indptr[:-1]).max().item()) if m > 0 else 0)
        lay, slice_shift, slice_width, n_slices = cls._choose_geometry(m, k, nnz, homo, max_row, slice_shift, slice_width, layout)
        seg = torch.empty(n_slices * m * 2, dtype=torch.int32, device=dev)   # {uint32 start, uint32 n4} pairs
        f_scr = fn('be_scatter_plan_scratch_bytes', c_i64, [c_i64, c_i64, c_int, c_int])
        scratch = A.workspace(f_scr(m, k, slice_shift, slice_width))
        blob_bytes = c_i64(0)
        order = None
        if keep_order is None and os.environ.get('BE_PLAN_KEEP_ORDER'):          # (A/B runs)
            keep_order = os.environ['BE_PLAN_KEEP_ORDER'] == '1'
        if lay != cls.LAYOUT_U16 and nnz and keep_order is not False:
            try:            # the rows' column order, written by the count pass and read back by the fill (one sort instead of two)
                order = torch.empty(nnz, dtype=torch.int16, device=dev)
            except getattr(torch, 'OutOfMemoryError', RuntimeError):
                torch.cuda.empty_cache()
        f_cnt = fn('be_scatter_plan_count_ordered', c_int,
                   [c_vp, c_vp, c_int, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_int, c_vp, c_vp, c_i64, ctypes.POINTER(c_i64),
                    c_vp, c_vp])
        check(f_cnt(A.ptr(indices), A.ptr(indptr), is64, row_len, m, k, slice_shift, slice_width, int(homo), lay, A.ptr(seg),
                    A.ptr(scratch), scratch.numel(), ctypes.byref(blob_bytes), A.ptr(order), st), 'be_scatter_plan_count_ordered')
        try:
            blob = torch.empty(int(blob_bytes.value) + 128, dtype=torch.uint8, device=dev)
        except getattr(torch, 'OutOfMemoryError', RuntimeError):
            if order is None:
                raise
            order = None            # the order and the blocks do not fit together: build without it
            torch.cuda.empty_cache()
            blob = torch.empty(int(blob_bytes.value) + 128, dtype=torch.uint8, device=dev)
        plan = cls(m, k, homo, slice_shift, seg, blob, 0, out_dtype, slice_width, lay)
        plan.nnz = nnz
        plan.order = order
        if lay == cls.LAYOUT_D8 and nnz:
            # lane-groups of 4 items per block (low 16 bits of the table's second word), over a strided sample of the blocks
            n_blk = m * n_slices
            ng = seg.view(n_blk, 2)[::max(1, n_blk >> 22), 1] & 0xffff
            plan.items_hint = max(1, min(1 << 20, int(round(4.0 * ng.double().mean().item()))))
        plan.row_len = int(row_len)
        plan.split_f64 = split
        plan._fill(weights, indices, indptr)
        plan.stamp = src_stamp
        if keep_order is None:
            keep_order = order is not None and not homo and order.numel() * 2 <= cls.KEEP_ORDER_MAX_BYTES
        if not keep_order or homo:      # (one shared weight is never re-encoded)
            plan.order = None
        return plan

    @classmethod
    def build_from_blocks(cls, get_block: Callable[[int, int], Tuple], block_rows: int, *, shape, nnz: int, max_row_len: int,
                          homo: bool, weight_dtype: torch.dtype = torch.float32, slice_shift: Optional[int] = None,
                          slice_width: Optional[int] = None, layout: Optional[str] = None) -> 'ScatterPlan':
        """Build the plan of a matrix whose raw arrays are **never resident in full**: ``get_block(r0, r1)`` returns the rows
        ``[r0, r1)`` as ``(weights, indices, indptr)`` on the device — ``indptr`` relative to the block (``indptr[0] == 0``,
        int32 or int64; or ``None`` with rows of exactly ``max_row_len`` entries) — and is called twice per block: once for the
        count pass, once for the fill.  Everything the two passes touch is local to a row except the block starts, which one
        scan over the whole segment table provides in between (``be_scatter_plan_begin / _count_rows / _scan``, then
        ``be_scatter_plan_fill_ordered`` per block).  The result is the plan :meth:`build` gives for the whole matrix (same
        segment table, same blocks); a matrix whose raw CSR and plan do not fit the device together — 1.5M x 1.5M with 15 000
        entries per row: 180 GB + 130 GB — is planned this way and used through :class:`PlannedMatrix`.  ``nnz`` and
        ``max_row_len`` describe the whole matrix (they pick the geometry); f64 per-entry weights are not served here."""
        m, k = int(shape[0]), int(shape[1])
        if weight_dtype == torch.float64 and not homo:
            raise ValueError("build_from_blocks: per-entry f64 weights are not served (build() splits them; do the same per block).")
        lay, slice_shift, slice_width, n_slices = cls._choose_geometry(m, k, int(nnz), bool(homo), int(max_row_len), slice_shift,
                                                                       slice_width, layout)
        dev, st = A.device(), A.stream_ptr()
        seg = torch.empty(n_slices * m * 2, dtype=torch.int32, device=dev)
        f_scr = fn('be_scatter_plan_scratch_bytes', c_i64, [c_i64, c_i64, c_int, c_int])
        scratch = A.workspace(f_scr(m, k, slice_shift, slice_width))
        check(fn('be_scatter_plan_begin', c_int, [c_i64, c_i64, c_int, c_int, c_vp, c_i64, c_vp])(
            m, k, slice_shift, slice_width, A.ptr(scratch), scratch.numel(), st), 'be_scatter_plan_begin')
        f_cnt = fn('be_scatter_plan_count_rows', c_int,
                   [c_vp, c_vp, c_int, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_int, c_vp, c_vp, c_i64, c_vp, c_vp])
        seg_rows = seg.view(m, n_slices * 2)

        def block_args(r0, r1):
            w, idx, ptr = get_block(r0, r1)
            w, idx = A.to_device(w).reshape(-1), A.to_device(idx).reshape(-1)
            assert idx.dtype == torch.int32 and (w.numel() == 1) == bool(homo)
            ptr = None if ptr is None else A.to_device(ptr)
            assert ptr is None or ptr.numel() == r1 - r0 + 1
            is64 = int(ptr is not None and ptr.dtype == torch.int64)
            return w, idx, ptr, is64, (int(max_row_len) if ptr is None else -1)

        n_seen = 0
        for r0 in range(0, m, int(block_rows)):
            r1 = min(m, r0 + int(block_rows))
            w, idx, ptr, is64, rl = block_args(r0, r1)
            n_seen += int(idx.numel())
            check(f_cnt(A.ptr(idx), A.ptr(ptr), is64, rl, r1 - r0, m, k, slice_shift, slice_width, int(homo), lay,
                        A.ptr(seg_rows[r0]), A.ptr(scratch), scratch.numel(), None, st), 'be_scatter_plan_count_rows')
            torch.cuda.current_stream().synchronize()         # the block's arrays may be released by the caller's next get_block
        assert n_seen == int(nnz), f"build_from_blocks: the blocks hold {n_seen} entries, nnz says {nnz}"
        blob_bytes = c_i64(0)
        check(fn('be_scatter_plan_scan', c_int, [c_i64, c_i64, c_int, c_int, c_vp, c_vp, c_i64, ctypes.POINTER(c_i64), c_vp])(
            m, k, slice_shift, slice_width, A.ptr(seg), A.ptr(scratch), scratch.numel(), ctypes.byref(blob_bytes), st),
            'be_scatter_plan_scan')
        blob = torch.empty(int(blob_bytes.value) + 128, dtype=torch.uint8, device=dev)
        plan = cls(m, k, bool(homo), slice_shift, seg, blob, 0, weight_dtype, slice_width, lay)
        plan.nnz, plan.order, plan.row_len, plan.split_f64 = int(nnz), None, -1, False
        if lay == cls.LAYOUT_D8 and nnz:
            n_blk = m * n_slices
            ng = seg.view(n_blk, 2)[::max(1, n_blk >> 22), 1] & 0xffff
            plan.items_hint = max(1, min(1 << 20, int(round(4.0 * ng.double().mean().item()))))
        f_fill = fn('be_scatter_plan_fill_ordered', c_int,
                    [c_vp, c_int, c_int, c_vp, c_vp, c_int, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp])
        maxabs = torch.zeros(2, dtype=torch.int32, device=dev)
        stats = torch.tensor([0, 0xffffffff], dtype=torch.int64, device=dev)          # max |w| bits, smallest non-zero |w| bits
        for r0 in range(0, m, int(block_rows)):
            r1 = min(m, r0 + int(block_rows))
            w, idx, ptr, is64, rl = block_args(r0, r1)
            check(f_fill(A.ptr(w), int(homo), A.wcode(w), A.ptr(idx), A.ptr(ptr), is64, rl, r1 - r0, k, slice_shift, slice_width,
                         lay, A.ptr(seg_rows[r0]), A.ptr(blob), A.ptr(maxabs), None, st), 'be_scatter_plan_fill_ordered')
            mb = maxabs.to(torch.int64) & 0xffffffff                                   # (every fill call starts its own statistics)
            stats = torch.stack([torch.maximum(stats[0], mb[0]), torch.minimum(stats[1], mb[1])])
            torch.cuda.current_stream().synchronize()
        plan.stamp = None
        if not homo:
            both = torch.where(stats > 0x7fffffff, stats - (1 << 32), stats).to(torch.int32)     # back to the two uint32 bit patterns
            plan.scale_exp = plan._plan_exponent(both.contiguous())
        return plan

    def _fill(self, weights: torch.Tensor, indices: torch.Tensor, indptr: Optional[torch.Tensor], keep_exp: bool = False):
        """Write the blocks (``be_scatter_plan_fill``) and derive the fixed-point exponent of heterogeneous weights.  The
        segment table — block starts and lengths — depends on the structure only, so the same call refreshes the weights of
        an existing plan (:meth:`refresh_weights`)."""
        is64 = int(indptr is not None and indptr.dtype == torch.int64)
        maxabs = torch.zeros(2, dtype=torch.int32, device=self.seg.device)   # f32 bits of max |w| / smallest non-zero |w|
        name = 'be_scatter_plan_refresh_weights_ordered' if keep_exp else 'be_scatter_plan_fill_ordered'
        f_fill = fn(name, c_int,
                    [c_vp, c_int, c_int, c_vp, c_vp, c_int, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp])
        check(f_fill(A.ptr(weights), int(self.homo), A.wcode(weights), A.ptr(indices), A.ptr(indptr), is64, self.row_len,
                     self.m, self.k, self.slice_shift, self.slice_width, self.layout, A.ptr(self.seg), A.ptr(self.blob),
                     A.ptr(maxabs), A.ptr(getattr(self, 'order', None)), A.stream_ptr()), name)
        self.stamp = weights_stamp(weights)
        if not self.homo:
            # a refresh keeps the exponent it was built with while that still cannot overflow (it is a launch argument: a
            # captured graph replays with the old one) — unless the new weights need the finer resolution.  The column
            # statistics come from the plan's own blocks (be_scatter_plan_exponent: a planned step over |w| with every row
            # active) instead of two passes of global atomics over the raw entries.
            self.scale_exp = self._plan_exponent(maxabs, keep=self.scale_exp if keep_exp else None)

    def _plan_exponent(self, maxabs: torch.Tensor, keep: Optional[int] = None) -> int:
        scratch = A.workspace(256)
        out = c_int(0)
        f = fn('be_scatter_plan_exponent', c_int,
               [c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_int, c_i64, c_vp, c_int, c_int, c_vp, c_i64, ctypes.POINTER(c_int), c_vp])
        rc = f(A.ptr(self.blob), A.ptr(self.seg), self.m, self.k, self.slice_shift, self.slice_width, self.layout, self.nnz,
               A.ptr(maxabs), self.MIN_WEIGHT_BITS, -(1 << 31) if keep is None else int(keep), A.ptr(scratch), scratch.numel(),
               ctypes.byref(out), A.stream_ptr())
        if rc == -4:               # BE_ERR_RANGE: inf / nan or a dynamic range the sums cannot resolve -> the caller falls back
            from ._lib import lib
            raise MathError((lib().be_last_error() or b'').decode())
        check(rc, 'be_scatter_plan_exponent')
        return int(out.value)

    def refresh_weights(self, weights, indices, indptr) -> None:
        """Re-encode the weights of an unchanged structure into the existing blocks (the reference's cached workspace holds
        task ranges only, ``_csr/main.py:58-88``, so in-place weight updates are legal there; here the blocks embed the
        weights and have to follow them).  Raises ``MathError`` like :meth:`build` when the new weights do not qualify."""
        weights = A.to_device(weights).reshape(-1)
        indices = A.to_device(indices).reshape(-1)
        indptr = None if indptr is None else A.to_device(indptr)
        assert (weights.numel() == 1) == self.homo, "refresh_weights cannot switch between one weight and per-entry weights"
        src_stamp = weights_stamp(weights)
        if self.split_f64:
            assert weights.dtype == torch.float64
            weights, indices, indptr, _ = _split_f64(weights, indices, indptr, self.row_len // 2 if indptr is None else -1)
        assert weights.numel() == 1 or weights.numel() == self.nnz, "refresh_weights: the structure must be unchanged"
        self._fill(weights, indices, indptr, keep_exp=True)
        self.stamp = src_stamp

    def is_stale(self, weights: torch.Tensor) -> bool:
        """Heterogeneous plans embed the weights: true when ``weights`` was modified in place or replaced since the fill
        (one shared weight is read by the reduce kernel on every call and cannot go stale)."""
        return (not self.homo) and self.stamp != weights_stamp(weights)


def _split_f64(weights: torch.Tensor, indices: torch.Tensor, indptr: Optional[torch.Tensor], row_len: int):
    """Per-entry f64 weights for the f32 block layouts: ``w = hi + lo`` with ``hi = f32(w)``, ``lo = f32(w - hi)`` — two f32
    entries on the same column, 48 significant bits of ``w`` between them.  Each is converted to 64-bit fixed point exactly
    (``fixed_from_f32``), the integer sums are order independent, and the f64 output is their sum scaled once: accumulated
    currents are good to ~2^-48 of the column's weight scale, inside the 1e-10 bar of the f64 tests.  The reference's f64
    variants (``binary_csrmv_hybrid.cu:789-821``) add doubles with atomics; this chip's global atomics run at 21 G/s."""
    hi = weights.float()
    lo = (weights - hi.double()).float()
    w2 = torch.stack([hi, lo], dim=1).reshape(-1)
    idx2 = indices.repeat_interleave(2)
    ptr2 = None if indptr is None else indptr * 2
    return w2, idx2, ptr2, (2 * int(row_len) if indptr is None else -1)


def weights_stamp(t: torch.Tensor):
    """Identity + in-place modification counter of a weight tensor (what a cached workspace was derived from)."""
    return (t.data_ptr(), t._version, t.numel())


def fixed_point_exponent(weights: torch.Tensor, indices: Optional[torch.Tensor], k: int, keep: Optional[int] = None,
                         min_bits: Optional[int] = None) -> int:
    """Fixed-point exponent ``e`` of a weight array (``be_fixed_point_exponent``, chosen inside the library so that a
    non-Python binder can set up the same workspaces).

    Overflow bound: the largest ``e`` with ``max_j sum_{entries of column j} |w| * 2^e < 2^62`` — the largest *column* sum
    with every row active, not ``rows * max|w|``: a row may list a column several times (the reference sums duplicates), so
    a column can receive more addends than there are rows.  Without ``indices`` all the weights bound a column.
    Accuracy gate: a sum of ``n`` weights carries an absolute error below ``n * 2^-e``; ``e`` is accepted when the largest
    weight *of every non-empty output column* keeps ``ScatterPlan.MIN_WEIGHT_BITS`` bits (cheap sufficient test first: the
    globally smallest non-zero ``|w|`` does; otherwise — e.g. U[0,1) weights: 1e10 samples contain values down to 2^-32 —
    one scatter-max pass over the entries decides).  ``keep``: an exponent to keep if it still cannot overflow (a refresh
    of weights: captured graphs hold it as a launch argument).  Raises ``MathError`` for inf / nan weights or a dynamic range
    the 64-bit sums cannot resolve.  One or two passes of global atomics over the entries, build time only (~1 s at 1e10)."""
    flat_w = weights.reshape(-1)
    f_scr = fn('be_fixed_point_scratch_bytes', c_i64, [c_i64])
    scratch = A.workspace(f_scr(int(k)))
    out = c_int(0)
    f = fn('be_fixed_point_exponent', c_int,
           [c_vp, c_int, c_vp, c_i64, c_i64, c_int, c_int, c_vp, c_i64, ctypes.POINTER(c_int), c_vp])
    rc = f(A.ptr(flat_w), A.wcode(flat_w), A.ptr(None if indices is None else indices.reshape(-1)), flat_w.numel(), int(k),
           ScatterPlan.MIN_WEIGHT_BITS if min_bits is None else int(min_bits), -(1 << 31) if keep is None else int(keep),
           A.ptr(scratch), scratch.numel(),
           ctypes.byref(out), A.stream_ptr())
    if rc == -4:            # BE_ERR_RANGE: not representable
        from ._lib import lib
        raise MathError((lib().be_last_error() or b'').decode())
    check(rc, 'be_fixed_point_exponent')
    return int(out.value)


def fresh_scatter_workspace(ws, weights: torch.Tensor, indices: torch.Tensor, indptr: Optional[torch.Tensor]):
    """``ws`` brought up to date with ``weights``: a cached :class:`ScatterPlan` / :class:`BinnedScatter` is re-derived when
    the weight tensor was modified in place (``data.mul_``, ``data.copy_`` — plasticity) or replaced since it was built;
    ``None`` (the direct route) when the new weights do not qualify for the fixed-point routes."""
    if ws is None or not ws.is_stale(weights):
        return ws
    try:
        ws.refresh_weights(weights, indices, indptr)
    except MathError:
        return None
    return ws


def choose_scatter_route(nse: int, m: int, k: int, weights: torch.Tensor) -> str:
    """``'plan'``, ``'binned'`` or ``'direct'`` for a matrix of ``nse`` entries, ``m`` stored rows and ``k`` outputs.

    Measured on FixedNumPerPre K = 1000, 1 % firing (``tools/exp_plan_vs_binned.py``; entries per (row, slice) -> planned vs
    binned, us/step): homo 20: 51 / 87, 12: 87 / 119, 8: 169 / 148, 4: 845 / 198; hetero 16: 45 / 90, 10: 71 / 105,
    6: 158 / 162, 4: 306 / 236.  The planned layout pays per block (and 128 bytes of memory per block), the binned route per
    entry: they cross at 6-10 entries per block (round 1, before blocks were decoded by part of a wave each: 18)."""
    from . import _tuning
    _tuning.ensure_resolved()          # the persisted thresholds of the current device kind (first call only)
    if nse < PLAN_MIN_NNZ or m <= 0 or k <= 0:
        return 'direct'
    homo = weights.numel() == 1
    shift = ScatterPlan.default_shift(k, homo)
    if weights.dtype == torch.float64 and not homo:
        nse = 2 * nse                     # stored as two f32 entries each (_split_f64)
    n_slices = -(-k // ScatterPlan.auto_geometry(m, k, nse, homo, shift)[1])
    per_block = nse / (m * n_slices)
    if n_slices <= 4096 and per_block >= (PLAN_MIN_SEGMENT_HOMO if homo else PLAN_MIN_SEGMENT):
        return 'plan'
    if BinnedScatter.applicable(weights, k):
        return 'binned'
    if n_slices <= 4096 and per_block >= PLAN_MIN_SEGMENT_NO_BINNED:
        return 'plan'
    return 'direct'


PLAN_KEEP_ORDER: Optional[bool] = None    # ScatterPlan.build(keep_order=...) of the plans the containers build by themselves
                                          # (True: weights that change in place — plasticity — get a gather-copy refresh)


#: a matrix with at least this many stored entries that ends up on the direct route (global float atomics: ~21 Geff/s on this
#: chip, 35x below the planned route at C2) says so once per cause
DIRECT_ROUTE_WARN_NNZ = 1 << 24


def _warn_direct(nse: int, why: str) -> None:
    if nse >= DIRECT_ROUTE_WARN_NNZ:
        import warnings
        warnings.warn(f"brainevent_amd: a matrix of {nse} stored entries runs its scatter product on the direct route (global float "
                      f"atomics, ~21 G updates/s) because {why}; the fixed-point routes (plan / binned) are 10-40x faster.",
                      stacklevel=3)


def make_scatter_workspace(route: str, weights, indices, indptr, m: int, k: int, nse: int, row_len: int = -1):
    """The scatter workspace of a matrix for the chosen ``route`` (``None`` = direct route).  A plan that does not fit the
    free device memory (128 bytes per (row, slice) block: up to ~3x the raw matrix for short blocks) falls back to the binned
    route, which needs no per-matrix layout; non-finite weights or an extreme dynamic range fall back to the direct route —
    with a warning for large matrices (``DIRECT_ROUTE_WARN_NNZ``): that is a 35x cliff at C2."""
    oom = getattr(torch, 'OutOfMemoryError', getattr(torch.cuda, 'OutOfMemoryError', RuntimeError))
    if route == 'direct':
        _warn_direct(nse, "neither the planned layout nor the binned route applies to its shape / weight dtype")
        return None
    try:
        if route == 'plan':
            try:
                return ScatterPlan.build(weights, indices, indptr, shape=(m, k), row_len=row_len, keep_order=PLAN_KEEP_ORDER)
            except oom:
                torch.cuda.empty_cache()
                route = 'binned' if BinnedScatter.applicable(weights, k) else 'direct'
        if route == 'binned':
            return BinnedScatter(weights, m, k, nse, indices=indices, indptr=indptr, row_len=row_len)
    except MathError as e:
        # inf / nan / extreme dynamic range: float atomics (direct route) handle those
        _warn_direct(nse, f"its weights do not qualify for fixed-point sums ({e})")
        return None
    except oom:
        torch.cuda.empty_cache()
        _warn_direct(nse, "no workspace fits the free device memory")
    return None


import weakref
_LIVE_BINNED: 'weakref.WeakSet' = weakref.WeakSet()


def check_binned_status(clear: bool = True) -> None:
    """:meth:`BinnedScatter.check_status` of every live binned workspace: raises ``KernelExecutionError`` if any step since the
    last call gave up on its append protocol (its outputs were NaN) or broke the conservation of its entries.  Synchronises every
    device that holds a workspace (all streams: a replay in flight on another stream is waited for, not read mid-step) — call it
    where the program synchronises anyway (after a batch of replays of a captured graph, at the end of an epoch, in a test)."""
    for ws in list(_LIVE_BINNED):
        ws.check_status(clear=clear)


class BinnedScatter:
    """Workspace of the *binned* scatter route: no per-matrix layout, only per-slice bins that are refilled every call
    (``be_binary_csrmv_t_binned``).  Used when a matrix is large but a :class:`ScatterPlan` does not pay — fewer than
    ``PLAN_MIN_SEGMENT`` entries per (row, slice), e.g. ``FixedNumPerPre`` with K = 1000 over 10M outputs.

    ``bin_capacity`` (entries per slice bin) is a tuning knob: bins that overflow are delivered through global atomics
    (correct, slower).  It is sized for ``max_active_fraction`` of the rows firing in one step.
    """

    def __init__(self, weights: torch.Tensor, m: int, k: int, nnz: int, *, max_active_fraction: float = 0.05,
                 slice_shift: Optional[int] = None, indices: Optional[torch.Tensor] = None, acc32: Optional[bool] = None,
                 indptr: Optional[torch.Tensor] = None, row_len: int = -1):
        self.m, self.k = int(m), int(k)
        self.homo = weights.numel() == 1
        if slice_shift is None and os.environ.get('BE_BIN_SHIFT'):      # A/B runs
            slice_shift = int(os.environ['BE_BIN_SHIFT'])
        # bins at most 2^slice_shift columns wide; the default leaves the width to the library: as wide as the LDS
        # accumulators of pass C allow, in a multiple of 256 bins (be_binned_bins)
        self.slice_shift = 16 if slice_shift is None else int(slice_shift)
        from . import _tuning
        _tuning.push_to_library()            # pass B's task size from the persisted tuning of this architecture
        self.max_active_fraction = float(max_active_fraction)
        self.scale_exp = 0
        self.nnz = int(nnz)
        self.acc32 = False
        # the row structure, when the caller has it: the column statistics then come from binned steps over all the rows
        # (_column_stats_by_steps) instead of passes of global atomics over the entries
        self._rows = (indptr, int(row_len)) if (indptr is not None or row_len >= 0) else None
        self._derive_exponent(weights, indices, acc32=acc32)
        self._set_geometry()
        self._ws: Dict = {}
        self.ws = self.workspace(1)
        _LIVE_BINNED.add(self)

    #: 32-bit fixed-point sums (twice the bin width: half the bins, one round of pass C; C4: 0.60 -> 0.53 ms per step).
    #: ``acc32=True`` (explicit): taken when every column's largest weight keeps ACC32_MIN_WEIGHT_BITS bits at the 32-bit exponent
    #: — an addend is then rounded by at most 2^-19 of its column's largest weight, i.e. an output is good to
    #: ``n_addends * 2^-19 * w_max(column)``: inside rtol = atol = 1e-5 for weights of one scale (C4, U[0,1): 1e-7 of a typical
    #: output), but an output made of ONE small weight (1e-3 of the column's largest) carries up to 5e-4 relative — coarser than the
    #: reference's f32 atomics there.  So it is never chosen silently for such weights: AUTOMATICALLY (``acc32=None``) the 32-bit
    #: sums are taken only when the SMALLEST non-zero |w| of the whole matrix keeps ACC32_AUTO_SMALLEST_BITS bits at that exponent —
    #: every single addend, hence every output of same-sign weights, is then within 2^-18 = 4e-6 relative, the 1e-5 bar of the path.
    ACC32_MIN_WEIGHT_BITS: Optional[int] = 18
    ACC32_AUTO_SMALLEST_BITS = 17
    #: ... and only over at least this many outputs (below, the 64-bit bins already are one round of pass C)
    ACC32_MIN_OUTPUTS = 256 * 20000
    #: stored entries from which the column statistics behind the exponent come from binned steps (below: one atomic pass is faster)
    STATS_BY_STEPS_MIN_NNZ = 1 << 24

    @property
    def kind(self) -> int:
        """The `homo` argument of the binned entry points: 0 per-entry weights / 64-bit sums, 1 one shared weight, 2 per-entry
        weights / 32-bit sums (BE_BINNED_ACC32)."""
        return 1 if self.homo else (2 if self.acc32 else 0)

    @staticmethod
    def serves(k: int, slice_shift: int = 16, homo: bool = False) -> bool:
        """Whether the binned route has a geometry for ``k`` outputs at this ``slice_shift`` (``be_binned_bins`` > 0: the
        write-combining blocks of all bins have to fit pass B's LDS)."""
        return int(fn('be_binned_bins', c_int, [c_i64, c_int, c_int])(int(k), int(slice_shift), int(bool(homo)))) > 0

    SHORT_ROW_ENTRIES = 256      # BE_BINNED_SHORT_ROWS: average stored row length up to which pass B runs one step ahead

    @property
    def step_kind(self) -> int:
        """``kind`` plus the hints of a step call: ``BE_BINNED_SHORT_ROWS`` (8) when the stored rows average at most
        ``SHORT_ROW_ENTRIES`` entries (the library cannot know the entry count behind an indptr without reading it back)."""
        return self.kind | (8 if self.nnz <= self.SHORT_ROW_ENTRIES * max(self.m, 1) else 0)

    def _set_geometry(self) -> None:
        self.n_slices = int(fn('be_binned_bins', c_int, [c_i64, c_int, c_int])(self.k, self.slice_shift, self.kind))
        if self.n_slices <= 0:
            raise ValueError(f"the binned route does not serve {self.k} outputs at slice_shift={self.slice_shift}")
        expect = self.max_active_fraction * self.nnz / max(self.n_slices, 1)
        self.bin_capacity = int(max(1024, min(2 ** 31, 1.25 * expect + 6 * math.sqrt(max(expect, 1.0)) + 64)))

    def workspace(self, n_batch: int = 1) -> torch.Tensor:
        """The workspace of steps over ``n_batch`` spike vectors (created and initialised once per batch size; never evicted:
        a captured HIP graph keeps its raw pointer)."""
        ws = self._ws.get(int(n_batch))
        if ws is None:
            f = fn('be_binary_csrmm_t_binned_workspace_bytes', c_i64, [c_i64, c_i64, c_i64, c_int, c_i64])
            ws = A.workspace(f(self.m, self.k, int(n_batch), self.slice_shift, self.bin_capacity))      # (sized for every kind)
            f = fn('be_binary_csrmm_t_binned_workspace_init', c_int, [c_vp, c_i64, c_i64, c_i64, c_i64, c_int, c_i64, c_vp])
            check(f(A.ptr(ws), ws.numel(), self.m, self.k, int(n_batch), self.slice_shift, self.bin_capacity, A.stream_ptr()),
                  'be_binary_csrmm_t_binned_workspace_init')
            self._ws[int(n_batch)] = ws
        return ws

    def audit(self, n_batch: int = 1):
        """The four conservation counters of the workspace for ``n_batch`` vectors since the last clear
        (``be_binned_workspace_audit``): entries in the active rows, tickets drawn by pass B, entries accumulated by pass C,
        entries delivered through the overflow image — ``a == b == c + d`` after complete steps.  Synchronises."""
        c = (ctypes.c_uint64 * 4)()
        check(fn('be_binned_workspace_audit', c_int, [c_vp, ctypes.POINTER(ctypes.c_uint64), c_vp])(
            A.ptr(self.workspace(n_batch)), c, A.stream_ptr()), 'be_binned_workspace_audit')
        return tuple(int(x) for x in c)

    def check_status(self, clear: bool = True) -> None:
        """Raise ``KernelExecutionError`` if a step on one of this object's workspaces gave up on its append protocol (such a
        step wrote NaN outputs instead of trapping the device) or if the workspace's conservation counters disagree — an entry
        lost or delivered twice between the row bounds and the accumulators (``be_binned_workspace_status``).  Synchronises the
        WHOLE device each workspace lives on before reading it (a step or a graph replay still running on another stream would be
        read between its passes — a spurious conservation error — and ``clear`` would zero the counters under it), and reads a
        workspace through its own device; call it at a point that synchronises anyway (the containers do after a mirror build;
        ``bench.py`` at its parity check)."""
        f = fn('be_binned_workspace_status', c_int, [c_vp, c_int, c_vp])
        for ws in self._ws.values():
            with torch.cuda.device(ws.device):
                torch.cuda.synchronize(ws.device)
                check(f(A.ptr(ws), int(clear), A.stream_ptr()), 'be_binned_workspace_status')

    def _column_stats_by_steps(self, weights: torch.Tensor, indices: torch.Tensor):
        """``(largest column sum of |w|, smallest column mean of |w| over the non-empty columns, max |w|, min non-zero |w|)``
        computed by the route itself: the rows are walked in chunks that fit the bins, every chunk once as a binned step over
        ``|w|`` (``BE_BINNED_ABS``, 64-bit sums at a provisional exponent that cannot overflow) and once as a counted step — LDS
        integer sums instead of one global float atomic per stored entry (C4, 1e10 entries: ~0.1 s instead of 0.47 s for the
        sums + 0.37 s for the per-column maxima).  The mean bounds a column's largest weight from below, which is what the
        accuracy gate needs; the exact maxima are only fetched (by the atomic pass) when that sufficient test fails."""
        indptr, row_len = self._rows
        m, k, dev = self.m, self.k, weights.device
        flat = weights.reshape(-1)
        mx, mn = ctypes.c_uint32(0), ctypes.c_uint32(0)
        scr = A.workspace(256)
        check(fn('be_weight_stats', c_int, [c_vp, c_int, c_i64, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), c_vp,
                                            c_i64, c_vp])(A.ptr(flat), A.wcode(flat), flat.numel(), ctypes.byref(mx), ctypes.byref(mn),
                                                          A.ptr(scr), scr.numel(), A.stream_ptr()), 'be_weight_stats')
        if mx.value >= 0x7f800000:
            raise MathError("weights contain inf / nan: the fixed-point routes do not apply")
        as_f32 = lambda b: float(np.array([b], dtype=np.uint32).view(np.float32)[0])
        wmax = as_f32(mx.value)
        wmin = as_f32(mn.value) if mn.value != 0xffffffff else float('inf')
        if wmax == 0.0:
            return 0.0, float('inf'), 0.0, float('inf')
        e0 = 62 - math.frexp(wmax * (self.nnz + 1) * 1.001)[1]             # every entry in one column could not overflow this
        e0 = max(-90, min(150, e0))
        n_slices = int(fn('be_binned_bins', c_int, [c_i64, c_int, c_int])(k, self.slice_shift, 0))
        if n_slices <= 0:
            raise ValueError(f"the binned route does not serve {k} outputs at slice_shift={self.slice_shift}")
        expect = self.max_active_fraction * self.nnz / n_slices
        cap = int(max(1024, min(2 ** 31, 1.25 * expect + 6 * math.sqrt(max(expect, 1.0)) + 64)))
        f_bytes = fn('be_binary_csrmm_t_binned_workspace_bytes', c_i64, [c_i64, c_i64, c_i64, c_int, c_i64])
        ws = A.workspace(f_bytes(m, k, 1, self.slice_shift, cap))
        check(fn('be_binary_csrmm_t_binned_workspace_init', c_int, [c_vp, c_i64, c_i64, c_i64, c_i64, c_int, c_i64, c_vp])(
            A.ptr(ws), ws.numel(), m, k, 1, self.slice_shift, cap, A.stream_ptr()), 'be_binary_csrmm_t_binned_workspace_init')
        f = fn('be_binary_csrmv_t_binned', c_int,
               [c_vp, c_int, c_int, c_vp, c_vp, c_int, c_i64, c_vp, c_int, c_vp, c_i64, c_i64, c_int, c_i64, c_int, c_vp, c_i64, c_vp])
        is64 = int(indptr is not None and indptr.dtype == torch.int64)
        one = torch.ones(1, dtype=torch.float32, device=dev)
        chunk = max(1, int(self.max_active_fraction * m))       # rows per step: what the bins are sized for (past it: the overflow image, still |w|)
        ids = torch.arange(m, dtype=torch.int32, device=dev)
        colsum = torch.zeros(k, dtype=torch.float32, device=dev)
        count = torch.zeros(k, dtype=torch.float32, device=dev)
        out = torch.empty(k, dtype=torch.float32, device=dev)
        wcode = A.wcode(flat)
        for lo in range(0, m, chunk):
            n_act = torch.tensor([min(chunk, m - lo)], dtype=torch.int32, device=dev)
            ev = A.ActiveIds(ids[lo:lo + chunk], n_act, m)
            for kind, w_arg, code, acc in ((4, flat, wcode, colsum), (1, one, A.BE_F32, count)):
                check(f(A.ptr(w_arg), kind, code, A.ptr(indices), A.ptr(indptr), is64, row_len, A.ptr(ev), A.BE_SPIKE_IDS, A.ptr(out),
                        m, k, self.slice_shift, cap, e0, A.ptr(ws), ws.numel(), A.stream_ptr()), 'be_binary_csrmv_t_binned')
                acc += out
        # the statistics decide the exponent of every later step: the steps that made them must have conserved their entries
        # (the reductions below synchronise anyway)
        check(fn('be_binned_workspace_status', c_int, [c_vp, c_int, c_vp])(A.ptr(ws), 1, A.stream_ptr()), 'be_binned_workspace_status')
        live = count > 0
        mean_min = float((colsum[live] / count[live]).min()) if bool(live.any()) else float('inf')
        return float(colsum.max()), mean_min, wmax, wmin

    def _exponent_by_steps(self, weights, indices, min_bits: int, keep: Optional[int]) -> int:
        """``be_fixed_point_exponent``'s answer (same bound, same gate, same ``keep`` rule) from :meth:`_column_stats_by_steps`."""
        if getattr(self, '_stats_stamp', None) != weights_stamp(weights):
            self._stats = self._column_stats_by_steps(weights, indices)
            self._stats_stamp = weights_stamp(weights)
        colsum_max, mean_min, wmax, wmin = self._stats
        need = 62 - (math.frexp(colsum_max * 1.001)[1] if colsum_max > 0 else 0)
        need = max(-90, min(150, need))
        for e in ([keep] if keep is not None and keep <= need else []) + [need]:
            thr = math.ldexp(1.0, min_bits - e)
            if wmin >= thr or mean_min >= thr:         # every column's largest weight >= its mean >= thr
                return int(e)
        # the sufficient tests failed: the exact per-column maxima decide (atomic pass of the library)
        return fixed_point_exponent(weights, indices, self.k, keep=keep, min_bits=min_bits)

    def _exponent(self, weights, indices, min_bits: Optional[int] = None, keep: Optional[int] = None) -> int:
        mb = ScatterPlan.MIN_WEIGHT_BITS if min_bits is None else int(min_bits)
        if self._rows is not None and indices is not None and weights.dtype in (torch.float32, torch.float16, torch.bfloat16) \
                and self.nnz >= self.STATS_BY_STEPS_MIN_NNZ and not os.environ.get('BE_BIN_ATOMIC_STATS'):
            return self._exponent_by_steps(weights, indices, mb, keep)
        return fixed_point_exponent(weights, indices, self.k, keep=keep, min_bits=mb)

    def _derive_exponent(self, weights: torch.Tensor, indices: Optional[torch.Tensor], keep_exp: bool = False,
                         acc32: Optional[bool] = None) -> None:
        """The fixed-point exponent of per-entry weights — and, at construction (``keep_exp=False``), whether the sums are 32 or
        64 bits wide: 32 when ``ACC32_MIN_WEIGHT_BITS`` allows it (``acc32=None``), or as forced.  A refresh keeps the width (the
        bins and the workspace were sized for it); weights that no longer qualify for 32-bit sums raise ``MathError`` and the
        container rebuilds its workspace."""
        self.stamp = weights_stamp(weights)
        if self.homo:
            return
        b32 = (self.ACC32_MIN_WEIGHT_BITS or 18) + 32
        if keep_exp:
            if self.acc32:
                try:
                    self.scale_exp = self._exponent(weights, indices, b32, keep=self.scale_exp + 32) - 32
                except MathError:       # no longer fine enough for 32-bit sums: back to 64-bit bins (new geometry, new workspaces)
                    self.scale_exp = self._exponent(weights, indices)
                    self.acc32 = False
                    self._set_geometry()
                    self._ws = {}
                    self.ws = self.workspace(1)
            else:
                self.scale_exp = self._exponent(weights, indices, keep=self.scale_exp)
            return
        want32 = acc32 if acc32 is not None else (self.ACC32_MIN_WEIGHT_BITS is not None and self.k >= self.ACC32_MIN_OUTPUTS
                                                  and weights.dtype == torch.float32 and not os.environ.get('BE_BIN_NO_ACC32'))
        if want32:
            try:        # the 64-bit exponent e leaves 2^62 of headroom; the same bound for 2^30 is e - 32
                e32 = self._exponent(weights, indices, b32) - 32
                if acc32 or self._smallest_weight(weights) >= math.ldexp(1.0, self.ACC32_AUTO_SMALLEST_BITS - e32):
                    self.acc32, self.scale_exp = True, e32
                    return
            except MathError:
                self.acc32 = False
                if acc32:
                    raise
        self.acc32 = False
        self.scale_exp = self._exponent(weights, indices)

    def _smallest_weight(self, weights: torch.Tensor) -> float:
        """Smallest non-zero |w| (inf if there is none): from the statistics already taken, else one streaming pass."""
        if getattr(self, '_stats_stamp', None) == weights_stamp(weights):
            return self._stats[3]
        flat = weights.reshape(-1)
        mx, mn = ctypes.c_uint32(0), ctypes.c_uint32(0)
        scr = A.workspace(256)
        check(fn('be_weight_stats', c_int, [c_vp, c_int, c_i64, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), c_vp,
                                            c_i64, c_vp])(A.ptr(flat), A.wcode(flat), flat.numel(), ctypes.byref(mx), ctypes.byref(mn),
                                                          A.ptr(scr), scr.numel(), A.stream_ptr()), 'be_weight_stats')
        return float(np.array([mn.value], dtype=np.uint32).view(np.float32)[0]) if mn.value != 0xffffffff else float('inf')

    def refresh_weights(self, weights, indices, indptr=None) -> None:
        """The bins are refilled from the matrix on every call; only the fixed-point exponent derives from the weights."""
        weights = A.to_device(weights).reshape(-1)
        assert (weights.numel() == 1) == self.homo, "refresh_weights cannot switch between one weight and per-entry weights"
        if indptr is not None and self._rows is None:
            self._rows = (A.to_device(indptr), -1)
        # a refresh means "the weights changed": the cached column statistics are void even when the tensor's stamp did not move —
        # Mirror.refreshed / _fresh_indexed_workspace rewrite the same tensor through a raw pointer (gather_by_perm(out=...)), which
        # torch's version counter does not see (a grown weight could wrap the int64 sums, a shrunk one lose the 1e-5 gate)
        self._stats_stamp = None
        self._derive_exponent(weights, None if indices is None else A.to_device(indices).reshape(-1), keep_exp=True)

    def is_stale(self, weights: torch.Tensor) -> bool:
        return (not self.homo) and self.stamp != weights_stamp(weights)

    @staticmethod
    def applicable(weights: torch.Tensor, k: int) -> bool:
        if weights.dtype not in (torch.float32, torch.float16, torch.bfloat16):
            return False                  # f64: the bins carry f32 weights (per-entry f64 weights take the planned route)
        return fn('be_binned_bins', c_int, [c_i64, c_int, c_int])(int(k), 16, int(weights.numel() == 1)) > 0      # (kind 0 / 1)


def _binned_call(ws: 'BinnedScatter', weights, indices, indptr, row_len, spikes, sd, out) -> None:
    f = fn('be_binary_csrmv_t_binned', c_int,
           [c_vp, c_int, c_int, c_vp, c_vp, c_int, c_i64, c_vp, c_int, c_vp, c_i64, c_i64, c_int, c_i64, c_int, c_vp, c_i64,
            c_vp])
    is64 = int(indptr is not None and indptr.dtype == torch.int64)
    check(f(A.ptr(weights), ws.step_kind, A.wcode(weights), A.ptr(indices), A.ptr(indptr), is64, row_len, A.ptr(spikes), sd,
            A.ptr(out), ws.m, ws.k, ws.slice_shift, ws.bin_capacity, ws.scale_exp, A.ptr(ws.ws), ws.ws.numel(),
            A.stream_ptr()), 'be_binary_csrmv_t_binned')


def binned_batch(ws: 'BinnedScatter', weights, indices, indptr, row_len, spikes_bm, sd, out_bm) -> None:
    """The binned step for a batch ``spikes_bm [n_batch, m]`` -> ``out_bm [n_batch, k]`` (``be_binary_csrmm_t_binned``: the rows
    with a spike in any batch row are read once for up to 32 batch rows at a time where the bins of all of them fit pass B's
    LDS, else one step per batch row as the reference does, ``binary_csrmm_hybrid.cu:16-57``).  An id list is a single vector."""
    if sd == A.BE_SPIKE_IDS or spikes_bm.ndim == 1:
        _binned_call(ws, weights, indices, indptr, row_len, spikes_bm, sd, out_bm)
        return
    nb = int(out_bm.shape[0])
    wsb = ws.workspace(nb)
    f = fn('be_binary_csrmm_t_binned', c_int,
           [c_vp, c_int, c_int, c_vp, c_vp, c_int, c_i64, c_vp, c_int, c_vp, c_i64, c_i64, c_i64, c_int, c_i64, c_int, c_vp, c_i64,
            c_vp])
    is64 = int(indptr is not None and indptr.dtype == torch.int64)
    check(f(A.ptr(weights), ws.kind, A.wcode(weights), A.ptr(indices), A.ptr(indptr), is64, row_len, A.ptr(spikes_bm), sd,
            A.ptr(out_bm), ws.m, ws.k, nb, ws.slice_shift, ws.bin_capacity, ws.scale_exp, A.ptr(wsb), wsb.numel(),
            A.stream_ptr()), 'be_binary_csrmm_t_binned')


def _plan_call(plan: ScatterPlan, weights: torch.Tensor, spikes_bm: torch.Tensor, sd: int, out_bm: torch.Tensor,
               parts: Optional[int] = None) -> None:
    """Planned scatter for a batch: ``spikes_bm`` is ``[n_batch, m]`` (or ``[m]``), ``out_bm`` ``[n_batch, k]``."""
    nb = 1 if spikes_bm.ndim == 1 else int(spikes_bm.shape[0])
    parts = plan.default_parts() if parts is None else int(parts)
    if nb > 1:
        parts = max(1, min(parts, 512 // (plan.n_slices * nb)))
    ws = plan.workspace(parts, nb)
    f = fn('be_binary_csrmm_t_plan', c_int,
           [c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_int, c_int, c_int, c_vp,
            c_i64, c_vp])
    check(f(A.ptr(weights), int(plan.homo), A.wcode(out_bm), A.ptr(plan.blob), A.ptr(plan.seg), A.ptr(spikes_bm), sd,
            A.ptr(out_bm), plan.m, plan.k, nb, plan.slice_shift, plan.slice_width, plan.layout, plan.block_hint, parts,
            plan.scale_exp, A.ptr(ws),
            ws.numel(),
            A.stream_ptr()), 'be_binary_csrmm_t_plan')


# =====================================================================================================
# the reference's task workspace, kept as a name-compatible handle
# =====================================================================================================
#: the two constants of the reference's hybrid scheduler that size its task queue (``brainevent/_csr/hybrid_config.py:77-88``)
HYBRID_TPR_THRESHOLD, HYBRID_TASK_NNZ = 128, 4096


def hybrid_task_capacity(indptr) -> int:
    """Task-queue capacity of the reference's hybrid kernel for ``indptr`` (``brainevent/_csr/hybrid_config.py:298-324``):
    ``sum over rows longer than 128 of ceil(len / 4096)``, same validation and errors.  Nothing here consumes the number — the
    per-matrix workspace of this build is a :class:`ScatterPlan` / :class:`BinnedScatter` — it exists so that code written
    against the reference (``_make_binary_csrmv_workspace(indptr)``, ``workspace.task_capacity``) keeps working."""
    ptr = indptr.detach().to('cpu').numpy() if isinstance(indptr, torch.Tensor) else np.asarray(indptr)
    ptr = ptr.astype(np.int64)
    if ptr.ndim != 1:
        raise ValueError(f"indptr must be one-dimensional, got shape={ptr.shape}.")
    if ptr.size == 0:
        raise ValueError("indptr must contain at least one element.")
    row_lengths = np.diff(ptr)
    if np.any(row_lengths < 0):
        raise ValueError("CSR row lengths must be non-negative.")
    chunks = np.where(row_lengths > HYBRID_TPR_THRESHOLD, (row_lengths + HYBRID_TASK_NNZ - 1) // HYBRID_TASK_NNZ, 0)
    cap = int(chunks.sum())
    if cap > np.iinfo(np.int32).max:
        raise ValueError("binary task capacity exceeds int32 range.")
    return cap


class _BinaryCsrmvTaskWorkspace:
    """The reference's explicit task workspace ``(task_capacity, task_begin, task_end, status)``
    (``brainevent/_csr/binary.py:76-120``, ``_csr/main.py:58-88``) as a handle: the four fields exist with the reference's
    shapes and dtypes (``task_begin`` / ``task_end``: ``(task_capacity,)`` in the dtype of ``indptr``; ``status``: int32
    ``(2,)``) so that code which builds, passes, unpacks or checks one keeps working; the kernels of this build do not read
    them.  Passed as ``workspace=`` to ``binary_csrmv`` / ``binary_csrmm`` it is where the matrix's real scatter workspace is
    cached after the first call — keyed on the arrays it was derived from, so one handle reused for another matrix of the same
    ``indptr`` (legal in the reference: its queue depends on ``indptr`` alone) re-derives it."""
    __slots__ = ('task_capacity', 'task_begin', 'task_end', 'status', '_native', '_native_key')

    def __init__(self, task_capacity, task_begin, task_end, status):
        self.task_capacity, self.task_begin, self.task_end, self.status = int(task_capacity), task_begin, task_end, status
        self._native, self._native_key = None, None

    def __iter__(self):
        return iter((self.task_capacity, self.task_begin, self.task_end, self.status))

    def block_until_ready(self):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        return self

    def native(self, weights, indices, indptr, shape):
        """The scatter workspace (plan / binned / None = direct) of ``(weights, indices, indptr)``, cached on this handle."""
        key = (indices.data_ptr(), indptr.data_ptr(), int(indices.numel()), tuple(int(x) for x in shape), weights.numel() == 1,
               weights.dtype)
        if self._native_key != key:
            m, k = int(shape[0]), int(shape[1])
            nse = int(indices.numel())
            self._native = make_scatter_workspace(choose_scatter_route(nse, m, k, weights), weights, indices, indptr, m, k, nse)
            self._native_key = key
        else:
            self._native = fresh_scatter_workspace(self._native, weights, indices, indptr)
        return self._native


_BinaryTaskWorkspace = _BinaryCsrmvTaskWorkspace          # the name of the containers' twin (``_csr/main.py:58-88``)


def _make_binary_csrmv_workspace(indptr) -> _BinaryCsrmvTaskWorkspace:
    """Reference ``brainevent/_csr/binary.py:108-120``: an empty task workspace sized by :func:`hybrid_task_capacity`."""
    cap = hybrid_task_capacity(indptr)
    if isinstance(indptr, torch.Tensor):
        mk = lambda n, dt: torch.empty(n, dtype=dt, device=indptr.device)
        return _BinaryCsrmvTaskWorkspace(cap, mk(cap, indptr.dtype), mk(cap, indptr.dtype), mk(2, torch.int32))
    dt = np.asarray(indptr).dtype
    return _BinaryCsrmvTaskWorkspace(cap, np.empty(cap, dt), np.empty(cap, dt), np.empty(2, np.int32))


def _make_binary_csrmv_benchmark_workspace(indptr) -> _BinaryCsrmvTaskWorkspace:
    return _make_binary_csrmv_workspace(indptr)


_make_binary_task_workspace = _make_binary_csrmv_workspace


def _resolve_workspace(workspace, weights, indices, indptr, shape, transpose):
    """``workspace=`` of the functional ops: a native workspace as it is; a reference-style task handle -> the native workspace
    cached on it (scatter direction only: the gather direction needs none); anything else: the direct route."""
    if isinstance(workspace, _BinaryCsrmvTaskWorkspace):
        return workspace.native(weights, indices, indptr, shape) if transpose else None
    return workspace


def _task_operands(workspace):
    """The three task arrays a reference ``*_p_call`` returns next to the result (``brainevent/_csr/binary.py:980-987``)."""
    if isinstance(workspace, _BinaryCsrmvTaskWorkspace):
        return workspace.task_begin, workspace.task_end, workspace.status
    return None, None, None


# =====================================================================================================
# functional ops
# =====================================================================================================
class PlannedMatrix:
    """``events @ M`` through a :class:`ScatterPlan` alone: the raw CSR arrays are not kept (a plan built with
    :meth:`ScatterPlan.build_from_blocks`, or the plan of a container whose arrays the caller wants to release).  Only the
    event-driven scatter product exists — vectors and batches, every spike encoding; everything that needs the raw arrays
    (the gather direction, ``todense``, a weight refresh) does not.  One shared weight is passed as ``weight``."""

    def __init__(self, plan: ScatterPlan, weight=None):
        if plan.homo and weight is None:
            raise ValueError("PlannedMatrix: a plan of one shared weight needs that weight.")
        self.plan = plan
        self.shape = (plan.m, plan.k)
        self.weight = None if weight is None else A.to_device(torch.as_tensor(weight)).reshape(-1)[:1].to(plan.weight_dtype)

    def __rmatmul__(self, other):
        if not is_event(other):
            raise NotImplementedError("only event operands (BinaryArray, BitPackedBinary, CompactBinary) are served.")
        v = _event_value(other, scatter=True)
        if v.ndim not in (1, 2):
            raise NotImplementedError(f"matmul with object of shape {v.shape}")
        if v.shape[-1] != self.plan.m:
            raise MathError(f"shapes {tuple(v.shape)} and {self.shape} not aligned.")
        sp, sd = A.spikes_to_device(v)
        spikes_bm = sp.reshape(1, -1) if v.ndim == 1 else sp.contiguous()          # batch-major [n_batch, m] (packed: words)
        out = torch.empty((1 if v.ndim == 1 else int(v.shape[0]), self.plan.k), dtype=self.plan.weight_dtype, device=A.device())
        w = self.weight if self.plan.homo else torch.empty(0, dtype=self.plan.weight_dtype, device=A.device())
        _plan_call(self.plan, w, spikes_bm, sd, out)
        r = out[0] if v.ndim == 1 else out
        return r.cpu().numpy() if A.wants_numpy(v) else r


def _variant(homo: bool, w: torch.Tensor, sd: int) -> str:
    return f"{'homo' if homo else 'hetero'}_{A.wsuffix(w)}_{'bool' if sd == A.BE_SPIKE_BOOL else 'float'}"


_CSRMM_ARGS = [c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp]


_CSRMM_GENERIC_ARGS = [c_vp, c_int, c_int, c_vp, c_vp, c_int, c_i64, c_vp, c_int, c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp]


def _csrmm_generic(weights, indices, indptr, row_len, spikes_bm, sd, out, m, k, nb, ws, transpose) -> None:
    """``be_binary_csrmm_{t,nt}`` with explicit dtype codes (any spike encoding; ``indptr=None`` = fixed row length)."""
    name = 'be_binary_csrmm_t' if transpose else 'be_binary_csrmm_nt'
    f = fn(name, c_int, _CSRMM_GENERIC_ARGS)
    is64 = int(indptr is not None and indptr.dtype == torch.int64)
    check(f(A.ptr(weights), int(weights.numel() == 1), A.wcode(weights), A.ptr(indices), A.ptr(indptr), is64, row_len,
            A.ptr(spikes_bm), sd, A.ptr(out), m, k, nb, A.ptr(ws), ws.numel(), A.stream_ptr()), name)


def _csr_batched(weights, indices, indptr, spikes_bm, sd, *, shape, transpose, workspace=None):
    """Run the CSR kernels on a batch-major spike matrix ``[n_batch, len]`` -> ``[n_batch, out_len]``."""
    m, k = int(shape[0]), int(shape[1])
    nb = int(spikes_bm.shape[0])
    homo = weights.numel() == 1
    out_len = k if transpose else m
    out = torch.empty((nb, out_len), dtype=weights.dtype, device=weights.device)
    if out_len == 0 or nb == 0:
        return out
    if m == 0 or k == 0 or indices.numel() == 0:
        return out.zero_()
    is64 = int(indptr.dtype == torch.int64)
    if transpose:
        if isinstance(workspace, ScatterPlan):
            assert workspace.m == m and workspace.k == k, "workspace was built for another matrix shape"
            _plan_call(workspace, weights, spikes_bm, sd, out)
            return out
        if isinstance(workspace, BinnedScatter):
            assert workspace.m == m and workspace.k == k, "workspace was built for another matrix shape"
            binned_batch(workspace, weights, indices, indptr, -1, spikes_bm, sd, out)
            return out
        f_ws = fn('be_binary_csrmm_t_workspace_bytes', c_i64, [c_i64, c_i64, c_i64, c_int])
        ws = A.workspace(f_ws(m, k, nb, A.wcode(weights)))
        f = None if sd >= A.BE_SPIKE_BITS else fn('be_binary_csrmm_t_' + _variant(homo, weights, sd), c_int, _CSRMM_ARGS)
    else:
        f_ws = fn('be_binary_csrmm_nt_workspace_bytes', c_i64, [c_i64, c_i64, c_i64])
        ws = A.workspace(f_ws(m, k, nb))
        f = None    # generic entry point: it takes the average row length as a hint for the kernel choice
    if f is None:     # (also: bit-packed events / id lists — the per-variant names cover bool / float only)
        hint = -1 if transpose else int(indices.numel() // max(m, 1))
        _csrmm_generic(weights, indices, indptr, hint, spikes_bm, sd, out, m, k, nb, ws, transpose)
        return out
    check(f(A.ptr(weights), A.ptr(indices), A.ptr(indptr), is64, A.ptr(spikes_bm), A.ptr(out), m, k, nb, A.ptr(ws),
            ws.numel(), A.stream_ptr()), f.__name__)
    return out


def _binary_csrmv_hip(weights, indices, indptr, vector, *, shape, transpose, workspace=None):
    spikes, sd = A.spikes_to_device(vector)
    return _csr_batched(weights, indices, indptr, spikes.reshape(1, -1), sd, shape=shape, transpose=transpose,
                        workspace=workspace)[0]


binary_csrmv_p = OpKernel('binary_csrmv')
binary_csrmv_p.def_kernel('hip', 'gpu', _binary_csrmv_hip, asdefault=True)
binary_csrmv_p.def_tags('csr', 'binary')


def _check_csr_structure_dtypes(indices, indptr):
    assert indices.dtype == torch.int32, f"indices must be int32; got {indices.dtype}."
    assert indptr.dtype in (torch.int32, torch.int64), f"indptr must be int32 or int64; got {indptr.dtype}."


def binary_csrmv_p_call(weights, indices, indptr, vector, workspace=None, *, shape, transpose, backend=None):
    """Validate, then dispatch (reference ``brainevent/_csr/binary.py:827-987``).  Returns the reference's 4-tuple
    ``(y, task_begin, task_end, status)``: the three task arrays are those of a reference-style ``workspace`` handle
    (:class:`_BinaryCsrmvTaskWorkspace`), else ``None`` — the kernels here do not produce them."""
    assert indptr.ndim == 1, "Indptr must be 1D."
    assert indices.ndim == 1, "Indices must be 1D."
    _check_csr_structure_dtypes(indices, indptr)
    if transpose:
        assert shape[0] == vector.shape[0], "Shape mismatch for transpose operation."
    else:
        assert shape[1] == vector.shape[0], "Shape mismatch for non-transpose operation."
    assert weights.dtype.is_floating_point, 'Weights must be a floating-point type.'
    if weights.ndim == 0:
        weights = weights.reshape(1)
    native = _resolve_workspace(workspace, weights, indices, indptr, shape, transpose)
    return (binary_csrmv_p(weights, indices, indptr, vector, shape=shape, transpose=transpose, workspace=native,
                           backend=backend),) + _task_operands(workspace)


binary_csrmv_p.def_call(binary_csrmv_p_call)


def _binary_csrmv_benchmark_data(*, platform):
    """A small slice of the reference's sweep (``brainevent/_csr/binary.py:760-800``: square matrices, size x density x
    direction x weight kind x event type), generated on the device."""
    dev = A.device()
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    for n in (1000, 10000):
        for conn_prob in (0.01, 0.1):
            n_conn = max(1, int(n * conn_prob))
            indptr = torch.arange(n + 1, dtype=torch.int32, device=dev) * n_conn
            indices = torch.randint(0, n, (n * n_conn,), dtype=torch.int32, device=dev, generator=g)
            for transpose in (False, True):
                for homo in (True, False):
                    weights = torch.ones(1 if homo else n * n_conn, device=dev)
                    for event_type in ('float', 'bool'):
                        data = torch.rand(n, device=dev, generator=g) < 0.01
                        vector = data.float() if event_type == 'float' else data
                        name = (f"{n}x{n},p={int(round(conn_prob * 100))}%,{'T' if transpose else 'NT'},"
                                f"{'homo' if homo else 'hetero'},{event_type}")
                        yield name, (weights, indices, indptr, vector, None), {'shape': (n, n), 'transpose': transpose}


binary_csrmv_p.def_benchmark_data(_binary_csrmv_benchmark_data)


def binary_csrmv(data, indices, indptr, v, *, shape, workspace=None, transpose: bool = False,
                 backend: Optional[str] = None):
    """Event-driven ``A @ v`` (``transpose=False``) or ``A.T @ v`` (``transpose=True``) for CSR ``A``.

    Same keyword signature as the reference (``brainevent/_csr/binary.py:128-138``); ``workspace`` is a
    :class:`ScatterPlan` prepared by the ``CSR`` / ``CSC`` classes, or ``None`` for the direct kernel.
    Output dtype = ``data`` dtype; output length ``shape[1]`` if ``transpose`` else ``shape[0]``.
    """
    as_np = A.wants_numpy(data, indices, indptr, v)
    w = A.to_device(data)
    idx = A.to_device(indices)
    ptr_ = A.to_device(indptr)
    if idx.dtype != torch.int32:
        idx = _as_int32_indices(idx, None, 'binary_csrmv', check_values=False)
    if ptr_.dtype not in (torch.int32, torch.int64):
        ptr_ = _as_indptr(ptr_, idx.shape[0], 'auto', 'binary_csrmv')
    vec = v if isinstance(v, torch.Tensor) else np.asarray(v)
    res = binary_csrmv_p_call(w, idx, ptr_, vec, workspace, shape=tuple(shape), transpose=transpose, backend=backend)[0]
    return A.to_result(res, as_np)


def _binary_csrmm_hip(weights, indices, indptr, B, *, shape, transpose, workspace=None):
    # one launch per stage for the whole batch (gridDim.y = columns of B); the kernels take the spike
    # matrix batch-major and emit a batch-major result, transposed back here exactly like the reference
    # does around its SRAW kernels (brainevent/_csr/binary.py:1263-1286).
    spikes_bm, sd = A.spikes_batch_major(B)
    out_bm = _csr_batched(weights, indices, indptr, spikes_bm, sd, shape=shape, transpose=transpose,
                          workspace=workspace)
    return out_bm.T


binary_csrmm_p = OpKernel('binary_csrmm')
binary_csrmm_p.def_kernel('hip', 'gpu', _binary_csrmm_hip, asdefault=True)
binary_csrmm_p.def_tags('csr', 'binary')


def binary_csrmm_p_call(weights, indices, indptr, B, workspace=None, *, shape, transpose, backend=None):
    """Validation of the matrix-operand op (reference ``brainevent/_csr/binary.py:1452-1607``)."""
    assert indptr.ndim == 1, "Indptr must be 1D."
    assert indices.ndim == 1, "Indices must be 1D."
    _check_csr_structure_dtypes(indices, indptr)
    assert B.ndim == 2, "B must be 2D."
    if transpose:
        assert shape[0] == B.shape[0], "Shape mismatch for transpose operation."
    else:
        assert shape[1] == B.shape[0], "Shape mismatch for non-transpose operation."
    assert weights.dtype.is_floating_point, 'Weights must be a floating-point type.'
    if weights.ndim == 0:
        weights = weights.reshape(1)
    native = _resolve_workspace(workspace, weights, indices, indptr, shape, transpose)
    return (binary_csrmm_p(weights, indices, indptr, B, shape=shape, transpose=transpose, workspace=native,
                           backend=backend),) + _task_operands(workspace)


binary_csrmm_p.def_call(binary_csrmm_p_call)


def binary_csrmm(data, indices, indptr, B, *, shape, workspace=None, transpose: bool = False,
                 backend: Optional[str] = None):
    """Event-driven ``A @ B`` / ``A.T @ B`` with a binary matrix ``B[rows, cols]``
    (reference ``brainevent/_csr/binary.py:264-384``): ``C[j, l] = sum_i A[i, j] * e(B[i, l])``."""
    as_np = A.wants_numpy(data, indices, indptr, B)
    w = A.to_device(data)
    idx = A.to_device(indices)
    ptr_ = A.to_device(indptr)
    if idx.dtype != torch.int32:
        idx = _as_int32_indices(idx, None, 'binary_csrmm', check_values=False)
    if ptr_.dtype not in (torch.int32, torch.int64):
        ptr_ = _as_indptr(ptr_, idx.shape[0], 'auto', 'binary_csrmm')
    Bm = B if isinstance(B, torch.Tensor) else np.asarray(B)
    res = binary_csrmm_p_call(w, idx, ptr_, Bm, workspace, shape=tuple(shape), transpose=transpose, backend=backend)[0]
    return A.to_result(res, as_np)


def _perm_of(perm, nse: int) -> torch.Tensor:
    p = A.to_device(perm).reshape(-1)
    if p.dtype not in (torch.int32, torch.int64):
        p = p.to(torch.int64 if nse > np.iinfo(np.int32).max else torch.int32)
    assert p.numel() == nse, f"perm must hold one entry per structural slot ({nse}); got {p.numel()}."
    return p


def _indexed_key(weights: torch.Tensor, perm: torch.Tensor):
    return weights_stamp(weights), (perm.data_ptr(), perm._version, perm.numel())


def indexed_workspace(data, indices, indptr, perm, *, shape, route: Optional[str] = None):
    """Scatter workspace of the re-indexed structure ``(data[perm], indices, indptr)`` — the per-matrix workspace the
    reference hands to ``binary_csrmv_indexed`` (``_csr/main.py:1647-1654``: ``_ensure_binary_workspace_and_get(self, "csc",
    csc_indptr)``).  The permuted weights are gathered ONCE, here (``be_gather_by_perm``); a planned workspace embeds them
    and the copy is dropped, a binned one keeps it.  The workspace remembers which ``(data, perm)`` it was derived from:
    :func:`binary_csrmv_indexed` re-derives it when ``data`` was modified in place, and never gathers per call."""
    from ._convert import gather_by_perm
    w, idx, ptr_ = A.to_device(data).reshape(-1), A.to_device(indices).reshape(-1), A.to_device(indptr)
    m, k = int(shape[0]), int(shape[1])
    nse = int(idx.numel())
    if w.numel() == 1:
        return make_scatter_workspace(route or choose_scatter_route(nse, m, k, w), w, idx, ptr_, m, k, nse)
    p = _perm_of(perm, nse)
    wp = gather_by_perm(w, p)
    ws = make_scatter_workspace(route or choose_scatter_route(nse, m, k, wp), wp, idx, ptr_, m, k, nse)
    if ws is not None:
        ws.indexed_key = _indexed_key(w, p)
        ws.indexed_data = wp if isinstance(ws, BinnedScatter) else None
    return ws


def _fresh_indexed_workspace(ws, w, idx, ptr_, p):
    """``ws`` (built by :func:`indexed_workspace`) brought up to date with the canonical weights ``w``; returns the weights
    argument of the step (the permuted copy of a binned workspace, unused by a planned one)."""
    from ._convert import gather_by_perm
    key = _indexed_key(w, p)
    if getattr(ws, 'indexed_key', None) != key:
        wp = gather_by_perm(w, p, out=getattr(ws, 'indexed_data', None))
        ws.refresh_weights(wp, idx, ptr_)
        ws.indexed_key = key
        if isinstance(ws, BinnedScatter):
            ws.indexed_data = wp
    return ws.indexed_data if isinstance(ws, BinnedScatter) else w


def _csr_batched_indexed(weights, indices, indptr, perm, spikes_bm, sd, *, shape, transpose, workspace=None):
    """``_csr_batched`` over ``weights[perm]`` without a per-call gather pass."""
    if weights.numel() == 1 or perm is None:             # one shared weight ignores perm, as in the reference
        return _csr_batched(weights, indices, indptr, spikes_bm, sd, shape=shape, transpose=transpose, workspace=workspace)
    m, k = int(shape[0]), int(shape[1])
    nb = int(spikes_bm.shape[0])
    out_len = k if transpose else m
    out = torch.empty((nb, out_len), dtype=weights.dtype, device=weights.device)
    if out_len == 0 or nb == 0:
        return out
    if m == 0 or k == 0 or indices.numel() == 0:
        return out.zero_()
    p = _perm_of(perm, int(indices.numel()))
    if transpose and isinstance(workspace, (ScatterPlan, BinnedScatter)):
        assert workspace.m == m and workspace.k == k, "workspace was built for another matrix shape"
        try:
            w_step = _fresh_indexed_workspace(workspace, weights.reshape(-1), indices, indptr, p)
        except MathError:
            workspace = None          # the new weights do not qualify for fixed point: the perm-fused direct kernel
        else:
            if isinstance(workspace, ScatterPlan):
                _plan_call(workspace, w_step, spikes_bm, sd, out)
            else:
                binned_batch(workspace, w_step, indices, indptr, -1, spikes_bm, sd, out)
            return out
    if transpose:
        ws = A.workspace(fn('be_binary_csrmm_t_workspace_bytes', c_i64, [c_i64, c_i64, c_i64, c_int])(m, k, nb, A.wcode(weights)))
        name = 'be_binary_csrmm_t_indexed'
    else:
        ws = A.workspace(fn('be_binary_csrmm_nt_workspace_bytes', c_i64, [c_i64, c_i64, c_i64])(m, k, nb))
        name = 'be_binary_csrmm_nt_indexed'
    f = fn(name, c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_int, c_i64, c_vp, c_int, c_vp, c_int, c_vp, c_i64, c_i64, c_i64, c_vp,
                         c_i64, c_vp])
    check(f(A.ptr(weights), 0, A.wcode(weights), A.ptr(indices), A.ptr(indptr), int(indptr.dtype == torch.int64), -1, A.ptr(p),
            int(p.dtype == torch.int64), A.ptr(spikes_bm), sd, A.ptr(out), m, k, nb, A.ptr(ws), ws.numel(), A.stream_ptr()), name)
    return out


def _binary_csrmv_indexed_hip(data, indices, indptr, perm, vector, *, shape, transpose, workspace=None):
    spikes, sd = A.spikes_to_device(vector)
    return _csr_batched_indexed(data, indices, indptr, perm, spikes.reshape(1, -1), sd, shape=shape, transpose=transpose,
                                workspace=workspace)[0]


def _binary_csrmm_indexed_hip(data, indices, indptr, perm, B, *, shape, transpose, workspace=None):
    spikes_bm, sd = A.spikes_batch_major(B)
    return _csr_batched_indexed(data, indices, indptr, perm, spikes_bm, sd, shape=shape, transpose=transpose,
                                workspace=workspace).T


def _indexed_operands(data, indices, indptr, context):
    w, idx, ptr_ = A.to_device(data), A.to_device(indices), A.to_device(indptr)
    if idx.dtype != torch.int32:
        idx = _as_int32_indices(idx, None, context, check_values=False)
    if ptr_.dtype not in (torch.int32, torch.int64):
        ptr_ = _as_indptr(ptr_, idx.shape[0], 'auto', context)
    assert indptr.ndim == 1 and indices.ndim == 1, "indices and indptr must be 1D."
    assert w.dtype.is_floating_point, 'Weights must be a floating-point type.'
    return (w.reshape(1) if w.ndim == 0 else w), idx, ptr_


def binary_csrmv_indexed(data, indices, indptr, perm, v, *, shape, workspace=None, transpose: bool = False,
                         backend: Optional[str] = None):
    """``binary_csrmv(data[perm], indices, indptr, v, ...)``: the product over a *re-indexed* structure (typically the CSC
    view of a CSR matrix, ``perm`` from :func:`csr_to_csc_index`) with the weights left in their canonical order
    (reference ``brainevent/_csr/binary_indexed.py:70-140``; slot ``j`` reads ``data[perm[j]]``,
    ``_csr/binary_indexed_csrmv_hybrid.cu:16-23``).  There is no per-call gather pass: ``workspace=None`` runs the perm-fused
    direct kernels (``be_binary_csrmm_{t,nt}_indexed`` — only the weights of active rows are read); a workspace from
    :func:`indexed_workspace` embeds (plan) or caches (binned) the permuted weights, keyed on ``(data, perm)``, and is
    re-derived only when ``data`` was modified in place.  One shared weight ignores ``perm``."""
    as_np = A.wants_numpy(data, indices, indptr, perm, v)
    w, idx, ptr_ = _indexed_operands(data, indices, indptr, 'binary_csrmv_indexed')
    vec = v if isinstance(v, torch.Tensor) else np.asarray(v)
    assert (shape[0] if transpose else shape[1]) == vec.shape[0], "Shape mismatch between the events and the structure."
    res = binary_csrmv_indexed_p(w, idx, ptr_, None if w.numel() == 1 else A.to_device(perm), vec, shape=tuple(shape),
                                 transpose=transpose, workspace=workspace, backend=backend)
    return A.to_result(res, as_np)


def binary_csrmm_indexed(data, indices, indptr, perm, B, *, shape, workspace=None, transpose: bool = False,
                         backend: Optional[str] = None):
    """Matrix-operand twin of :func:`binary_csrmv_indexed` (reference ``_csr/binary_indexed.py:615``)."""
    as_np = A.wants_numpy(data, indices, indptr, perm, B)
    w, idx, ptr_ = _indexed_operands(data, indices, indptr, 'binary_csrmm_indexed')
    Bm = B if isinstance(B, torch.Tensor) else np.asarray(B)
    assert Bm.ndim == 2, "B must be 2D."
    assert (shape[0] if transpose else shape[1]) == Bm.shape[0], "Shape mismatch between the events and the structure."
    res = binary_csrmm_indexed_p(w, idx, ptr_, None if w.numel() == 1 else A.to_device(perm), Bm, shape=tuple(shape),
                                 transpose=transpose, workspace=workspace, backend=backend)
    return A.to_result(res, as_np)


#: operator objects of the indexed products (reference ``_csr/binary_indexed.py``: ``binary_csrmv_indexed_p`` / ``binary_csrmm_indexed_p``)
binary_csrmv_indexed_p = OpKernel('binary_csrmv_indexed')
binary_csrmv_indexed_p.def_kernel('hip', 'gpu', _binary_csrmv_indexed_hip, asdefault=True)
binary_csrmv_indexed_p.def_tags('csr', 'binary', 'indexed')
binary_csrmm_indexed_p = OpKernel('binary_csrmm_indexed')
binary_csrmm_indexed_p.def_kernel('hip', 'gpu', _binary_csrmm_indexed_hip, asdefault=True)
binary_csrmm_indexed_p.def_tags('csr', 'binary', 'indexed')


# =====================================================================================================
# containers
# =====================================================================================================
#: matrices with fewer stored elements than this use the direct kernel (plan build is not worth it)
PLAN_MIN_NNZ = 1 << 15          # below this the direct kernel is used (nothing to gain from a layout)
#: below this average number of entries per (row, slice) segment the plan degenerates into pointer chasing
PLAN_MIN_SEGMENT = 8        # entries per (row, slice) from which the planned layout beats the binned route (measured: choose_scatter_route)
PLAN_MIN_SEGMENT_HOMO = 10  # ... for one shared weight (the binned route moves 2 B per counted entry)
PLAN_MIN_SEGMENT_NO_BINNED = 8   # ... and from which it beats the direct route when the binned route does not apply
# (the four constants above are the gfx950 defaults; brainevent_amd._tuning replaces them from the persisted per-architecture
#  store when one exists — applied at the end of this module)


#: the gather direction of a matrix with at least this many stored entries builds its mirror on first use (None: never
#: automatically; ``prepare(mirror=True)`` / ``build_mirror()`` always do).  Below it the gather kernel streams the matrix
#: in less time than a scatter step's launches take (2^24 entries x 8 B at ~6 TB/s = 22 us).
AUTO_MIRROR_MIN_NNZ: Optional[int] = 1 << 24
#: raw CSC arrays of a *planned* mirror are released above this many entries unless asked otherwise (the plan alone serves
#: every step; a weight update then rebuilds the mirror instead of re-encoding it)
MIRROR_KEEP_RAW_MAX_NNZ = 1 << 28
#: the permutation (4 / 8 bytes per entry) stays with a mirror whose raw arrays stay, up to this many entries: a weight
#: update is then one gather-copy (``be_gather_by_perm``) + the workspace's own refresh
MIRROR_KEEP_PERM_MAX_NNZ = 1 << 28


def _free_device_bytes() -> int:
    """Bytes a new allocation can get: what the driver reports free plus what torch's allocator holds unused."""
    free, _ = torch.cuda.mem_get_info()
    return int(free + torch.cuda.memory_reserved() - torch.cuda.memory_allocated())


def _mirror_bytes(nse: int, data: torch.Tensor, planned: bool) -> int:
    """Upper estimate of what a mirror of ``nse`` entries holds while it is built (raw CSC arrays + the plan's blocks)."""
    homo = data.numel() == 1
    raw = nse * (4 + (0 if homo else data.element_size()))
    return int(raw + (nse * (2.5 if homo else 7.5) if planned else 0))


def auto_mirror_wanted(nse: int, m: int, k: int, data: torch.Tensor) -> bool:
    """Whether the first gather-direction product of a matrix builds the mirror by itself: large enough for the event-driven
    route to pay, and the mirror fits in half of the free device memory (otherwise: the gather kernel, and a warning)."""
    if AUTO_MIRROR_MIN_NNZ is None or nse < AUTO_MIRROR_MIN_NNZ or m <= 0 or k <= 0:
        return False
    need = _mirror_bytes(nse, data, planned=True)
    if need > 0.5 * _free_device_bytes():
        import warnings
        warnings.warn(f"brainevent_amd: the event-driven mirror of this matrix ({need / 2**30:.1f} GiB) does not fit beside it; "
                      f"the gather direction streams the whole matrix on every call (build_mirror() forces the build).")
        return False
    return True


class Mirror:
    """The transposed structure of a CSR-like matrix ``A (m, k)`` with a scatter workspace of its own: ``A @ e`` is evaluated as
    the scatter of the *active columns* (``shape = (k, m)``: mirror row ``j`` lists the rows ``i`` with ``A[i, j]`` stored).
    ``data`` are the weights in mirror order (moved by the build, or the one shared weight); ``indices`` / ``indptr`` are
    ``None`` when the raw arrays were released (a planned mirror needs only its plan for the step); ``perm`` (mirror slot ->
    position in the source arrays) is kept for small mirrors so that a weight update is a gather-copy."""

    def __init__(self, shape, data, indices, indptr, plan, perm, stamp, homo, counts=None):
        self.shape = (int(shape[0]), int(shape[1]))
        self.data, self.indices, self.indptr = data, indices, indptr
        self.plan, self.perm, self.stamp, self.homo = plan, perm, stamp, bool(homo)
        self.counts = counts            # stored entries per mirror row (int64 [shape[0]]): the work of an event on that row
        self.check = None               # set by a container that built this mirror BY ITSELF: the streaming gather product, run once

    @property
    def released(self) -> bool:
        return self.indices is None

    def nbytes(self) -> int:
        n = 0
        for t in (self.data, self.indices, self.indptr, self.perm):
            n += 0 if t is None else t.numel() * t.element_size()
        if isinstance(self.plan, ScatterPlan):
            n += self.plan.nbytes()
        return n

    def is_stale(self, src_data: torch.Tensor) -> bool:
        return (not self.homo) and self.stamp != weights_stamp(src_data)

    def refreshed(self, src_data, src_indices, src_indptr, row_len, m, k) -> 'Mirror':
        """This mirror after the source weights changed in place: a gather-copy through ``perm`` into the same buffers + the
        workspace's own refresh when the permutation was kept, a rebuild otherwise."""
        if self.perm is None or self.released:
            return build_mirror_of(src_data, src_indices, src_indptr, row_len, m, k, keep_raw=not self.released,
                                   keep_perm=self.perm is not None)
        from ._convert import gather_by_perm
        gather_by_perm(src_data.reshape(-1), self.perm, out=self.data)      # in place: captured graphs keep the pointer
        self.stamp = weights_stamp(src_data)
        if self.plan is not None:       # (the library wrote through the raw pointer: torch's version counter did not move)
            try:
                self.plan.refresh_weights(self.data, self.indices, self.indptr)
            except MathError:
                self.plan = None
        return self

    def apply(self, v, backend=None):
        """``A @ e(v)`` for an event vector ``v [k]`` or a matrix operand ``v [k, n]`` -> ``[m]`` / ``[m, n]``.

        A mirror that a container built on its own initiative (first use of a large matrix, ``AUTO_MIRROR_MIN_NNZ``) is
        **cross-checked once** against the streaming gather kernel over the source arrays, on the first plain event vector it
        serves (one full-matrix pass, 12 ms at C2 against seconds of build; ``BRAINEVENT_AMD_MIRROR_CHECK=0`` skips it): the
        caller did not ask for the mirror, so it must not be able to change a result silently.  A disagreement returns the gather
        kernel's result, drops the mirror for good and warns with the numbers (round 4 saw one unexplained mismatch on this
        route)."""
        out = self._apply(v, backend)
        if self.check is not None and isinstance(v, torch.Tensor) and v.ndim == 1 and v.dtype in (torch.bool, torch.uint8, torch.float32):
            check, self.check = self.check, None
            if os.environ.get('BRAINEVENT_AMD_MIRROR_CHECK', '1') != '0':
                ref, drop = check(v)
                if self.homo:
                    bad = not torch.equal(out, ref)
                else:
                    tol = 1e-4 * ref.abs() + 1e-6 * float(ref.abs().max())
                    bad = bool(((out - ref).abs() > tol).any())
                if bad:
                    n_bad = int((out != ref).sum()) if self.homo else int(((out - ref).abs() > tol).sum())
                    drop()
                    import warnings
                    warnings.warn(f"brainevent_amd: the automatically built mirror of a {self.shape[1]} x {self.shape[0]} matrix disagreed "
                                  f"with the gather kernel on {n_bad} of {ref.numel()} outputs (max |diff| "
                                  f"{float((out - ref).abs().max()):.3e}); the mirror was dropped, this and later products use the gather "
                                  f"kernel.  Please report this.", RuntimeWarning, stacklevel=3)
                    return ref
        return out

    def _apply(self, v, backend=None):
        if not self.released:
            call = binary_csrmv_p_call if v.ndim == 1 else binary_csrmm_p_call
            return call(self.data, self.indices, self.indptr, v, self.plan, shape=self.shape, transpose=True, backend=backend)[0]
        if v.ndim == 1:
            sp, sd = A.spikes_to_device(v)
            spikes_bm = sp.reshape(1, -1)
        else:
            spikes_bm, sd = A.spikes_batch_major(v)
        out = torch.empty((int(spikes_bm.shape[0]), self.shape[1]), dtype=self.plan.weight_dtype, device=A.device())
        _plan_call(self.plan, self.data, spikes_bm, sd, out)
        return out[0] if v.ndim == 1 else out.T


def build_mirror_of(data, indices, indptr, row_len, m: int, k: int, *, keep_raw: Optional[bool] = None,
                    keep_perm: Optional[bool] = None) -> Mirror:
    """The :class:`Mirror` of a CSR-like matrix (``indptr=None`` + ``row_len``: fixed-length rows).

    Built by the library's column-block kernels (``_convert.CscBuilder``: one counting pass over the column ids, then the
    blocks' fills), so there is no entry-count limit: C2 / C4 (1e10 entries) convert on the device.  The route of the mirror's
    own scatter is chosen like any matrix's (``choose_scatter_route`` over ``k`` stored rows and ``m`` outputs):
      * planned: when the raw CSC arrays fit beside the plan they are built whole, planned, and released above
        ``MIRROR_KEEP_RAW_MAX_NNZ`` entries (``keep_raw``); when they do not fit, the plan is built from column blocks that
        are resident one at a time (``ScatterPlan.build_from_blocks``) and the raw arrays never exist;
      * binned / direct: the raw CSC arrays are the mirror.
    ``keep_perm``: keep the permutation (default: up to ``MIRROR_KEEP_PERM_MAX_NNZ`` entries, only with the raw arrays)."""
    from ._convert import CscBuilder
    data = A.to_device(data)
    flat = data.reshape(-1)
    homo = flat.numel() == 1
    nse = int(A.to_device(indices).numel())
    b = CscBuilder(indptr, indices, shape=(m, k), row_len=row_len)
    stamp = weights_stamp(data)
    route = choose_scatter_route(nse, k, m, flat) if nse >= PLAN_MIN_NNZ else 'direct'
    ptr_dtype = torch.int64 if nse > np.iinfo(np.int32).max else torch.int32
    w1 = flat[:1] if homo else None

    def whole(want_perm: bool):
        rows, w, perm = b.block(0, k, data=None if homo else flat, perm=want_perm)
        return (w1 if homo else w), rows, b.csc_indptr.to(ptr_dtype), perm

    oom = getattr(torch, 'OutOfMemoryError', getattr(torch.cuda, 'OutOfMemoryError', RuntimeError))
    if route == 'plan':
        need = _mirror_bytes(nse, flat, planned=True) + 2 * nse        # (+ the sorted layouts' row order, 2 B per entry)
        blocked = need > 0.85 * _free_device_bytes() and not (flat.dtype == torch.float64 and not homo)
        if keep_raw is None:
            keep_raw = nse <= MIRROR_KEEP_RAW_MAX_NNZ
        if not blocked:
            want_perm = bool(keep_raw and not homo and (keep_perm if keep_perm is not None else nse <= MIRROR_KEEP_PERM_MAX_NNZ))
            try:
                t_data, t_idx, t_ptr, perm = whole(want_perm)
                try:
                    plan = ScatterPlan.build(t_data, t_idx, t_ptr, shape=(k, m), keep_order=PLAN_KEEP_ORDER if keep_raw else False)
                except MathError:        # weights the fixed-point sums cannot resolve: the mirror runs the direct kernel
                    return Mirror((k, m), t_data, t_idx, t_ptr, None, perm, stamp, homo, b.counts)
                if keep_raw:
                    return Mirror((k, m), t_data, t_idx, t_ptr, plan, perm, stamp, homo, b.counts)
                plan.order = None
                return Mirror((k, m), w1 if homo else torch.empty(0, dtype=flat.dtype, device=flat.device), None, None, plan, None,
                              stamp, homo, b.counts)
            except oom:
                t_data = t_idx = t_ptr = perm = plan = None
                torch.cuda.empty_cache()
        # column blocks resident one at a time: each at most a quarter of what stays free beside the plan itself
        raw_bytes = _mirror_bytes(nse, flat, planned=False)
        plan_bytes = _mirror_bytes(nse, flat, planned=True) - raw_bytes
        budget = max(1 << 28, int(0.25 * (_free_device_bytes() - plan_bytes)))
        n_blocks = max(2, -(-raw_bytes // budget))
        block_cols = -(-k // n_blocks)

        def get_block(c0, c1):
            rows, w, _ = b.block(c0, c1, data=None if homo else flat)
            return (w1 if homo else w), rows, b.block_indptr(c0, c1)

        try:
            plan = ScatterPlan.build_from_blocks(get_block, block_cols, shape=(k, m), nnz=nse, max_row_len=b.max_col_count,
                                                 homo=homo, weight_dtype=flat.dtype)
            return Mirror((k, m), w1 if homo else torch.empty(0, dtype=flat.dtype, device=flat.device), None, None, plan, None,
                          stamp, homo, b.counts)
        except MathError:
            pass                      # fall through: the raw arrays + the direct kernel
    want_perm = bool(not homo and (keep_perm if keep_perm is not None else nse <= MIRROR_KEEP_PERM_MAX_NNZ))
    t_data, t_idx, t_ptr, perm = whole(want_perm)
    ws = None
    if route == 'binned' or (route == 'plan' and BinnedScatter.applicable(flat, m)):
        try:
            ws = BinnedScatter(t_data, k, m, nse, indices=t_idx, indptr=t_ptr)
        except MathError:
            ws = None
    return Mirror((k, m), t_data, t_idx, t_ptr, ws, perm, stamp, homo, b.counts)


class CompressedSparseData(DataRepresentation):
    """Common base of :class:`CSR` and :class:`CSC` (reference ``_csr/main.py:182-277``)."""
    _compressed_format = 'csr'

    def __init__(self, data, indices=None, indptr=None, *, shape, backend: Optional[str] = None,
                 buffers: Optional[Dict] = None, indptr_dtype="auto", check_structure: bool = True):
        if indices is None and indptr is None:
            args = data
        else:
            args = (data, indices, indptr)
        assert len(args) == 3, "Expected three arguments: data, indices, indptr."
        data_arr, indices_arg, indptr_arg = args
        fmt = type(self)._compressed_format
        shape = (int(shape[0]), int(shape[1]))
        secondary_dim = shape[1] if fmt == 'csr' else shape[0]
        context = f"{fmt.upper()} constructor"
        self._numpy_result = A.wants_numpy(data_arr, indices_arg, indptr_arg)
        self.data = A.to_device(data_arr)
        self.indices = _as_int32_indices(A.to_device(indices_arg), secondary_dim, context, check_values=check_structure)
        nse = self.indices.shape[0]
        self.indptr = _as_indptr(A.to_device(indptr_arg), nse, indptr_dtype, context)
        if check_structure:
            _check_compressed_structure(self.indices, self.indptr, shape, format=fmt, check_values=True)
        self.shape = shape
        self.backend = backend
        self._init_buffers(buffers)

    @classmethod
    def _from_parts(cls, data, indices, indptr, *, shape, backend=None, buffers=None, numpy_result=None):
        obj = cls((data, indices, indptr), shape=shape, backend=backend, buffers=buffers, check_structure=False)
        if numpy_result is not None:
            obj._numpy_result = numpy_result
        return obj

    @classmethod
    def fromdense(cls, mat, *, nse: Optional[int] = None, backend: Optional[str] = None):
        """Build from a dense matrix (host-side helper; non-zeros in row-major order for ``CSR``, column-major for
        ``CSC`` — reference ``CSR.fromdense`` / ``CSC.fromdense``, ``brainevent/_csr/main.py``)."""
        dense = mat.cpu().numpy() if isinstance(mat, torch.Tensor) else np.asarray(mat)
        if dense.ndim != 2:
            raise ValueError(f"{cls.__name__}.fromdense expects a 2-D matrix; got {dense.ndim}-D.")
        view = dense if cls._compressed_format == 'csr' else dense.T
        rows, cols = np.nonzero(view)
        if nse is not None:
            rows, cols = rows[:nse], cols[:nse]
        data = view[rows, cols]
        indptr = np.zeros(view.shape[0] + 1, dtype=np.int32)
        np.cumsum(np.bincount(rows, minlength=view.shape[0]), out=indptr[1:])
        obj = cls((data, cols.astype(np.int32), indptr), shape=dense.shape, backend=backend)
        obj._numpy_result = not isinstance(mat, torch.Tensor)
        return obj

    # -- properties ------------------------------------------------------------------------------
    @property
    def nse(self) -> int:
        return int(self.indices.shape[0])

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def ndim(self) -> int:
        return 2

    def with_data(self, data):
        """Same structure, new weights (cached plans are dropped: they embed the weights)."""
        data = A.to_device(data)
        assert data.shape == self.data.shape
        return type(self)._from_parts(data, self.indices, self.indptr, shape=self.shape, backend=self.backend,
                                       numpy_result=self._numpy_result)

    def todense(self):
        """Dense ``(shape)`` copy; duplicates are summed (host-side helper for tests / small matrices)."""
        idx = self.indices.cpu().numpy()
        ptr_ = self.indptr.cpu().numpy()
        w = self.data.float().cpu().numpy() if self.data.dtype == torch.bfloat16 else self.data.cpu().numpy()
        primary = np.repeat(np.arange(len(ptr_) - 1), np.diff(ptr_))
        vals = np.broadcast_to(w.reshape(-1), idx.shape) if w.size == 1 else w
        out = np.zeros(self.shape, dtype=vals.dtype)
        if type(self)._compressed_format == 'csr':
            np.add.at(out, (primary, idx), vals)
        else:
            np.add.at(out, (idx, primary), vals)
        return out

    # -- per-matrix workspace ----------------------------------------------------------------------
    def _plan_shape(self) -> Tuple[int, int]:
        """(rows, cols) of the stored compressed structure (CSC stores the transpose)."""
        return self.shape if type(self)._compressed_format == 'csr' else self.shape[::-1]

    def _scatter_workspace(self) -> Optional[ScatterPlan]:
        """Plan for the scatter direction, built on first use and cached in ``buffers``
        (reference ``_ensure_binary_workspace_and_get``, ``_csr/main.py:148-161``)."""
        if 'scatter_plan' in self.buffers:
            plan = self.buffers['scatter_plan'] = fresh_scatter_workspace(self.buffers['scatter_plan'], self.data, self.indices,
                                                                          self.indptr)
            return plan
        m, k = self._plan_shape()
        plan = None
        if self.nse >= PLAN_MIN_NNZ and m > 0 and k > 0:
            plan = make_scatter_workspace(choose_scatter_route(self.nse, m, k, self.data), self.data, self.indices, self.indptr,
                                          m, k, self.nse)
        self.buffers['scatter_plan'] = plan
        return plan

    def prepare(self, mirror: bool = False, keep_order: Optional[bool] = None, release_raw: bool = False):
        """Build the scatter workspace now (otherwise it is built by the first ``spk @ matrix``).
        ``mirror=True`` also builds the transposed mirror so that the *gather* direction runs event-driven too.
        ``keep_order=True``: a sorted-layout plan keeps its rows' column order (2 bytes per entry), which turns the refresh
        after an in-place weight update from a re-sort into a gather-copy (``ScatterPlan.build``).
        ``release_raw=True``: return a :class:`PlannedMatrix` that serves ``events @ M`` from the plan alone — the caller drops
        this container and with it the raw arrays (C2: 58 GB resident instead of 138 GB).  Needs a planned route (``MathError``
        / ``ValueError`` otherwise: the binned and the direct route read the raw arrays on every step)."""
        if release_raw:
            plan = self._scatter_workspace() if keep_order is None else self.prepare(keep_order=keep_order)._scatter_workspace()
            if not isinstance(plan, ScatterPlan):
                raise ValueError("prepare(release_raw=True): this matrix has no scatter plan (route: "
                                 f"{type(plan).__name__ if plan is not None else 'direct'}); its steps read the raw arrays.")
            plan.order = None            # (a weight refresh needs the raw arrays anyway)
            return PlannedMatrix(plan, self.data.reshape(-1)[:1] if plan.homo else None)
        if keep_order is not None and 'scatter_plan' not in self.buffers:
            global PLAN_KEEP_ORDER
            saved, PLAN_KEEP_ORDER = PLAN_KEEP_ORDER, keep_order
            try:
                self._scatter_workspace()
            finally:
                PLAN_KEEP_ORDER = saved
        self._scatter_workspace()
        if mirror:
            self.build_mirror()
        return self

    def refresh_weights(self):
        """Bring the cached workspaces up to date after ``self.data`` was modified in place.  The products check this by
        themselves on every call (``data._version``); call it explicitly between replays of a captured HIP graph, whose
        launches cannot."""
        if 'scatter_plan' in self.buffers:
            self._scatter_workspace()
        if 'mirror' in self.buffers:
            self._fresh_mirror()
        return self

    # -- transposed mirror: makes the unfavourable (gather) direction event-driven -----------------------------
    def build_mirror(self, *, keep_raw: Optional[bool] = None, keep_perm: Optional[bool] = None):
        """Materialise the transposed structure once (the reference's ``_weight_indices`` / ``csr_to_csc_index`` route,
        ``brainevent/_csr/main.py:1321-1357``, ``brainevent/_misc.py:1516``) and give it a scatter workspace: afterwards
        ``CSR @ spk`` / ``spk @ CSC`` scatter over the *active columns* instead of reading the whole matrix
        (:class:`Mirror`, :func:`build_mirror_of`: the library's column-block count / scan / fill kernels, any entry count)."""
        if 'mirror' in self.buffers:
            return self.buffers['mirror']
        m, k = self._plan_shape()                      # stored structure: m rows, k secondary ids
        self.buffers['mirror'] = build_mirror_of(self.data, self.indices, self.indptr, -1, m, k, keep_raw=keep_raw,
                                                 keep_perm=keep_perm)
        return self.buffers['mirror']

    def _fresh_mirror(self, auto: bool = False):
        """The cached mirror, brought up to date with ``self.data`` (it holds a permuted copy of the weights); with ``auto``
        a matrix large enough for the event-driven route to pay gets its mirror on first use, like the reference builds
        its CSC triple on the first ``CSR @ events`` (``_csr/main.py:1321-1357``)."""
        mr = self.buffers.get('mirror')
        if mr is None:
            if not auto or 'mirror' in self.buffers:      # (a cached None: the automatic build was refused once)
                return None
            m, k = self._plan_shape()
            if not auto_mirror_wanted(self.nse, m, k, self.data):
                self.buffers['mirror'] = None
                return None
            mr = self.build_mirror()

            owner = weakref.ref(self)            # (no cycle container -> mirror -> closure -> container: 80-GB arrays must not wait for the GC)

            def gather(v, _shape=(m, k)):
                c = owner()
                ref = binary_csrmv_p_call(c.data, c.indices, c.indptr, v, None, shape=_shape, transpose=False, backend=c.backend)[0]
                return ref, lambda: c.buffers.__setitem__('mirror', None)
            mr.check = gather
            return mr
        if mr.is_stale(self.data):
            m, k = self._plan_shape()
            mr = self.buffers['mirror'] = mr.refreshed(self.data, self.indices, self.indptr, -1, m, k)
        return mr

    def _gather_via_mirror(self, v):
        """Event-driven evaluation of the gather direction through the mirror, or ``None`` if there is no mirror.  ``v``:
        the event vector ``[k]`` or a matrix operand ``[k, n]`` (result ``[m, n]``)."""
        mr = self._fresh_mirror(auto=True)
        if mr is None:
            return None
        return mr.apply(v, backend=self.backend)

    def _res(self, t):
        return A.to_result(t, self._numpy_result)


def _event_value(other, scatter: bool = False):
    """Kernel operand of an event container: id list / packed words for 1-D compacted / bit-packed containers, else
    the value."""
    return event_operand(other, scatter=scatter)


def _dense_product(M, other, *, shape, transpose: bool, operand_on_left: bool):
    """A dense (non-event) operand against the stored arrays of a CSR / CSC container: ``op(A) @ x`` or ``x @ op(A)`` with
    ``op(A) = A.T if transpose else A`` and ``A`` the CSR reading of the arrays (``shape``).  Float-operand twins
    (``_float.csrmv`` / ``csrmm``; reference ``_csr/main.py:1595-1697``, ``:1699-1776``): ``x @ op(A) = (op(A).T @ x.T).T``."""
    from ._float import csrmv_p_call, csrmm_p_call
    x = other if isinstance(other, torch.Tensor) else np.asarray(other)
    t = (not transpose) if operand_on_left else transpose
    data, indices, indptr = M.data, M.indices, M.indptr
    if t:       # the scatter direction runs on float atomics (~21 G/s); the mirror turns it into a gather — an existing one, or
        #             the one a large matrix gets on first use as for event operands (auto_mirror_wanted) — but only while a
        #             mirror of that size keeps its raw arrays: a plan-only mirror cannot serve a dense operand
        mr = M._fresh_mirror(auto=M.nse <= MIRROR_KEEP_RAW_MAX_NNZ)
        if mr is not None and not mr.released and mr.indices is not None and tuple(mr.shape) == tuple(shape[::-1]):
            data, indices, indptr, shape, t = mr.data, mr.indices, mr.indptr, tuple(mr.shape), False
    if x.ndim == 1:
        r = csrmv_p_call(data, indices, indptr, x, shape=shape, transpose=t, backend=M.backend)[0]
    elif x.ndim == 2:
        r = csrmm_p_call(data, indices, indptr, x.T if operand_on_left else x, shape=shape, transpose=t, backend=M.backend)[0]
        r = r.T if operand_on_left else r
    else:
        raise NotImplementedError(f"matmul with object of shape {tuple(x.shape)}")
    return M._res(r) if A.wants_numpy(x) else r


class CSR(CompressedSparseData):
    """Compressed sparse row matrix with event-driven products (reference ``_csr/main.py:977``)."""
    _compressed_format = 'csr'

    def __matmul__(self, other):      # csr @ other
        if is_event(other):
            v = _event_value(other)
            r = None
            if v.ndim in (1, 2) and self._fresh_mirror(auto=True) is not None:
                # the mirror turns this product into a scatter: a compacted container hands its id list over (no compaction launch)
                v = _event_value(other, scatter=True)
                r = self._gather_via_mirror(v)
            if r is not None:
                pass
            elif v.ndim == 1:
                r = binary_csrmv_p_call(self.data, self.indices, self.indptr, v, None, shape=self.shape,
                                        transpose=False, backend=self.backend)[0]
            elif v.ndim == 2:
                r = binary_csrmm_p_call(self.data, self.indices, self.indptr, v, None, shape=self.shape,
                                        transpose=False, backend=self.backend)[0]
            else:
                raise NotImplementedError(f"matmul with object of shape {v.shape}")
            return self._res(r) if A.wants_numpy(v) else r
        return _dense_product(self, other, shape=self.shape, transpose=False, operand_on_left=False)      # csr @ x

    def __rmatmul__(self, other):     # other @ csr
        if is_event(other):
            v = _event_value(other, scatter=True)
            ws = self._scatter_workspace()
            if v.ndim == 1:
                r = binary_csrmv_p_call(self.data, self.indices, self.indptr, v, ws, shape=self.shape, transpose=True,
                                        backend=self.backend)[0]
            elif v.ndim == 2:
                r = binary_csrmm_p_call(self.data, self.indices, self.indptr, v.T, ws, shape=self.shape,
                                        transpose=True, backend=self.backend)[0].T
            else:
                raise NotImplementedError(f"matmul with object of shape {v.shape}")
            return self._res(r) if A.wants_numpy(v) else r
        return _dense_product(self, other, shape=self.shape, transpose=False, operand_on_left=True)       # x @ csr

    def transpose(self, axes=None):
        assert axes is None, "transpose does not support axes argument."
        return CSC._from_parts(self.data, self.indices, self.indptr, shape=self.shape[::-1], backend=self.backend,
                               numpy_result=self._numpy_result)

    def tocsr(self):
        return self

    def tocsc(self):
        """The *same* matrix (same ``shape``) re-encoded column-major (reference ``_csr/main.py:1198-1218``); unlike
        :meth:`transpose`, which reinterprets the arrays as ``W.T``."""
        from ._convert import csr_to_csc_index
        cptr, cidx, perm = csr_to_csc_index(self.indptr, self.indices, shape=self.shape)
        cdata = self.data if self.data.numel() == 1 else self.data[perm.long()]
        return CSC._from_parts(cdata, cidx, cptr.to(self.indptr.dtype), shape=self.shape, backend=self.backend,
                               numpy_result=self._numpy_result)

    @property
    def T(self):
        return self.transpose()


class CSC(CompressedSparseData):
    """Compressed sparse column matrix (reference ``_csr/main.py:1890``): ``indices`` are row ids,
    ``indptr`` has ``shape[1] + 1`` entries — i.e. the CSR arrays of the transpose."""
    _compressed_format = 'csc'

    def __matmul__(self, other):      # csc @ other : scatter over the active columns
        if is_event(other):
            v = _event_value(other, scatter=True)
            ws = self._scatter_workspace()
            if v.ndim == 1:
                r = binary_csrmv_p_call(self.data, self.indices, self.indptr, v, ws, shape=self.shape[::-1],
                                        transpose=True, backend=self.backend)[0]
            elif v.ndim == 2:
                r = binary_csrmm_p_call(self.data, self.indices, self.indptr, v, ws, shape=self.shape[::-1],
                                        transpose=True, backend=self.backend)[0]
            else:
                raise NotImplementedError(f"matmul with object of shape {v.shape}")
            return self._res(r) if A.wants_numpy(v) else r
        return _dense_product(self, other, shape=self.shape[::-1], transpose=True, operand_on_left=False)  # csc @ x = A'.T @ x

    def __rmatmul__(self, other):     # other @ csc : gather
        if is_event(other):
            v = _event_value(other)
            r = None
            if v.ndim == 1:
                if self._fresh_mirror(auto=True) is not None:
                    v = _event_value(other, scatter=True)      # (the mirror scatters: id lists are taken as they are)
                r = self._gather_via_mirror(v)
            elif v.ndim == 2:
                r = self._gather_via_mirror(v.T)
                r = None if r is None else r.T
            if r is not None:
                pass
            elif v.ndim == 1:
                r = binary_csrmv_p_call(self.data, self.indices, self.indptr, v, None, shape=self.shape[::-1],
                                        transpose=False, backend=self.backend)[0]
            elif v.ndim == 2:
                r = binary_csrmm_p_call(self.data, self.indices, self.indptr, v.T, None, shape=self.shape[::-1],
                                        transpose=False, backend=self.backend)[0].T
            else:
                raise NotImplementedError(f"matmul with object of shape {v.shape}")
            return self._res(r) if A.wants_numpy(v) else r
        return _dense_product(self, other, shape=self.shape[::-1], transpose=True, operand_on_left=True)   # x @ csc = x @ A'.T

    def transpose(self, axes=None):
        assert axes is None, "transpose does not support axes argument."
        return CSR._from_parts(self.data, self.indices, self.indptr, shape=self.shape[::-1], backend=self.backend,
                               numpy_result=self._numpy_result)

    def tocsc(self):
        return self

    def tocsr(self):
        """The *same* matrix (same ``shape``) re-encoded row-major (reference ``_csr/main.py:2107-2135``)."""
        from ._convert import csc_to_csr_index
        rptr, ridx, perm = csc_to_csr_index(self.indptr, self.indices, shape=self.shape)
        rdata = self.data if self.data.numel() == 1 else self.data[perm.long()]
        return CSR._from_parts(rdata, ridx, rptr.to(self.indptr.dtype), shape=self.shape, backend=self.backend,
                               numpy_result=self._numpy_result)

    @property
    def T(self):
        return self.transpose()


def _apply_persisted_tuning():
    """At import: the environment override or the defaults — never the per-device store, whose lookup would initialise the GPU
    (``_tuning.ensure_resolved`` does that at the first route choice, once the caller has picked a device)."""
    try:
        from . import _tuning
        _tuning.apply_scatter_tuning(_tuning.get_scatter_tuning(resolve_device=False), _resolved_key=('import', None))
    except Exception as e:          # noqa: BLE001 - a bad override must not make the package unimportable: say so, keep the defaults
        import warnings
        warnings.warn(f"brainevent_amd: persisted scatter tuning ignored ({e!r}); using the built-in defaults.")


_apply_persisted_tuning()
