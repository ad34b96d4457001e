"""The neuron half of the COBA step loop: one fused, in-place state update of a population of
conductance-based LIF neurons with exponential synapses (``be_lif_coba_step``), to sit between the two ``spikes @ CSR``
scatters of a time step.  The reference example composes the same dynamics from brainstate modules
(``examples/COBA_2005.py:35-87``); as separate elementwise launches they are ~20 launches per step — all a 4000-neuron
network's step consists of."""
import ctypes
import math

import torch

from . import _array as A
from ._lib import check, fn

__all__ = ['lif_coba_step', 'lif_cuba_step']


def _check_state(who, v, g_exc, g_inh, refractory, in_exc, in_inh, spikes, spike_bits, spike_count) -> int:
    n = int(v.numel())
    for t in (v, g_exc, g_inh, refractory, in_exc, in_inh):
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous() or t.numel() != n:
            raise ValueError(f'{who}: state and input tensors must be contiguous f32 device tensors of one length.')
    if spikes is None and spike_bits is None:
        raise ValueError(f'{who}: give spikes, spike_bits or both.')
    if spikes is not None and (spikes.numel() != n or spikes.dtype not in (torch.bool, torch.uint8) or not spikes.is_cuda
                               or not spikes.is_contiguous()):
        raise ValueError(f'{who}: spikes must be a contiguous bool / uint8 device tensor of the same length.')
    if spike_bits is not None and (spike_bits.dtype != torch.int32 or spike_bits.numel() < (n + 31) // 32 or not spike_bits.is_cuda
                                   or not spike_bits.is_contiguous()):
        raise ValueError(f'{who}: spike_bits must be a contiguous int32 device tensor of ceil(n / 32) words.')
    if spike_count is not None and (spike_count.dtype != torch.float32 or spike_count.numel() != n or not spike_count.is_cuda):
        raise ValueError(f'{who}: spike_count must be an f32 device tensor of the same length.')
    return n


def lif_coba_step(v: torch.Tensor, g_exc: torch.Tensor, g_inh: torch.Tensor, refractory: torch.Tensor,
                  in_exc: torch.Tensor, in_inh: torch.Tensor, spikes: torch.Tensor = None, spike_count: torch.Tensor = None, *,
                  spike_bits: torch.Tensor = None,
                  dt: float = 0.1, tau_m: float = 20.0, v_rest: float = -60.0, v_th: float = -50.0, v_reset: float = -60.0,
                  t_ref: float = 5.0, e_exc: float = 0.0, e_inh: float = -80.0, tau_exc: float = 5.0, tau_inh: float = 10.0,
                  i_ext: float = 20.0, syn_scale: float = 1e-3, in_scale_exc: float = 1.0, in_scale_inh: float = 1.0) -> None:
    """Advance ``v``, ``g_exc``, ``g_inh``, ``refractory`` (f32 device tensors of one length) by one step **in place**, write
    this step's spikes (``bool`` / ``uint8``) to ``spikes`` and add them to ``spike_count`` if given.  ``in_exc`` / ``in_inh``
    are this step's synaptic inputs (the outputs of ``BinaryArray(spikes) @ W_exc`` / ``@ W_inh``).  Defaults: the COBA
    benchmark network (Vogels & Abbott 2005).  Every operation is rounded separately, in the order the header states, so the
    result equals the same formulas written as elementwise tensor ops bit for bit.

    ``spike_bits`` (int32 ``[ceil(n / 32)]``, optional): the spikes are also — or, with ``spikes=None``, only — written
    bit-packed, the form ``BitPackedBinary.from_packed(spike_bits, n) @ conn`` and the multi-GPU spike exchange consume as they
    are: a step loop that keeps its spikes as words has no pack launch (``be_lif_coba_step_packed``).

    ``in_scale_exc`` / ``in_scale_inh``: the inputs are multiplied by these first (``be_lif_step_scaled_packed``).  With both
    projections stacked into ONE ``n x 2n`` matrix of weight 1, ``counts = BinaryArray(spikes) @ W`` is one scatter and
    ``lif_coba_step(..., counts[:n], counts[n:], ..., in_scale_exc=w_exc, in_scale_inh=w_inh)`` equals the two-projection step bit
    for bit (``count * w`` is rounded exactly as inside a scatter with weight ``w``) with half the launches."""
    n = _check_state('lif_coba_step', v, g_exc, g_inh, refractory, in_exc, in_inh, spikes, spike_bits, spike_count)
    _scaled_step(0, v, g_exc, g_inh, refractory, in_exc, in_inh, in_scale_exc, in_scale_inh, spikes, spike_bits, spike_count, n,
                 dt, tau_m, v_rest, v_th, v_reset, t_ref, e_exc, e_inh, tau_exc, tau_inh, i_ext, syn_scale)


def _scaled_step(current_based, v, g_exc, g_inh, refractory, in_exc, in_inh, s_exc, s_inh, spikes, spike_bits, spike_count, n, dt,
                 tau_m, v_rest, v_th, v_reset, t_ref, e_exc, e_inh, tau_exc, tau_inh, i_ext, syn_scale) -> None:
    c_d, c_vp = ctypes.c_double, ctypes.c_void_p
    f = fn('be_lif_step_scaled_packed', ctypes.c_int, [ctypes.c_int] + [c_vp] * 6 + [c_d, c_d] + [c_vp] * 3 + [ctypes.c_int64]
           + [c_d] * 12 + [c_vp])
    check(f(int(current_based), A.ptr(v), A.ptr(g_exc), A.ptr(g_inh), A.ptr(refractory), A.ptr(in_exc), A.ptr(in_inh),
            float(s_exc), float(s_inh), A.ptr(spikes), A.ptr(spike_bits), A.ptr(spike_count), n, dt, tau_m, v_rest, v_th, v_reset,
            t_ref, e_exc, e_inh, math.exp(-dt / tau_exc), math.exp(-dt / tau_inh), i_ext, syn_scale, A.stream_ptr()),
          'be_lif_step_scaled_packed')


def lif_cuba_step(v: torch.Tensor, g_exc: torch.Tensor, g_inh: torch.Tensor, refractory: torch.Tensor,
                  in_exc: torch.Tensor, in_inh: torch.Tensor, spikes: torch.Tensor = None, spike_count: torch.Tensor = None, *,
                  spike_bits: torch.Tensor = None,
                  dt: float = 0.1, tau_m: float = 20.0, v_rest: float = -49.0, v_th: float = -50.0, v_reset: float = -60.0,
                  t_ref: float = 5.0, tau_exc: float = 5.0, tau_inh: float = 10.0, i_ext: float = 20.0, syn_scale: float = 1.0,
                  in_scale_exc: float = 1.0, in_scale_inh: float = 1.0) -> None:
    """The current-based twin of :func:`lif_coba_step` (``be_lif_cuba_step_packed``): the synaptic current is
    ``(g_exc + g_inh) * syn_scale`` — no reversal potentials; an inhibitory projection carries a negative weight.  Defaults: the
    reference's CUBA benchmark network (``examples/CUBA_2005.py:35-66``: V_rest -49 mV, weights 1.62 / -9.0 mS times one volt).
    Same argument contract, rounding order and spike outputs as :func:`lif_coba_step`."""
    n = _check_state('lif_cuba_step', v, g_exc, g_inh, refractory, in_exc, in_inh, spikes, spike_bits, spike_count)
    _scaled_step(1, v, g_exc, g_inh, refractory, in_exc, in_inh, in_scale_exc, in_scale_inh, spikes, spike_bits, spike_count, n,
                 dt, tau_m, v_rest, v_th, v_reset, t_ref, 0.0, 0.0, tau_exc, tau_inh, i_ext, syn_scale)
