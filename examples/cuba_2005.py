#!/usr/bin/env python3
"""CUBA E/I balanced network (Vogels & Abbott 2005 / Brette et al. 2007) on brainevent_amd — the current-based twin of
`coba_2005.py` (reference: examples/CUBA_2005.py:35-82, which builds the same network with brainstate).

3200*scale excitatory + 800*scale inhibitory LIF neurons (V_rest -49 mV, V_th -50 mV, V_reset -60 mV, tau 20 ms,
refractory 5 ms, V0 ~ N(-55, 2) mV, constant input 20), every neuron projects to 80 random targets (weights 1.62 /
-9.0 mS), exponential synapses (tau 5 / 10 ms) whose state is injected as a current (CUBA output, scale 1 V), dt = 0.1 ms.
The two projections are `BinaryArray(spikes) @ CSR`; the neuron and synapse update is either plain torch ops (`run`) or
the library's fused step `be.lif_cuba_step` (`run_fused`: three launches per time step, replayed as a HIP graph; the same
spikes bit for bit).  The reference's example reports 24-25 Hz (A6000: 24.98 Hz at scale 1, 24.3 Hz at scale 100).

    python examples/cuba_2005.py [scale] [steps]
"""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be

V_REST, V_TH, V_RESET, TAU, T_REF, I_EXT = -49.0, -50.0, -60.0, 20.0, 5.0, 20.0
W_EXC, W_INH, TAU_E, TAU_I = 1.62, -9.0, 5.0, 10.0


def build(scale, dev, seed=0, drawn=None):
    drawn = [] if drawn is None else drawn
    g = torch.Generator(device=dev); g.manual_seed(seed)
    n_exc, n_inh = int(3200 * scale), int(800 * scale)
    n = n_exc + n_inh

    def proj(n_pre, w):
        indptr = torch.arange(n_pre + 1, dtype=torch.int32, device=dev) * 80
        indices = torch.randint(0, n, (n_pre * 80,), dtype=torch.int32, device=dev, generator=g)
        drawn.append(indices)
        return be.CSR((torch.full((1,), w, device=dev), indices, indptr), shape=(n_pre, n), check_structure=False).prepare()

    return n_exc, n_inh, n, proj(n_exc, W_EXC), proj(n_inh, W_INH), g


def elementwise_step(V, ge, gi, refr, spk, E, I, n_exc, dt):
    """One time step as separate tensor ops (the formulation `be.lif_cuba_step` reproduces bit for bit); returns the new state."""
    ge = ge * math.exp(-dt / TAU_E) + (be.BinaryArray(spk[:n_exc]) @ E)
    gi = gi * math.exp(-dt / TAU_I) + (be.BinaryArray(spk[n_exc:]) @ I)
    I_syn = (ge + gi) * 1.0                      # g [mS] x 1 V = mA, beside the 20 mA drive (R = 1 ohm)
    dV = (-(V - V_REST) + I_syn + I_EXT) * (dt / TAU)
    active = refr <= 0
    V = torch.where(active, V + dV, V)
    spk = active & (V >= V_TH)
    V = torch.where(spk, torch.full_like(V, V_RESET), V)
    refr = torch.where(spk, torch.full_like(refr, T_REF), refr - dt)
    return V, ge, gi, refr, spk


def run(scale=1.0, steps=10000, dt=0.1):
    dev = torch.device('cuda', 0)
    n_exc, n_inh, n, E, I, g = build(scale, dev)
    V = torch.empty(n, device=dev).normal_(-55.0, 2.0, generator=g)
    ge, gi, refr, count = (torch.zeros(n, device=dev) for _ in range(4))
    spk = torch.zeros(n, dtype=torch.bool, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        V, ge, gi, refr, spk = elementwise_step(V, ge, gi, refr, spk, E, I, n_exc, dt)
        count += spk
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return n, el, float(count.sum().item()) / n / (steps * dt * 1e-3)


def build_combined(scale, dev, seed=0):
    """The same two projections (the very same draw) stacked into ONE matrix of weight 1: row r of the n x 2n matrix is neuron r's
    80 targets, in columns [0, n) for an excitatory neuron and [n, 2n) for an inhibitory one.  `BinaryArray(spikes) @ W` then
    counts, per target, the excitatory spikes in the first half of its output and the inhibitory ones in the second — one scatter
    for both projections; the weights are applied by the neuron step (`in_scale_exc` / `in_scale_inh`)."""
    drawn = []
    n_exc, n_inh, n, _, _, g = build(scale, dev, seed, drawn)
    indices = torch.cat([drawn[0], drawn[1] + n])
    indptr = torch.arange(n + 1, dtype=torch.int32, device=dev) * 80
    W = be.CSR((torch.ones(1, device=dev), indices, indptr), shape=(n, 2 * n), check_structure=False).prepare()
    return n_exc, n_inh, n, W, g


def run_fused(scale=1.0, steps=10000, dt=0.1, graph=True, unroll=1, combined=False):
    """The same simulation with the neuron / synapse update as ONE launch (`be.lif_cuba_step`): a time step is two scatters and
    one neuron kernel, captured once and replayed as a HIP graph."""
    dev = torch.device('cuda', 0)
    if combined:          # one scatter per step for both projections (build_combined): the same spikes bit for bit, half the launches
        n_exc, n_inh, n, W, g = build_combined(scale, dev)
    else:
        n_exc, n_inh, n, E, I, g = build(scale, dev)
    V = torch.empty(n, device=dev).normal_(-55.0, 2.0, generator=g)
    ge, gi, refr, count = (torch.zeros(n, device=dev) for _ in range(4))
    spk = torch.zeros(n, dtype=torch.bool, device=dev)

    def step_combined():
        cnt = be.BinaryArray(spk) @ W
        be.lif_cuba_step(V, ge, gi, refr, cnt[:n], cnt[n:], spk, count, dt=dt, v_rest=V_REST, v_th=V_TH, v_reset=V_RESET, tau_m=TAU,
                         t_ref=T_REF, tau_exc=TAU_E, tau_inh=TAU_I, i_ext=I_EXT, syn_scale=1.0, in_scale_exc=W_EXC, in_scale_inh=W_INH)

    def step_two():
        in_e = be.BinaryArray(spk[:n_exc]) @ E
        in_i = be.BinaryArray(spk[n_exc:]) @ I
        be.lif_cuba_step(V, ge, gi, refr, in_e, in_i, spk, count, dt=dt, v_rest=V_REST, v_th=V_TH, v_reset=V_RESET, tau_m=TAU,
                         t_ref=T_REF, tau_exc=TAU_E, tau_inh=TAU_I, i_ext=I_EXT, syn_scale=1.0)

    step = step_combined if combined else step_two

    unroll = unroll if graph else 1              # `unroll` time steps per replayed graph (one host call each)
    fn_ = be.capture_step(step, repeat=unroll) if graph else step
    # (the capture's warm-up and recording advanced the state by a few steps: the count restarts here, the dynamics simply go on)
    count.zero_()
    steps = (steps // unroll) * unroll
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps // unroll):
        fn_()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return n, el, float(count.sum().item()) / n / (steps * dt * 1e-3), V, spk


if __name__ == '__main__':
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    run(scale, 200)                                   # warm up
    n, el, rate = run(scale, steps)
    print(f'scale={scale:g}, size={n}, steps={steps} (dt 0.1 ms), time = {el:.3f} s ({el / steps * 1e6:.1f} us/step), '
          f'firing rate = {rate:.2f} Hz', flush=True)
    for graph in (False, True):
        n, el, rate, _, _ = run_fused(scale, steps, graph=graph)
        print(f'  fused neuron step{", HIP-graph replay" if graph else ""}: time = {el:.3f} s ({el / steps * 1e6:.1f} us/step), '
              f'firing rate = {rate:.2f} Hz', flush=True)
