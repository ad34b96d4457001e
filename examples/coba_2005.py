#!/usr/bin/env python3
"""COBA E/I balanced network (Vogels & Abbott 2005 / Brette et al. 2007) on brainevent_amd — the plumbing config
(BASELINE.json configs[0]; reference: examples/COBA_2005.py:35-87, which builds the same network with brainstate).

3200*scale excitatory + 800*scale inhibitory LIF neurons (V_rest -60 mV, V_th -50 mV, V_reset -60 mV, tau 20 ms,
refractory 5 ms, V0 ~ N(-55, 2) mV, constant input 20), every neuron projects to 80 random targets
(weights 0.6 / 6.7 mS), exponential conductance synapses (tau 5 / 10 ms, reversal 0 / -80 mV), dt = 0.1 ms.
The two projections are `BinaryArray(spikes) @ CSR`; neuron and synapse state updates are plain torch ops (`run`,
`run_graph`) or the library's fused neuron step `be.lif_coba_step` (`run_fused`: three launches per time step, replayed
as a HIP graph; the same spikes bit for bit).

    python examples/coba_2005.py [scale] [steps]
"""
import sys
import time

import torch

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import brainevent_amd as be


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

    return n_exc, n_inh, n, proj(n_exc, 0.6), proj(n_inh, 6.7), g


def run_graph(scale=1.0, steps=10000, dt=0.1):
    """Same simulation with the whole time step captured once in a HIP graph (torch.cuda.CUDAGraph) and replayed:
    the C-ABI calls are asynchronous on the current stream and allocate nothing themselves, so they capture like
    any other kernel launch."""
    dev = torch.device('cuda', 0)
    n_exc, n_inh, n, E, I, g = build(scale, dev)
    st = {'V': torch.empty(n, device=dev).normal_(-55.0, 2.0, generator=g), 'ge': torch.zeros(n, device=dev),
          'gi': torch.zeros(n, device=dev), 'refr': torch.zeros(n, device=dev),
          'spk': torch.zeros(n, dtype=torch.bool, device=dev), 'count': torch.zeros(n, device=dev)}
    dec_e, dec_i = float(torch.exp(torch.tensor(-dt / 5.0))), float(torch.exp(torch.tensor(-dt / 10.0)))

    def step():
        V, spk = st['V'], st['spk']
        ge = st['ge'] * dec_e + (be.BinaryArray(spk[:n_exc]) @ E)
        gi = st['gi'] * dec_i + (be.BinaryArray(spk[n_exc:]) @ I)
        I_syn = (ge * (0.0 - V) + gi * (-80.0 - V)) * 1e-3
        dV = (-(V - (-60.0)) + I_syn + 20.0) * (dt / 20.0)
        active = st['refr'] <= 0
        Vn = torch.where(active, V + dV, V)
        s = active & (Vn >= -50.0)
        st['V'].copy_(torch.where(s, torch.full_like(Vn, -60.0), Vn))
        st['refr'].copy_(torch.where(s, torch.full_like(V, 5.0), st['refr'] - dt))
        st['ge'].copy_(ge); st['gi'].copy_(gi); st['spk'].copy_(s)
        st['count'] += s

    graphed = be.capture_step(step)      # warm-up on a side stream (plans, workspaces), then one capture
    st['count'].zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        graphed()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return n, el, float(st['count'].sum().item()) / n / (steps * dt * 1e-3)


def run(scale=1.0, steps=10000, dt=0.1):
    dev = torch.device('cuda', 0)
    n_exc, n_inh, n, E, I, g = build(scale, dev)
    V = torch.empty(n, device=dev).normal_(-55.0, 2.0, generator=g)
    ge = torch.zeros(n, device=dev)
    gi = torch.zeros(n, device=dev)
    refr = torch.zeros(n, device=dev)              # remaining refractory time (ms)
    spk = torch.zeros(n, dtype=torch.bool, device=dev)
    count = torch.zeros(n, device=dev)
    dec_e, dec_i = float(torch.exp(torch.tensor(-dt / 5.0))), float(torch.exp(torch.tensor(-dt / 10.0)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        # synaptic input of the spikes of the previous step (event-driven scatter through the C ABI)
        ge = ge * dec_e + (be.BinaryArray(spk[:n_exc]) @ E)
        gi = gi * dec_i + (be.BinaryArray(spk[n_exc:]) @ I)
        # units as in the reference: g [mS] * (E - V) [mV] = uA, added to the 20 mA drive -> factor 1e-3 (R = 1 ohm)
        I_syn = (ge * (0.0 - V) + gi * (-80.0 - V)) * 1e-3
        dV = (-(V - (-60.0)) + I_syn + 20.0) * (dt / 20.0)
        active = refr <= 0
        V = torch.where(active, V + dV, V)
        spk = active & (V >= -50.0)
        V = torch.where(spk, torch.full_like(V, -60.0), V)
        refr = torch.where(spk, torch.full_like(refr, 5.0), refr - dt)
        count += spk
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    rate = float(count.sum().item()) / n / (steps * dt * 1e-3)
    return n, el, rate


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
    """Same simulation with the neuron / synapse update as ONE launch (``be.lif_coba_step``: the formulas of :func:`run`, every
    operation rounded separately in the same order — identical spikes): a time step is two scatters and one neuron kernel."""
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
        be.lif_coba_step(V, ge, gi, refr, cnt[:n], cnt[n:], spk, count, dt=dt, in_scale_exc=0.6, in_scale_inh=6.7)

    def step_two():
        in_e = be.BinaryArray(spk[:n_exc]) @ E
        in_i = be.BinaryArray(spk[n_exc:]) @ I
        be.lif_coba_step(V, ge, gi, refr, in_e, in_i, spk, count, dt=dt)

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
    try:
        n, el, rate = run_graph(scale, steps)
        print(f'  HIP-graph replay: time = {el:.3f} s ({el / steps * 1e6:.1f} us/step), firing rate = {rate:.2f} Hz', flush=True)
    except Exception as e:      # capture support depends on the torch build; the eager loop above is the reference
        print('  HIP-graph replay unavailable:', repr(e)[:200], flush=True)
    for graph in (False, True):
        try:
            n, el, rate, _, _ = run_fused(scale, steps, graph=graph)
            print(f'  fused neuron step{", HIP-graph replay" if graph else ""}: time = {el:.3f} s ({el / steps * 1e6:.1f} us/step), '
                  f'firing rate = {rate:.2f} Hz', flush=True)
        except Exception as e:
            print('  fused neuron step unavailable:', repr(e)[:200], flush=True)
