#!/usr/bin/env python3
"""A million LIF neurons, 1e10 synapses, one GPU: the headline operator inside a simulation loop.

N leaky integrate-and-fire neurons (tau 20 ms, threshold 1, reset 0, dt 1 ms) receive a noisy external drive that
keeps them near 10 Hz (1 % of the population per step) and the recurrent current ``BinaryArray(spikes) @ CSR`` of a
random f32 matrix with 1 % density (C2 of BASELINE.json: 1M x 1M, 10 000 synapses per row, 80 GB of CSR + the plan).
Excitatory (80 %) and inhibitory (20 %, 4x stronger) rows balance, so the recurrent input shapes the rate without
taking it over.  This is a throughput demonstration, not a model from the literature.

    python examples/lif_network_1m.py [n] [steps]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
from bench import gen_csr_on_device


def main(n=1_000_000, steps=500):
    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev); g.manual_seed(1)
    n_conn = max(1, n // 100)
    t0 = time.perf_counter()
    w, idx, ptr = gen_csr_on_device(n, n, n_conn, False, 11, dev)            # U[0,1) weights
    n_exc = int(0.8 * n)
    scale = 0.2 / (n_conn * 0.01)                                            # ~0.2 of threshold arrives per step at 1 % firing
    w[: n_exc * n_conn] *= scale
    w[n_exc * n_conn:] *= -4.0 * scale
    csr = be.CSR((w, idx, ptr), shape=(n, n), check_structure=False).prepare()
    torch.cuda.synchronize()
    print(f'{n} neurons, {n * n_conn:.3g} synapses, route {type(csr.buffers["scatter_plan"]).__name__}, '
          f'setup {time.perf_counter() - t0:.1f} s', flush=True)

    V = torch.rand(n, device=dev, generator=g)
    spk = torch.zeros(n, dtype=torch.bool, device=dev)
    count = torch.zeros((), dtype=torch.int64, device=dev)
    decay = float(torch.exp(torch.tensor(-1.0 / 20.0)))

    def step():
        nonlocal V, spk, count
        I_rec = be.BinaryArray(spk) @ csr
        drive = torch.empty(n, device=dev).normal_(0.046, 0.05, generator=g)
        V = V * decay + drive + I_rec
        spk = V >= 1.0
        V = torch.where(spk, torch.zeros_like(V), V)
        count += spk.sum()

    for _ in range(50):
        step()
    torch.cuda.synchronize()
    count.zero_()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rate = float(count.item()) / n / steps * 1000.0
    ev = float(count.item()) * n_conn
    print(f'{steps} steps of 1 ms in {dt:.3f} s: {dt / steps * 1e6:.0f} us/step, mean rate {rate:.1f} Hz, '
          f'{ev / dt / 1e9:.0f} G synaptic events/s delivered', flush=True)


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, int(sys.argv[2]) if len(sys.argv) > 2 else 500)
