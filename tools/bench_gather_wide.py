#!/usr/bin/env python3
"""Gather (CSR @ spikes) over more input columns than the LDS holds bits for (> 1.2M): the coarse-bitmap spike test."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
from bench import gen_csr_on_device
dev = torch.device('cuda', 0)
for (m, k, nc, fire) in ((2_000_000, 2_000_000, 100, 0.01), (2_000_000, 4_000_000, 100, 0.01), (1_000_000, 10_000_000, 1000, 0.01),
                          (2_000_000, 4_000_000, 100, 0.2)):
    w, idx, ptr = gen_csr_on_device(m, k, nc, False, 3, dev)
    spk = torch.rand(k, device=dev) < fire
    for _ in range(3):
        out = be.binary_csrmv(w, idx, ptr, spk, shape=(m, k), transpose=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out = be.binary_csrmv(w, idx, ptr, spk, shape=(m, k), transpose=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f'gather m={m} k={k} nnz/row={nc} fire={fire}: {dt*1e3:.3f} ms, {m*nc*8/dt/1e9:.0f} GB/s of matrix stream', flush=True)
    del w, idx, ptr, out
    torch.cuda.empty_cache()
