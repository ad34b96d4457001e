#!/usr/bin/env python3
"""Gather direction (CSR @ spikes, transpose=False): streams the whole matrix; reports achieved HBM bandwidth."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
from bench import gen_csr_on_device
dev = torch.device('cuda', 0)
for (m, k, nc, homo) in ((100_000, 1_000_000, 10_000, False), (100_000, 1_000_000, 10_000, True), (2_000_000, 2_000_000, 100, False)):
    w, idx, ptr = gen_csr_on_device(m, k, nc, homo, 3, dev)
    spk = torch.rand(k, device=dev) < 0.01
    for _ in range(3):
        out = be.binary_csrmv(w, idx, ptr, spk, shape=(m, k), transpose=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        out = be.binary_csrmv(w, idx, ptr, spk, shape=(m, k), transpose=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    nnz = m * nc
    byts = nnz * (4 if homo else 8)
    print(f'gather m={m} k={k} nnz/row={nc} {"homo" if homo else "hetero"}: {dt*1e3:.3f} ms, {byts/dt/1e9:.0f} GB/s of matrix stream, {nnz/dt/1e9:.1f} G synapses/s', flush=True)
    del w, idx, ptr
    torch.cuda.empty_cache()

# batched gather (binary_csrmm transpose=False): fused over the batch vs one pass per column
for (m, k, nc, homo, nb, fire) in ((100_000, 1_000_000, 10_000, False, 32, 0.01), (100_000, 1_000_000, 10_000, True, 32, 0.01),
                                   (100_000, 1_000_000, 10_000, False, 32, 0.2), (100_000, 1_000_000, 10_000, False, 8, 0.01)):
    w, idx, ptr = gen_csr_on_device(m, k, nc, homo, 3, dev)
    B = torch.rand((k, nb), device=dev) < fire
    for _ in range(2):
        out = be.binary_csrmm(w, idx, ptr, B, shape=(m, k), transpose=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        out = be.binary_csrmm(w, idx, ptr, B, shape=(m, k), transpose=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    col = be.binary_csrmv(w, idx, ptr, B[:, 0].contiguous(), shape=(m, k), transpose=False)
    torch.cuda.synchronize()
    dt1 = time.perf_counter() - t0
    ok = torch.allclose(out[:, 0], col, rtol=1e-5, atol=1e-5)
    byts = m * nc * (4 if homo else 8)
    print(f'batched gather m={m} nnz/row={nc} {"homo" if homo else "hetero"} B={nb} fire={fire}: {dt*1e3:.2f} ms '
          f'({byts/dt/1e9:.0f} GB/s of one matrix pass); one column alone {dt1*1e3:.2f} ms -> per-column route would be '
          f'{dt1*nb*1e3:.0f} ms; column 0 matches: {ok}', flush=True)
    del w, idx, ptr
    torch.cuda.empty_cache()
