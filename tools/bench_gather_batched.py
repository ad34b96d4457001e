#!/usr/bin/env python3
"""Batched gather (binary_csrmm, transpose=False) over the row length at 2e8 entries: per-column passes of the vector
kernel against the kernel fused over the batch (BE_FUSED_MIN_ROW selects from which average row length it is used)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
from bench import gen_csr_on_device
dev = torch.device('cuda', 0)
for nc in [int(x) for x in os.environ.get('BE_EXP_ROWS', '24,100,250,1000').split(',')]:
    m, k = 200_000_000 // nc, 1_000_000
    w, idx, ptr = gen_csr_on_device(m, k, nc, False, 3, dev)
    for nb in (8, 32):
        B = torch.rand((k, nb), device=dev) < 0.01
        for _ in range(2):
            out = be.binary_csrmm(w, idx, ptr, B, shape=(m, k), transpose=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out = be.binary_csrmm(w, idx, ptr, B, shape=(m, k), transpose=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f'batched gather nnz/row={nc} m={m} B={nb}: {dt*1e3:.2f} ms ({dt/nb*1e3:.3f} ms per column)', flush=True)
    del w, idx, ptr
    torch.cuda.empty_cache()
