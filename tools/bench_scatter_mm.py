#!/usr/bin/env python3
"""Batched scatter (binary_csrmm, transpose=True) through the plan: one launch per stage for the whole batch
(gridDim.y = batch); reports us per batch column next to the single-vector step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
from bench import gen_csr_on_device
dev = torch.device('cuda', 0)
for (m, k, nc, homo) in ((1_000_000, 1_000_000, 1000, False), (1_000_000, 1_000_000, 1000, True), (300_000, 300_000, 3000, False)):
    w, idx, ptr = gen_csr_on_device(m, k, nc, homo, 3, dev)
    csr = be.CSR((w, idx, ptr), shape=(m, k), check_structure=False).prepare()
    res = []
    for nb in (1, 4, 16, 32):
        B = torch.rand((m, nb), device=dev) < 0.01
        ev = be.BinaryArray(B[:, 0].contiguous()) if nb == 1 else be.BinaryArray(B.T.contiguous())
        for _ in range(3):
            out = ev @ csr
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            out = ev @ csr
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        res.append(f'B={nb}: {dt*1e6:.0f} us ({dt*1e6/nb:.0f}/col)')
    print(f'scatter mm m={m} k={k} nnz/row={nc} {"homo" if homo else "hetero"}: ' + ', '.join(res), flush=True)
    del csr, w, idx, ptr
    torch.cuda.empty_cache()
