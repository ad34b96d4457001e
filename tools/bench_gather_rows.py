#!/usr/bin/env python3
"""Gather direction (CSR @ spikes) over the row length at 2e8 stored entries: the lanes-per-row tiers of be_csr.hip
(k_csrmv_nt_vec with 2 ... 32 lanes per row up to 512 entries per row, a wave per row beyond)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
from bench import gen_csr_on_device
dev = torch.device('cuda', 0)
for homo in (False, True):
    for nc in [int(x) for x in os.environ.get('BE_EXP_ROWS', '4,8,12,24,48,100,250,500,1000').split(',')]:
        m = 200_000_000 // nc
        k = 1_000_000
        w, idx, ptr = gen_csr_on_device(m, k, nc, homo, 3, dev)
        spk = torch.rand(k, device=dev) < 0.01
        if os.environ.get('BE_EXP_FIXED') == '1':      # rows of one known length (FixedNumPerPost @ spk: no indptr)
            conn = be.FixedNumPerPost((w if homo else w.view(m, nc), idx.view(m, nc)), shape=(k, m), check_indices=False)
            ev = be.BinaryArray(spk)
            call = lambda: ev @ conn
        else:
            call = lambda: be.binary_csrmv(w, idx, ptr, spk, shape=(m, k), transpose=False)
        for _ in range(3):
            out = call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            out = call()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        byts = m * nc * (4 if homo else 8)
        print(f'gather {"homo" if homo else "hetero"} nnz/row={nc:5d} m={m}: {dt*1e3:.3f} ms, {byts/dt/1e9:.0f} GB/s of matrix stream', flush=True)
        del w, idx, ptr, out
        torch.cuda.empty_cache()
