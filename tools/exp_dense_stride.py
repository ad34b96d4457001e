#!/usr/bin/env python3
"""W @ S.T (fp16, 32 batch columns, MFMA): does a power-of-two row stride (k = 65536 -> 128 KiB between the 32 rows of a
load instruction) cost bandwidth?  Same row count, k = 65536 vs 65536 + 64 / + 192."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
dev = torch.device('cuda', 0)
m = 65536
for k in (65536, 65536 + 64, 65536 + 192, 65536 + 1024):
    W = torch.empty((m, k), dtype=torch.float16, device=dev).normal_()
    S = torch.rand((k, 32), device=dev) < 0.5
    ev = be.BinaryArray(S)
    for _ in range(3): y = W @ ev
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): y = W @ ev
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    St = torch.rand((32, m), device=dev) < 0.5
    evt = be.BinaryArray(St)
    for _ in range(3): z = evt @ W
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): z = evt @ W
    torch.cuda.synchronize(); dt2 = (time.perf_counter() - t0) / 10
    print(f'k={k}: W @ S.T {dt*1e3:.3f} ms ({m*k*2/dt/1e9:.0f} GB/s) | S @ W {dt2*1e3:.3f} ms ({m*k*2/dt2/1e9:.0f} GB/s)', flush=True)
    del W, S, St
    torch.cuda.empty_cache()
