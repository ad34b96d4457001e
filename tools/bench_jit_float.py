#!/usr/bin/env python3
"""Float-operand twins of the JIT-connectivity products at the C3 shape (4M x 4M, prob 1e-3): gather and scatter orientation."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
dev = torch.device('cuda', 0)
n, prob = 4_000_000, 1e-3
x = torch.randn(n, device=dev)
w = torch.tensor(1.0, device=dev)
for corder in (True, False):
    f = lambda: be.jitsmv(w, prob, x, 42, shape=(n, n), corder=corder)
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        f()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 3
    print(f'jitsmv 4M x 4M prob 1e-3, {"gather (corder=True)" if corder else "scatter (corder=False)"}: {t * 1e3:8.2f} ms = {n * n * prob / t / 1e9:6.1f} G edges/s', flush=True)
