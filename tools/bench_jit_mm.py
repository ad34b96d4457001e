#!/usr/bin/env python3
"""JIT-connectivity products with a batch of event vectors (mm ops, lane stride 4): scatter and gather orientations."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
dev = torch.device('cuda', 0)
n, prob, nb = int(os.environ.get('N', 1_000_000)), 0.001, 32
for corder in (True, False):
    M = be.JITCScalarR((np.float32(1.0), prob, 42), shape=(n, n), corder=corder)
    S = torch.rand((nb, n), device=dev) < 0.01
    for _ in range(2):
        out = be.BinaryArray(S) @ M
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        out = be.BinaryArray(S) @ M
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    edges = float(out.double().sum().item())
    kind = 'scatter' if corder else 'gather'
    gen = n * n * prob
    print(f'S[{nb},{n}] @ JITCScalarR(corder={corder}) ({kind}): {dt*1e3:.2f} ms, delivered {edges:.3g} updates '
          f'({edges/dt/1e9:.0f} Geff/s); generated edges/s (gather walks all {gen:.3g}): {gen/dt/1e9:.0f} G', flush=True)
