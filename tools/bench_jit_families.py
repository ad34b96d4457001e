#!/usr/bin/env python3
"""JIT-connectivity mv products of the three weight families, both orientations (n x n, prob 1e-3, 1 % firing)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
dev = torch.device('cuda', 0)
n, prob = int(os.environ.get('N', 1_000_000)), 0.001
spk = torch.rand(n, device=dev) < 0.01
def timeit(f, reps=5):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for name, mk in (('scalar', lambda c: be.JITCScalarR((np.float32(1.0), prob, 42), shape=(n, n), corder=c)),
                 ('uniform', lambda c: be.JITCUniformR((np.float32(0.0), np.float32(1.0), prob, 42), shape=(n, n), corder=c)),
                 ('normal', lambda c: be.JITCNormalR((np.float32(0.0), np.float32(1.0), prob, 42), shape=(n, n), corder=c))):
    for corder in (True, False):
        M = mk(corder)
        ev = be.BinaryArray(spk)
        t = timeit(lambda: ev @ M)
        t2 = timeit(lambda: M @ ev)
        print(f'{name} corder={corder}: spk @ M {t*1e3:.3f} ms | M @ spk {t2*1e3:.3f} ms', flush=True)
