#!/usr/bin/env python3
"""Float-operand twins: achieved matrix-stream bandwidth of `csr @ x` (gather) and rate of `x @ csr` (scatter through float
atomics) — secondary operators, not on the event-driven hot path."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
from bench import gen_csr_on_device
dev = torch.device('cuda', 0)


def timeit(f, n=5):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for n, n_conn in ((1_000_000, 10_000), (1_000_000, 1000), (1_000_000, 100), (4_000_000, 16)):
    for homo in (False, True):
        w, idx, ptr = gen_csr_on_device(n, n, n_conn, homo, 0, dev)
        x = torch.randn(n, device=dev)
        nnz = n * n_conn
        bytes_nt = nnz * (4 if homo else 8)
        t_nt = timeit(lambda: be.csrmv(w, idx, ptr, x, shape=(n, n)))
        t_t = timeit(lambda: be.csrmv(w, idx, ptr, x, shape=(n, n), transpose=True), n=2) if nnz <= 2_000_000_000 else float('nan')
        X = torch.randn(n, 8, device=dev)
        t_mm = timeit(lambda: be.csrmm(w, idx, ptr, X, shape=(n, n)), n=2) if nnz <= 2_000_000_000 else float('nan')
        print(f'n={n} row={n_conn} {"homo" if homo else "hetero"}: csr @ x {t_nt * 1e3:9.3f} ms = {bytes_nt / t_nt / 1e9:7.0f} GB/s of matrix stream | '
              f'x @ csr {t_t * 1e3:9.3f} ms = {nnz / t_t / 1e9:6.1f} G atomics/s | csr @ X[:, 8] {t_mm * 1e3:9.3f} ms', flush=True)
        del w, idx, ptr, x, X
        torch.cuda.empty_cache()
