"""Round-4 JITC measurements: (a) the reference's default orientation at C3 (`spk @ JITCScalarR(corder=False)`: the gather walk)
on the fly vs on the stored twin (`prepare()`); (b) the scalar mm gather at n = 1M for 8 / 32 batch columns."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import brainevent_amd as be

def ms(fn, n=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(3)
if 'c3' in sys.argv or len(sys.argv) == 1:
    n = 4_000_000
    M = be.JITCScalarR((np.float32(1.0), 0.001, 42), shape=(n, n), corder=False)
    spk = be.BinaryArray(torch.rand(n, device=dev, generator=g) < 0.01)
    a = spk @ M
    print(f'C3 default orientation (spk @ JITCScalarR(corder=False)), on the fly: {ms(lambda: spk @ M, n=3, warm=1):.3f} ms/step', flush=True)
    torch.cuda.synchronize(); t = time.perf_counter()
    M.prepare(); b = spk @ M; torch.cuda.synchronize()
    print(f'prepare() + first product (materialise, mirror, plan): {time.perf_counter() - t:.2f} s; stored nse = {M.buffers["materialized_mv"].nse}', flush=True)
    print(f'C3 default orientation on the stored twin: {ms(lambda: spk @ M, n=50, warm=5):.4f} ms/step; equal to the walk: {bool(torch.equal(a, b))}', flush=True)
    print(f'C3 other direction (M @ spk) on the stored twin: {ms(lambda: M @ spk, n=50, warm=5):.4f} ms/step', flush=True)
    del M, a, b
    torch.cuda.empty_cache()
if 'mm' in sys.argv or len(sys.argv) == 1:
    n = 1_000_000
    for nc in (8, 32):
        B = torch.rand((n, nc), device=dev, generator=g) < 0.01
        f = lambda: be.binary_jitsmm(torch.tensor(1.0, device=dev), 0.001, B, 42, shape=(n, n), transpose=False, corder=True)
        print(f'scalar mm gather n = {n}, {nc} columns: {ms(f, n=3, warm=1):.3f} ms', flush=True)
