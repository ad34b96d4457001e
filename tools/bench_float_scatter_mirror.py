#!/usr/bin/env python3
"""`x @ csr` (the scatter direction with a dense operand): float atomics against the gather over the automatically built mirror."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
from brainevent_amd import _csr as C
from bench import gen_csr_on_device
dev = torch.device('cuda', 0)


def timeit(f, n=5):
    f(); f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for n, row in ((1_000_000, 200), (1_000_000, 1000)):
    w, idx, ptr = gen_csr_on_device(n, n, row, False, 0, dev)
    x = torch.randn(n, device=dev)
    saved = C.AUTO_MIRROR_MIN_NNZ
    C.AUTO_MIRROR_MIN_NNZ = None                      # no automatic mirror: float atomics
    csr = be.CSR((w, idx, ptr), shape=(n, n), check_structure=False)
    t_atomic = timeit(lambda: x @ csr)
    C.AUTO_MIRROR_MIN_NNZ = saved
    csr2 = be.CSR((w, idx, ptr), shape=(n, n), check_structure=False)
    t0 = time.perf_counter(); r = x @ csr2; torch.cuda.synchronize(); t_first = time.perf_counter() - t0
    mr = csr2.buffers.get('mirror')
    t_mirror = timeit(lambda: x @ csr2)
    print(f'n={n} row={row} ({n * row:.1e} entries): atomics {t_atomic * 1e3:8.3f} ms | with the automatic mirror {t_mirror * 1e3:8.3f} ms '
          f'(first call incl. the build {t_first * 1e3:.0f} ms; mirror {"none" if mr is None else ("plan only" if mr.released else "raw arrays kept")})', flush=True)
    del csr, csr2, w, idx, ptr, x, r, mr
    torch.cuda.empty_cache()
