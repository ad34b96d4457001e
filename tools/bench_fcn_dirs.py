#!/usr/bin/env python3
"""The four FixedNumConn products at one size (N x N, K per row, 1 % firing): which kernel family each one lands on."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
dev = torch.device('cuda', 0)
n, K = int(os.environ.get('N', 1_000_000)), int(os.environ.get('K', 100))
g = torch.Generator(device=dev); g.manual_seed(0)
def timeit(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for homo in (False, True):
    idx = torch.randint(0, n, (n, K), dtype=torch.int32, device=dev, generator=g)
    w = torch.ones(1, device=dev) if homo else torch.rand((n, K), device=dev, generator=g)
    spk = torch.rand(n, device=dev, generator=g) < 0.01
    S = torch.rand((8, n), device=dev, generator=g) < 0.01
    for cls in (be.FixedNumPerPre, be.FixedNumPerPost):
        conn = cls((w, idx), shape=(n, n), check_indices=False) if cls is be.FixedNumPerPre else cls((w, idx), shape=(n, n), check_indices=False)
        ev = be.BinaryArray(spk)
        t1 = timeit(lambda: ev @ conn)
        t2 = timeit(lambda: conn @ ev)
        evb = be.BinaryArray(S)
        t3 = timeit(lambda: evb @ conn, 3)
        print(f'{cls.__name__} N={n} K={K} {"homo" if homo else "hetero"}: spk @ conn {t1*1e6:.0f} us | conn @ spk {t2*1e6:.0f} us | spk[8] @ conn {t3*1e6:.0f} us', flush=True)
        del conn
    del idx, w
    torch.cuda.empty_cache()
