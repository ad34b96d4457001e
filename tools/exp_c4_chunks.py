#!/usr/bin/env python3
"""Would the binned route (C4) gain from working in row chunks whose bins stay in the 256 MiB Infinity Cache between pass B
(written) and pass C (read back)?  Emulation without touching the kernels: the spike vector is split into C disjoint parts
(contiguous ranges of the pre population) and C whole binned steps run back to back; every step pays the fixed costs again
(output reset, compaction over all N neurons, the output slice stores of pass C), so this is a lower bound on the gain."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
from bench import gen_fixed_num_on_device
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(0)
n, K = 10_000_000, 1000
for homo in (False, True):
    w, idx = gen_fixed_num_on_device(n, K, n, homo, dev, g)
    conn = be.FixedNumPerPre((w, idx), shape=(n, n), check_indices=False)
    conn.prepare()
    spk = torch.rand(n, device=dev, generator=g) < 0.01
    for C in (1, 2, 3, 4, 6, 8):
        parts = []
        for c in range(C):
            p = torch.zeros_like(spk)
            lo, hi = n * c // C, n * (c + 1) // C
            p[lo:hi] = spk[lo:hi]
            parts.append(be.BinaryArray(p))
        def step():
            return [ev @ conn for ev in parts]
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f'{"homo" if homo else "hetero"} chunks={C}: {dt*1e6:.0f} us for all chunks', flush=True)
    del conn, w, idx
    torch.cuda.empty_cache()
