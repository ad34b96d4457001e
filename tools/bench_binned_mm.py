#!/usr/bin/env python3
"""Batched scatter on the binned route: spk[B] @ FixedNumPerPre (N = 1M, K = 100; N = 10M, K = 1000) next to one vector."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
from brainevent_amd import _csr as C
from bench import gen_fixed_num_on_device
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(3)
for (n, K) in ((1_000_000, 100), (10_000_000, 1000)):
    for homo in (False, True):
        w, idx = gen_fixed_num_on_device(n, K, n, homo, dev, g)
        conn = be.FixedNumPerPre((w, idx), shape=(n, n), check_indices=False)
        conn.buffers['scatter_plan'] = C.BinnedScatter(w.reshape(-1), n, n, n * K, indices=idx.reshape(-1))
        res = []
        for nb in (1, 8, 32) if n <= 1_000_000 else (1, 8):
            S = torch.rand((nb, n), device=dev, generator=g) < 0.01
            ev = be.BinaryArray(S[0].contiguous()) if nb == 1 else be.BinaryArray(S)
            for _ in range(3):
                out = ev @ conn
            torch.cuda.synchronize(); t0 = time.perf_counter()
            reps = 20
            for _ in range(reps):
                out = ev @ conn
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            res.append(f'B={nb}: {dt*1e6:.0f} us')
        print(f"binned mm N={n} K={K} {'homo' if homo else 'hetero'}: " + ', '.join(res), flush=True)
        del conn, w, idx
        torch.cuda.empty_cache()
