#!/usr/bin/env python3
"""C4 (FixedNumPerPre N = 10M, K = 1000, 1 % firing) through the binned route as a function of the capacity its regions are
sized for (BinnedScatter(max_active_fraction=...)): does the spacing of the regions in memory matter to pass C?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
from brainevent_amd import _csr as C
from bench import gen_fixed_num_on_device
homo = '--homo' in sys.argv
n, K = 10_000_000, 1000
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(7)
w, idx = gen_fixed_num_on_device(n, K, n, homo, dev, g)
conn = be.FixedNumPerPre((w, idx), shape=(n, n), check_indices=False)
spk = [torch.rand(n, device=dev, generator=g) < 0.01 for _ in range(8)]
for frac in (0.05, 0.025, 0.0125):
    conn.buffers['scatter_plan'] = C.BinnedScatter(w.reshape(-1), n, n, n * K, indices=idx.reshape(-1), max_active_fraction=frac)
    for i in range(5):
        be.BinaryArray(spk[i % 8]) @ conn
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(40):
        be.BinaryArray(spk[i % 8]) @ conn
    torch.cuda.synchronize()
    print(f"{'homo' if homo else 'hetero'} sized for {frac:g} firing: {(time.perf_counter() - t0) / 40 * 1e6:.1f} us/step", flush=True)
