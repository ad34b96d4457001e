#!/usr/bin/env python3
"""Phase-by-phase time of k_bin_rows (diagnostic build).  On the GPU box:
    BE_HIPCC_FLAGS=-DBE_BIN_PROF python tools/bin_phase_prof.py [--homo]
rebuilds the library with the in-kernel stamps (s_memtime, 100 MHz), runs the C4 shape and prints the mean share of each
phase per batch.  Rebuild without the flag afterwards (the shipped library carries no stamps)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brainevent_amd import _lib
_lib.build(force=True)
import brainevent_amd as be
from bench import gen_fixed_num_on_device

homo = '--homo' in sys.argv
n, K = 10_000_000, 1000
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(7)
w, idx = gen_fixed_num_on_device(n, K, n, homo, dev, g)
conn = be.FixedNumPerPre((w, idx), shape=(n, n), check_indices=False).prepare()
spk = [torch.rand(n, device=dev, generator=g) < 0.01 for _ in range(4)]
for i in range(5):
    be.BinaryArray(spk[i % 4]) @ conn
torch.cuda.synchronize()
f = _lib.fn('be_debug_bin_prof', ctypes.c_int, [ctypes.c_void_p, ctypes.c_int])
f(None, 1)
steps = 20
for i in range(steps):
    be.BinaryArray(spk[i % 4]) @ conn
torch.cuda.synchronize()
buf = np.zeros(256 * 8, np.uint64)
f(buf.ctypes.data, 0)
t = buf.reshape(256, 8).astype(np.float64) / steps / 100.0      # us per step per workgroup (s_memtime ticks at 100 MHz)
names = ['form batch', 'issue loads', 'loads land + histogram', 'scan + reserve', 'placement', 'copy-out']
print('homo' if homo else 'hetero', 'per step, mean over workgroups (us):')
for i, nme in enumerate(names):
    print(f'  {nme:26s} {t[:, i].mean():8.1f}   (min {t[:, i].min():.1f} max {t[:, i].max():.1f})')
print(f'  total {t.sum(axis=1).mean():.1f}')
