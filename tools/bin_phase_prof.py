#!/usr/bin/env python3
"""Phase-by-phase time of k_bin_stream, pass B of the binned route (diagnostic build).  On the GPU box:
    BE_HIPCC_FLAGS=-DBE_BIN_PROF python tools/bin_phase_prof.py [--homo]
rebuilds the library with the in-kernel stamps (s_memtime, 100 MHz), runs the C4 shape and prints the mean time a wave
spends in each phase per step, and the append-loop / flush counts per chunk of 64 entries.  Rebuild without the flag afterwards (the shipped library carries no stamps)."""
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
n, K = 10_000_000, int(os.environ.get('BE_PROF_K', 1000))
n_post = int(os.environ.get('BE_PROF_NPOST', n))      # BE_PROF_K=125 BE_PROF_NPOST=1250000: one post slice of an 8-way cut
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(7)
w, idx = gen_fixed_num_on_device(n, K, n_post, homo, dev, g)
conn = be.FixedNumPerPre((w, idx), shape=(n, n_post), check_indices=False).prepare()
print(type(conn.buffers.get('scatter_plan')).__name__)
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
t = buf.reshape(256, 8).astype(np.float64) / steps
us = t / 100.0 / 16.0                 # us per step per wave (s_memtime ticks at 100 MHz; 16 waves per workgroup add up)
names = {0: 'task header (ids, row bounds, scan)', 1: 'issue loads', 2: 'wait for the loads', 3: 'append + flush', 7: 'ticket / tail'}
print('homo' if homo else 'hetero', 'per step, mean over workgroups, us per wave:')
for i, nme in names.items():
    print(f'  {nme:36s} {us[:, i].mean():8.1f}   (min {us[:, i].min():.1f} max {us[:, i].max():.1f})')
print(f'  total {us[:, [0, 1, 2, 3, 7]].sum(axis=1).mean():.1f}')
chunks = t[:, 6].mean()
print(f'  chunks per workgroup {chunks:.0f}; append-loop iterations per chunk {t[:, 4].mean() / max(chunks, 1):.2f}; '
      f'flushes per chunk {t[:, 5].mean() / max(chunks, 1):.2f}')
