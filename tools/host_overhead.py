#!/usr/bin/env python3
"""Host-side issue cost of one `spk @ csr` (how long Python + ctypes + the launch calls take, GPU work excluded)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
from brainevent_amd import _csr as C
C.PLAN_MIN_NNZ = 1
rng = np.random.default_rng(0)
m, k, nc = 2000, 70000, 300
indptr = (np.arange(m + 1) * nc).astype(np.int32)
indices = rng.integers(0, k, m * nc).astype(np.int32)
w = rng.random(m * nc).astype(np.float32)
csr = be.CSR((torch.tensor(w, device='cuda'), torch.tensor(indices, device='cuda'), torch.tensor(indptr, device='cuda')), shape=(m, k)).prepare()
spk = torch.tensor(rng.random(m) < 0.01, device='cuda')
for _ in range(50):
    out = be.BinaryArray(spk) @ csr
torch.cuda.synchronize()
n = 2000
t0 = time.perf_counter()
for _ in range(n):
    out = be.BinaryArray(spk) @ csr
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_total = time.perf_counter() - t0
print(f'planned route, tiny matrix: host issue {t_issue / n * 1e6:.1f} us/call, end-to-end {t_total / n * 1e6:.1f} us/call')
