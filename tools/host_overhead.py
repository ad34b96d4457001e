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

# JIT connectivity (scatter) and its 8-way shard, fixed-number connectivity (direct + binned), dense
def issue_cost(label, f, n=1000):
    for _ in range(30):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_total = time.perf_counter() - t0
    print(f'{label}: host issue {t_issue / n * 1e6:.1f} us/call, end-to-end {t_total / n * 1e6:.1f} us/call')

M = be.JITCScalarR((np.float32(1.0), 0.01, 42), shape=(20000, 20000), corder=True)
s2 = torch.tensor(rng.random(20000) < 0.01, device='cuda')
issue_cost('JITCScalarR scatter, small', lambda: be.BinaryArray(s2) @ M)
sh = M.scatter_shard(8, 0)
issue_cost('JITC scatter shard 1/8, small', lambda: be.BinaryArray(s2) @ sh)
idx = torch.tensor(rng.integers(0, 50000, (20000, 20)).astype(np.int32), device='cuda')
F = be.FixedNumPerPre((torch.ones(1, device='cuda'), idx), shape=(20000, 50000))
issue_cost('FixedNumPerPre scatter (direct), small', lambda: be.BinaryArray(s2) @ F)
W = torch.randn((2000, 512), device='cuda')
issue_cost('dense mv, small', lambda: be.BinaryArray(spk) @ W)
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    be.BinaryArray(s2) @ M
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
