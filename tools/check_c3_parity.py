#!/usr/bin/env python3
"""Full-size parity property for config C3 (BinaryArray @ JITCScalarR, N = 4M, prob = 0.1 %):
materialise the drawn connectivity as CSR on the device (be_jitc_csr_count / _fill) and check that the on-the-fly
scatter delivers exactly the same integer counts as the stored-matrix scatter for fresh spike vectors.
Run on an MI355X:  python tools/check_c3_parity.py [N]   (N = 4_000_000 needs ~70 GB for the CSR)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import brainevent_amd as be

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
prob, seed = 0.001, 42
M = be.JITCScalarR((np.float32(1.0), prob, seed), shape=(n, n), corder=True)
t0 = time.perf_counter()
S = M.materialize('mv')          # native form (CSR for this orientation): no re-encoding of 1.6e10 entries
torch.cuda.synchronize()
print(f'materialised {S.nse} edges ({S.nse / (n * n * prob):.4f} of n*n*prob) in {time.perf_counter() - t0:.2f} s', flush=True)
g = torch.Generator(device='cuda'); g.manual_seed(1)
for step in range(3):
    spk = torch.rand(n, device='cuda', generator=g) < 0.01
    a = be.BinaryArray(spk) @ M          # on the fly (LDS residue-class scatter)
    b = be.BinaryArray(spk) @ S          # stored matrix (plan / binned / direct route, whichever applies)
    assert torch.equal(a, b), (step, (a - b).abs().max().item())
    print(f'step {step}: {int(a.sum().item())} edges delivered, on-the-fly == materialised (exact)', flush=True)
print('C3 parity ok')
