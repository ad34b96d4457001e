#!/usr/bin/env python3
"""Full-size parity properties for the headline config C2 (BinaryArray @ CSR, 1M x 1M, 1 % density, 1 % firing).
The CPU oracle cannot hold 80 GB, so at full size the planned route is checked on the device against
  (a) an exact integer histogram of the active rows' columns (homogeneous weight 1: counts must match EXACTLY),
  (b) a float64 index_add of the active rows' (column, weight) pairs (heterogeneous: relative error <= 1e-5),
  (c) the direct route (global atomics) of the same library,
  (d) bitwise repeatability of the planned route.
Run on an MI355X:  python tools/check_c2_parity.py [n]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
from brainevent_amd import _csr as C
from bench import gen_csr_on_device

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_conn = max(1, int(n * 0.01))
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(5)


def active_entries(indices, spk):
    rows = torch.nonzero(spk).flatten()
    pos = (rows[:, None] * n_conn + torch.arange(n_conn, device=dev)[None, :]).flatten()
    return pos, indices[pos].to(torch.int64)


for homo in (True, False):
    w, idx, ptr = gen_csr_on_device(n, n, n_conn, homo, 77, dev)
    csr = be.CSR((w, idx, ptr), shape=(n, n), check_structure=False).prepare()
    assert isinstance(csr.buffers['scatter_plan'], C.ScatterPlan)
    for step in range(2):
        spk = torch.rand(n, device=dev, generator=g) < 0.01
        out = be.BinaryArray(spk) @ csr
        out2 = be.BinaryArray(spk) @ csr
        assert torch.equal(out, out2), 'planned route is not bitwise repeatable'
        pos, cols = active_entries(idx, spk)
        if homo:
            ref = torch.bincount(cols, minlength=n)
            assert torch.equal(out.to(torch.int64), ref), 'homo counts differ from the integer histogram'
            print(f'homo  step {step}: {int(ref.sum())} updates, counts == integer histogram (exact), repeatable', flush=True)
        else:
            ref = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, cols, w[pos].double())
            rel = ((out.double() - ref).abs() / ref.abs().clamp_min(1e-30)).max().item()
            assert rel <= 1e-5, rel
            direct = be.binary_csrmv(w, idx, ptr, spk, shape=(n, n), transpose=True)      # workspace=None: global atomics
            rel_d = ((direct.double() - ref).abs() / ref.abs().clamp_min(1e-30)).max().item()
            assert rel_d <= 1e-5, rel_d
            print(f'hetero step {step}: {cols.numel()} updates, max rel err vs f64 index_add: planned {rel:.2e}, direct {rel_d:.2e}; '
                  f'checksum {out.double().sum().item():.6f} vs {ref.sum().item():.6f}', flush=True)
    del csr, w, idx, ptr
    torch.cuda.empty_cache()
print('C2 parity ok')
