#!/usr/bin/env python3
"""Full-size parity properties for config C4 on one GPU (BinaryArray @ FixedNumPerPre, N = 10M, K = 1000, 1 % firing):
the binned route (no per-matrix layout) is checked on the device against
  (a) an exact integer histogram of the active rows' targets (homogeneous weight 1: counts must match EXACTLY),
  (b) a float64 index_add of the active rows' (target, weight) pairs (heterogeneous: relative error <= 1e-5),
and, for one of eight post slices (what one rank of the 8-GPU partition holds), the sharded product against the same
reference restricted to the slice.  Run on an MI355X:  python tools/check_c4_parity.py [n] [k]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
from brainevent_amd import _csr as C, _dist as D

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(11)
idx = torch.empty((n, K), dtype=torch.int32, device=dev)
for lo in range(0, n, 200_000):
    hi = min(n, lo + 200_000)
    idx[lo:hi] = torch.randint(0, n, (hi - lo, K), dtype=torch.int32, device=dev, generator=g)

for homo in (True, False):
    w = torch.ones(1, device=dev) if homo else torch.empty((n, K), device=dev).uniform_(0, 1, generator=g)
    conn = be.FixedNumPerPre((w, idx), shape=(n, n), check_indices=False).prepare()
    print('route:', type(conn.buffers['scatter_plan']).__name__, flush=True)
    assert isinstance(conn.buffers['scatter_plan'], C.BinnedScatter)
    for step in range(2):
        spk = torch.rand(n, device=dev, generator=g) < 0.01
        out = be.BinaryArray(spk) @ conn
        rows = torch.nonzero(spk).flatten()
        if homo:
            ref = torch.zeros(n, dtype=torch.int64, device=dev)
            for lo in range(0, rows.numel(), 20_000):
                ref += torch.bincount(idx[rows[lo:lo + 20_000]].flatten().long(), minlength=n)
            assert torch.equal(out.to(torch.int64), ref), 'homo counts differ from the integer histogram'
            print(f'homo  step {step}: {int(ref.sum())} updates, counts == integer histogram (exact)', flush=True)
        else:
            ref = torch.zeros(n, dtype=torch.float64, device=dev)
            for lo in range(0, rows.numel(), 20_000):
                r = rows[lo:lo + 20_000]
                ref.index_add_(0, idx[r].flatten().long(), w[r].flatten().double())
            rel = ((out.double() - ref).abs() / ref.abs().clamp_min(1e-30)).max().item()
            assert rel <= 1e-5, rel
            print(f'hetero step {step}: {rows.numel() * K} updates, max rel err vs f64 index_add {rel:.2e}; '
                  f'checksum {out.double().sum().item():.6f} vs {ref.sum().item():.6f}', flush=True)
    if homo:      # one rank of the 8-way post-slice partition (ragged shard -> CSR), same spikes
        sw, si, sp, sshape = D.shard_fixed_num_by_post(w, idx, (n, n), 8, 3)
        lo, hi = D.post_slice_bounds(n, 8, 3)
        shard = be.CSR((sw, si, sp), shape=sshape, check_structure=False).prepare()
        print('shard route:', type(shard.buffers['scatter_plan']).__name__, 'nnz', si.numel(), flush=True)
        out_s = be.BinaryArray(spk) @ shard
        assert torch.equal(out_s.to(torch.int64), ref[lo:hi]), 'shard counts differ from the slice of the histogram'
        print(f'post slice 3/8 [{lo}, {hi}): shard output == slice of the full histogram (exact)', flush=True)
        del shard, sw, si, sp
    del conn, w
    torch.cuda.empty_cache()
print('C4 parity ok')
