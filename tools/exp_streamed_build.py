#!/usr/bin/env python3
"""A matrix whose raw CSR and plan do NOT fit one GPU together, planned from row blocks and used from the plan alone.

Default: 1.5M x 1.5M, 15 000 entries per row (1 % density, U[0,1) weights) = 2.25e10 entries: raw CSR 180 GB + d8 plan ~130 GB
> 288 GB.  `ScatterPlan.build_from_blocks` sees one block of rows at a time (generated on the device from a per-block seed, twice:
count pass and fill), `PlannedMatrix` serves `spikes @ M`; the last step is checked against a float64 `index_add` over the active
rows of every regenerated block.
    python tools/exp_streamed_build.py [n] [per_row] [rows_per_block]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
from brainevent_amd._csr import ScatterPlan, PlannedMatrix

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_500_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 15_000
RB = int(sys.argv[3]) if len(sys.argv) > 3 else 50_000
dev = torch.device('cuda', 0)


def get_block(r0, r1):
    g = torch.Generator(device=dev)
    g.manual_seed(1000 + r0)
    idx = torch.randint(0, n, ((r1 - r0) * K,), dtype=torch.int32, device=dev, generator=g)
    w = torch.empty((r1 - r0) * K, dtype=torch.float32, device=dev).uniform_(0.0, 1.0, generator=g)
    return w, idx, None                                   # rows of exactly K entries


torch.cuda.synchronize(); t0 = time.perf_counter()
plan = ScatterPlan.build_from_blocks(get_block, RB, shape=(n, n), nnz=n * K, max_row_len=K, homo=False)
torch.cuda.synchronize(); t_build = time.perf_counter() - t0
M = PlannedMatrix(plan)
free, total = torch.cuda.mem_get_info()
print(f'{n} x {n}, {K} per row = {n * K:.3g} entries: raw CSR would be {n * K * 8 / 1e9:.0f} GB; plan {plan.nbytes() / 1e9:.1f} GB '
      f'(layout {plan.layout}, {plan.n_slices} slices of {plan.slice_width}), built from blocks of {RB} rows in {t_build:.1f} s '
      f'(generation of every block twice included); device memory in use {(total - free) / 1e9:.0f} of {total / 1e9:.0f} GB', flush=True)
g = torch.Generator(device=dev); g.manual_seed(5)
spikes = [torch.rand(n, device=dev, generator=g) < 0.01 for _ in range(8)]
for i in range(10):
    out = be.BinaryArray(spikes[i % 8]) @ M
torch.cuda.synchronize(); t0 = time.perf_counter()
steps = 100
for i in range(steps):
    out = be.BinaryArray(spikes[i % 8]) @ M
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
act = sum(int(s.sum()) for s in spikes) / 8
print(f'step {dt * 1e6:.0f} us, {act * K / dt / 1e9:.0f} Geff/s ({act:.0f} active rows of {K} entries)', flush=True)
ref = torch.zeros(n, dtype=torch.float64, device=dev)
s = spikes[(steps - 1) % 8]
for r0 in range(0, n, RB):
    r1 = min(n, r0 + RB)
    rows = s[r0:r1].nonzero().reshape(-1)
    if rows.numel():
        w, idx, _ = get_block(r0, r1)
        ref.index_add_(0, idx.view(r1 - r0, K)[rows].reshape(-1).long(), w.view(r1 - r0, K)[rows].reshape(-1).double())
        del w, idx
err = ((out.double() - ref).abs() / ref.abs().clamp_min(1e-30)).max().item()
print(f'last step vs float64 index_add over the regenerated blocks: max rel err {err:.2e} ({"ok" if err <= 1e-5 else "FAIL"})', flush=True)
