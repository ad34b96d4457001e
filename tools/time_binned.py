#!/usr/bin/env python3
"""ms per step of `spikes @ FixedNumPerPre` on the binned route for a given shape (default: one post slice of an 8-way cut of C4).
usage: python tools/time_binned.py [--n N] [--k K] [--n-post P] [--homo] [--fire F] [--steps S]   (wrap in tools/prof_any.sh for kernels)"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
from brainevent_amd import _csr as C
from bench import gen_fixed_num_on_device

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=10_000_000); ap.add_argument('--k', type=int, default=125)
ap.add_argument('--n-post', type=int, default=1_250_000); ap.add_argument('--homo', action='store_true')
ap.add_argument('--fire', type=float, default=0.01); ap.add_argument('--steps', type=int, default=100)
ap.add_argument('--check', action='store_true')
a = ap.parse_args()
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(7)
w, idx = gen_fixed_num_on_device(a.n, a.k, a.n_post, a.homo, dev, g)
conn = be.FixedNumPerPre((w, idx), shape=(a.n, a.n_post), check_indices=False)
conn.buffers['scatter_plan'] = C.BinnedScatter(w.reshape(-1), a.n, a.n_post, a.n * a.k, indices=idx.reshape(-1), row_len=a.k)
ws = conn.buffers['scatter_plan']
spk = [be.BinaryArray(torch.rand(a.n, device=dev, generator=g) < a.fire) for _ in range(10)]
for i in range(10):
    out = spk[i % 10] @ conn
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(a.steps):
    out = spk[i % 10] @ conn
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / a.steps * 1e3
ws.check_status()
line = f"binned {'homo' if a.homo else 'hetero'} n={a.n} K={a.k} n_post={a.n_post} fire={a.fire}: bins {ws.n_slices} acc32 {ws.acc32}: {ms:.4f} ms/step"
if a.check and not a.homo:
    s = spk[(a.steps - 1) % 10].value
    rows = torch.nonzero(s).flatten()
    ref = torch.zeros(a.n_post, dtype=torch.float64, device=dev)
    for lo in range(0, rows.numel(), 20000):
        r = rows[lo:lo + 20000]
        ref.index_add_(0, idx[r].flatten().long(), w[r].flatten().double())
    rel = ((out.double() - ref).abs() / ref.abs().clamp_min(1e-30)).max().item()
    line += f", max rel err vs f64 {rel:.2e}"
print(line)
