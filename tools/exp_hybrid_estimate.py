"""What would pass B cost if the C4 rows were pre-cut into S column slices (a per-matrix layout)?  A workgroup of the
sliced step streams (row, slice) blocks of K / S entries into n_bins / S bins — the same work per workgroup as the binned
step of a matrix with K / S entries per row over k / S outputs at S times the firing rate.  That matrix runs through the
existing route here: FixedNumPerPre m x (k / S), K / S per row, fire * S, bins of the C4 width, over block sizes (BE_BIN_CAP).
  BE_EXP_S=8 BE_BIN_CAP=64 python tools/exp_hybrid_estimate.py"""
import os, sys, time
import torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import brainevent_amd as be
import brainevent_amd._csr as C
from brainevent_amd import _array as A

dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(0)
S = int(os.environ.get('BE_EXP_S', '8'))
m = int(os.environ.get('BE_EXP_M', '10000000'))
K = 1000 // S
k = (m // S + 31) // 32 * 32
fire = 0.01 * S
homo = bool(int(os.environ.get('BE_EXP_HOMO', '0')))
idx = torch.empty((m, K), dtype=torch.int32, device=dev)
for lo in range(0, m, 1_000_000):
    hi = min(m, lo + 1_000_000)
    idx[lo:hi] = torch.randint(0, k, (hi - lo, K), dtype=torch.int32, device=dev, generator=g)
w = torch.ones(1, device=dev) if homo else torch.empty((m, K), device=dev).uniform_(0, 1, generator=g)
spikes = [(torch.rand(m, device=dev, generator=g) < fire).to(torch.uint8) for _ in range(4)]
ws = C.BinnedScatter(w.reshape(-1), m, k, m * K, max_active_fraction=min(1.0, 1.3 * fire), indices=idx.reshape(-1))
out = torch.empty(k, device=dev)
def step(i):
    C._binned_call(ws, w.reshape(-1), idx.reshape(-1), None, K, spikes[i % 4], A.BE_SPIKE_BOOL, out)
for i in range(4):
    step(i)
torch.cuda.synchronize()
ref = torch.zeros(k, device=dev, dtype=torch.float64)
rows = spikes[3].nonzero().reshape(-1)
for lo in range(0, rows.numel(), 100_000):
    r = rows[lo:lo + 100_000]
    ref.index_add_(0, idx[r].reshape(-1).long(), (w.expand(m, K) if homo else w)[r].reshape(-1).double() if not homo else torch.ones(r.numel() * K, device=dev, dtype=torch.float64))
err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
t0 = time.perf_counter()
n = 20
for i in range(n):
    step(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f'S={S} m={m} k={k} K={K} fire={fire} homo={homo} bins={ws.n_slices} cap={os.environ.get("BE_BIN_CAP", "auto")}: '
      f'{dt * 1e6:.0f} us/step, {rows.numel() * K / dt / 1e9:.0f} Geff/s, rel err {err:.2e}', flush=True)
