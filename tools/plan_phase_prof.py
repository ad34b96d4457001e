#!/usr/bin/env python3
"""Phase-by-phase time of k_plan_accumulate_d8 as workgroup 0..255's first thread sees it (diagnostic build).  On the GPU box:
    BE_HIPCC_FLAGS=-DBE_PLAN_PROF python tools/plan_phase_prof.py [N] [K] [N_POST]
Rebuild without the flag afterwards (the shipped library carries no stamps)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brainevent_amd import _lib
_lib.build(force=True)
import brainevent_amd as be
from brainevent_amd import _csr as C, _array as A
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
n_post = int(sys.argv[3]) if len(sys.argv) > 3 else n      # e.g. 1000000 1250 125000: one post slice of an 8-way cut of C2
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(0)
idx = torch.randint(0, n_post, (n, K), dtype=torch.int32, device=dev, generator=g)
w = torch.empty((n, K), device=dev).uniform_(0, 1, generator=g)
plan = C.ScatterPlan.build(w, idx.reshape(-1), None, shape=(n, n_post), row_len=K)
if os.environ.get('BE_PLAN_SEGT') == '1':      # experiment: slice-major segment table
    plan.seg = plan.seg.view(n, plan.n_slices, 2).permute(1, 0, 2).contiguous().view(-1)
spikes = [(torch.rand(n, device=dev, generator=g) < 0.01).to(torch.uint8) for _ in range(4)]
out = torch.empty(n_post, dtype=torch.float32, device=dev)
for i in range(5):
    C._plan_call(plan, w, spikes[i % 4], A.BE_SPIKE_BOOL, out)
torch.cuda.synchronize()
f = _lib.fn('be_debug_plan_prof', ctypes.c_int, [ctypes.c_void_p, ctypes.c_int])
f(None, 1)
steps = 20
for i in range(steps):
    C._plan_call(plan, w, spikes[i % 4], A.BE_SPIKE_BOOL, out)
torch.cuda.synchronize()
buf = np.zeros(256 * 8, np.uint64)
f(buf.ctypes.data, 0)
t = buf.reshape(256, 8).astype(np.float64) / steps
names = ['zero LDS', 'spike count / list', 'first row ids + segments', "wave 0's blocks", 'slowest wave', 'store partial sums']
print(f'N={n} K={K}: {plan.n_slices} slices x {plan.default_parts()} parts, layout {plan.layout}, hint {plan.block_hint}; ticks per step, mean over 256 workgroups')
tot = t[:, :6].sum(axis=1).mean()
for i, nme in enumerate(names):
    print(f'  {nme:28s} {t[:, i].mean():10.0f}  ({100 * t[:, i].mean() / tot:5.1f} %)')
print(f'  total {tot:.0f} ticks')
