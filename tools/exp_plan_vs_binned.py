"""Planned layout vs binned route as a function of the entries per (row, slice) block (FixedNumPerPre K = 1000, 1 % firing):
the measurement behind brainevent_amd._csr.choose_scatter_route / PLAN_MIN_SEGMENT."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import brainevent_amd as be
import brainevent_amd._csr as C
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(0)
K = 1000
for n in [int(x) for x in os.environ.get('BE_EXP_NS', '500000,1000000,1500000').split(',')]:
    idx = torch.empty((n, K), dtype=torch.int32, device=dev)
    for lo in range(0, n, 200_000):
        hi = min(n, lo + 200_000)
        idx[lo:hi] = torch.randint(0, n, (hi - lo, K), dtype=torch.int32, device=dev, generator=g)
    spikes = [torch.rand(n, device=dev, generator=g) < 0.01 for _ in range(10)]
    for homo in ((True,) if os.environ.get('BE_EXP_HOMO_ONLY') else (True, False)):
        w = torch.ones(1, device=dev) if homo else torch.empty((n, K), device=dev).uniform_(0, 1, generator=g)
        for seg_min in (1, 1000):
            C.PLAN_MIN_SEGMENT = C.PLAN_MIN_SEGMENT_HOMO = seg_min
            conn = be.FixedNumPerPre((w, idx), shape=(n, n), check_indices=False).prepare()
            for i in range(5):
                out = be.BinaryArray(spikes[i]) @ conn
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(30):
                out = be.BinaryArray(spikes[i % 10]) @ conn
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 30
            ws = conn.buffers['scatter_plan']
            desc = type(ws).__name__ + (f' {ws.n_slices} slices, {K / ws.n_slices:.1f} per block' if isinstance(ws, C.ScatterPlan) else '')
            print(f'N={n} {"homo" if homo else "hetero"} {desc}: {dt*1e6:.0f} us/step', flush=True)
            del conn, ws
            torch.cuda.empty_cache()
        del w
    del idx, spikes
    torch.cuda.empty_cache()
