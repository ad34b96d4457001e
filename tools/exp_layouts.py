"""Block layouts of the scatter plan against each other as a function of the entries per (row, slice) block
(FixedNumPerPre K = 1000, 1 % firing): u16 vs the sorted-delta layouts (d8 hetero / h8 homo), each at its own balanced
geometry.  The measurement behind ScatterPlan.build's automatic layout choice (DELTA_MIN_BLOCK)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import brainevent_amd as be
import brainevent_amd._csr as C
from brainevent_amd import _array as A
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(0)
K = int(os.environ.get('BE_EXP_K', 1000))
PARTS = int(os.environ['BE_EXP_PARTS']) if os.environ.get('BE_EXP_PARTS') else None
for n in [int(x) for x in os.environ.get('BE_EXP_NS', '100000,200000,350000,500000,1000000,1500000,2500000').split(',')]:
    idx = torch.randint(0, n, (n, K), dtype=torch.int32, device=dev, generator=g)
    spikes = [(torch.rand(n, device=dev, generator=g) < 0.01).to(torch.uint8) for _ in range(10)]
    out = torch.empty(n, dtype=torch.float32, device=dev)
    for homo in ((True,) if os.environ.get('BE_EXP_HOMO_ONLY') else (False,) if os.environ.get('BE_EXP_HETERO_ONLY') else (True, False)):
        w = torch.ones(1, device=dev) if homo else torch.empty((n, K), device=dev).uniform_(0, 1, generator=g)
        res = []
        for layout in [l for l in ('u16', 'h8' if homo else 'd8', None) if str(l) in os.environ.get('BE_EXP_LAYOUTS', 'u16,d8,h8,None').split(',')]:
            width = 0 if layout is None else int(os.environ.get('BE_EXP_WIDTH16' if layout == 'u16' else 'BE_EXP_WIDTH', 0))
            plan = C.ScatterPlan.build(w, idx.reshape(-1), None, shape=(n, n), row_len=K, layout=layout, slice_width=width or None)
            if os.environ.get('BE_EXP_HINT'):
                plan.block_hint_override = int(os.environ['BE_EXP_HINT'])
            for i in range(5):
                C._plan_call(plan, w, spikes[i], A.BE_SPIKE_BOOL, out, parts=PARTS)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(40):
                C._plan_call(plan, w, spikes[i % 10], A.BE_SPIKE_BOOL, out, parts=PARTS)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 40
            res.append(f'{layout or "auto=" + ("u16", "d8", "h8")[plan.layout]} {plan.n_slices}x{PARTS or plan.default_parts()} ({K / plan.n_slices:.0f}/block) {dt*1e6:.0f} us')
            del plan
            torch.cuda.empty_cache()
        print(f'N={n} K={K} {"homo" if homo else "hetero"}: ' + ' | '.join(res), flush=True)
        del w
    del idx, spikes
    torch.cuda.empty_cache()
