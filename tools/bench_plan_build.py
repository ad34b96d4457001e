#!/usr/bin/env python3
"""Plan build time (d8 / h8 layouts sort every row in LDS twice) for random and for canonical (ascending) rows."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
from brainevent_amd._csr import ScatterPlan
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(0)
for n, K in ((1_000_000, 1000), (200_000, 10000)):
    idx = torch.randint(0, n, (n, K), dtype=torch.int32, device=dev, generator=g)
    w = torch.rand((n, K), device=dev, generator=g)
    for name, ix in (('random rows', idx), ('ascending rows', torch.sort(idx, dim=1).values.contiguous())):
        for layout in ('d8', 'u16'):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            plan = ScatterPlan.build(w.reshape(-1), ix.reshape(-1), None, shape=(n, n), row_len=K, layout=layout)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print(f'N={n} K={K} {name}, layout {layout}: build {dt*1e3:.0f} ms ({n*K/dt/1e9:.1f} G entries/s)', flush=True)
            del plan
    del idx, w
    torch.cuda.empty_cache()
