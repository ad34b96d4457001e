#!/usr/bin/env python3
"""Plan build and weight-refresh time for random and for canonical (ascending) rows.  The sorted layouts (d8 / h8) need every
row in column order: with the order stored by the count pass (keep_order) the fill reads it back and a refresh is a
gather-copy; without it (keep_order=False, the round-1/2 behaviour) the rows are sorted in LDS by the count pass, the fill
and every refresh.  BE_BUILD_CASES="n:K,n:K" picks the sizes (default 1M x 1000, 200k x 10000; C2 = 1000000:10000)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
from brainevent_amd._csr import ScatterPlan
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(0)
cases = [tuple(int(x) for x in c.split(':')) for c in os.environ.get('BE_BUILD_CASES', '1000000:1000,200000:10000').split(',')]


def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = f()
    torch.cuda.synchronize()
    return r, time.perf_counter() - t0


for n, K in cases:
    idx = torch.empty((n, K), dtype=torch.int32, device=dev)
    for lo in range(0, n, 100_000):
        idx[lo:lo + 100_000] = torch.randint(0, n, (min(100_000, n - lo), K), dtype=torch.int32, device=dev, generator=g)
    w = torch.empty((n, K), device=dev).uniform_(0, 1, generator=g)
    kinds = [('random rows', idx)]
    if n * K <= 2_000_000_000:
        kinds.append(('ascending rows', torch.sort(idx, dim=1).values.contiguous()))
    for name, ix in kinds:
        for layout, ko in (('d8', True), ('d8', False), ('u16', None)):
            plan, dt = timed(lambda: ScatterPlan.build(w.reshape(-1), ix.reshape(-1), None, shape=(n, n), row_len=K, layout=layout,
                                                       keep_order=ko))
            w.mul_(0.99)
            _, dr = timed(lambda: plan.refresh_weights(w.reshape(-1), ix.reshape(-1), None))
            tag = layout + ('' if ko is None else (' + stored order' if ko else ', sorting every time'))
            print(f'N={n} K={K} {name}, {tag}: build {dt*1e3:.0f} ms ({n*K/dt/1e9:.1f} G entries/s), '
                  f'weight refresh {dr*1e3:.0f} ms, plan {plan.nbytes()/1e9:.2f} GB', flush=True)
            del plan
            torch.cuda.empty_cache()
    del idx, w, kinds
    torch.cuda.empty_cache()
