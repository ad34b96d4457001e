"""Planned layout vs direct route (global atomics) on small and mid-size matrices: the measurement behind PLAN_MIN_NNZ."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import brainevent_amd as be
import brainevent_amd._csr as C
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(0)
for n, nc in ((4000, 80), (20000, 100), (50000, 100), (100000, 100), (100000, 1000)):
    ptr = torch.arange(n + 1, dtype=torch.int32, device=dev) * nc
    idx = torch.randint(0, n, (n * nc,), dtype=torch.int32, device=dev, generator=g)
    for homo in (True, False):
        w = torch.ones(1, device=dev) if homo else torch.rand(n * nc, device=dev, generator=g)
        for fire in (0.01, 0.1):
            spikes = [torch.rand(n, device=dev, generator=g) < fire for _ in range(10)]
            res = []
            for route in ('direct', 'plan'):
                csr = be.CSR((w, idx, ptr), shape=(n, n), check_structure=False)
                csr.buffers['scatter_plan'] = C.ScatterPlan.build(w, idx, ptr, shape=(n, n)) if route == 'plan' else None
                for i in range(5):
                    out = be.BinaryArray(spikes[i]) @ csr
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(200):
                    out = be.BinaryArray(spikes[i % 10]) @ csr
                torch.cuda.synchronize()
                res.append((time.perf_counter() - t0) / 200 * 1e6)
            print(f'n={n} nnz={n*nc:.1e} {"homo" if homo else "hetero"} fire={fire}: direct {res[0]:.0f} us, plan {res[1]:.0f} us', flush=True)
