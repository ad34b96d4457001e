import os, sys, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import brainevent_amd as be
import brainevent_amd._csr as C
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(0)
n, nc = 20000, 100
ptr = torch.arange(n + 1, dtype=torch.int32, device=dev) * nc
idx = torch.randint(0, n, (n * nc,), dtype=torch.int32, device=dev, generator=g)
w = torch.rand(n * nc, device=dev, generator=g)
csr = be.CSR((w, idx, ptr), shape=(n, n), check_structure=False).prepare()
print(type(csr.buffers['scatter_plan']).__name__)
spk = torch.rand(n, device=dev, generator=g) < 0.01
for _ in range(50): be.BinaryArray(spk) @ csr
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000): be.BinaryArray(spk) @ csr
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f'issue {(t1 - t0) / 2000 * 1e6:.1f} us/call')
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): be.BinaryArray(spk) @ csr
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
