import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
dev = torch.device('cuda', 0)
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
n = 32768
W = torch.empty((n, n), dtype=torch.float32, device=dev).normal_()
for nb in (8, 16, 32):
    for p in (0.01, 0.5):
        S = torch.rand((nb, n), device=dev) < p
        tT = timeit(lambda: be.BinaryArray(S) @ W)
        tN = timeit(lambda: W @ be.BinaryArray(S.T.contiguous()))
        union = int(S.any(dim=0).sum())
        print(f'f32 n={n} B={nb} p={p}: S@W {tT*1e3:.3f} ms ({union*n*4/tT/1e9:.0f} GB/s of union rows) | W@S.T {tN*1e3:.3f} ms ({n*n*4/tN/1e9:.0f} GB/s streamed)', flush=True)
