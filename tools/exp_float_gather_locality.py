"""Is `csr @ x` bound by the random gather of x?  The same matrix stream (1M rows x 1000 entries) against operands of 1e3 ... 4e6 columns."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
from bench import gen_csr_on_device
dev = torch.device('cuda', 0)
n, row = 1_000_000, 1000
for k in (1000, 16_000, 100_000, 1_000_000, 4_000_000):
    w, idx, ptr = gen_csr_on_device(n, k, row, False, 0, dev)
    x = torch.randn(k, device=dev)
    for _ in range(2):
        be.csrmv(w, idx, ptr, x, shape=(n, k))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        be.csrmv(w, idx, ptr, x, shape=(n, k))
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5
    print(f'k={k:8d}: {t * 1e3:7.3f} ms = {n * row * 8 / t / 1e9:6.0f} GB/s of matrix stream', flush=True)
    del w, idx, ptr, x
    torch.cuda.empty_cache()
