"""JITC mm gather (corder=True) against the mv gather at the same shape: one pass over the generated edges should serve
all batch columns."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import brainevent_amd as be
dev = torch.device('cuda', 0)
n = int(os.environ.get('BE_EXP_N', 1000000)); prob = float(os.environ.get('BE_EXP_P', 0.001))
g = torch.Generator(device=dev); g.manual_seed(0)
def t(f, reps=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
v = torch.rand(n, device=dev, generator=g) < 0.01
for B in (1, 4, 8, 32):
    M = torch.rand((n, B), device=dev, generator=g) < 0.01
    for name, call_mv, call_mm in (
        ('scalar', lambda: be.binary_jitsmv(1.0, prob, v, 42, shape=(n, n), transpose=False, corder=True),
                   lambda: be.binary_jitsmm(1.0, prob, M, 42, shape=(n, n), transpose=False, corder=True)),
        ('uniform', lambda: be.binary_jitumv(0.0, 1.0, prob, v, 42, shape=(n, n), transpose=False, corder=True),
                    lambda: be.binary_jitumm(0.0, 1.0, prob, M, 42, shape=(n, n), transpose=False, corder=True)),
    ):
        print(f'n={n} p={prob} B={B} {name}: mv gather {t(call_mv):.2f} ms | mm gather {t(call_mm):.2f} ms', flush=True)
