"""Planned step on one post slice of an 8-way cut of C2 (1M x 125k, ~1250 entries per row, 1 % firing): dominant-kernel and
whole-call time against the plan geometry (slices x parts, block layout).  Usage: python tools/exp_shard_geometry.py [--homo]"""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import brainevent_amd as be
from brainevent_amd import _csr as C, _array as A, _lib
from bench import gen_csr_shard_on_device
dev = torch.device('cuda', 0)
homo = '--homo' in sys.argv
n = 1_000_000
w, idx, ptr, shape, _ = gen_csr_shard_on_device(n, n, 10000, homo, 1234, dev, 8, 0)
g = torch.Generator(device=dev); g.manual_seed(1)
spikes = [(torch.rand(n, device=dev, generator=g) < 0.01).to(torch.uint8) for _ in range(10)]
out = torch.empty(shape[1], dtype=torch.float32, device=dev)
pe = _lib.fn('be_profile_enable', ctypes.c_int, [ctypes.c_int]); pr = _lib.fn('be_profile_read', ctypes.c_int, [ctypes.c_void_p, ctypes.c_int])
cfgs = [(None, None, None)] + [(lay, wd, p) for lay in (('h8', 'u16') if homo else ('d8', 'u16'))
                                for wd, p in ((5000, 10), (5000, 5), (2500, 5), (2500, 3), (10000, 10), (10000, 19), (20000 if lay != 'u16' else 16000, 36), (20000 if lay != 'u16' else 16000, 18), (1250, 2), (1250, 1))]
for lay, wd, p in cfgs:
    try:
        plan = C.ScatterPlan.build(w, idx, ptr, shape=shape, layout=lay, slice_width=wd)
    except Exception as e:
        print(lay, wd, p, 'build failed', repr(e)[:80]); continue
    parts = p or plan.default_parts()
    for i in range(5):
        C._plan_call(plan, w, spikes[i], A.BE_SPIKE_BOOL, out, parts=parts)
    torch.cuda.synchronize(); pe(40)
    t0 = time.perf_counter()
    for i in range(40):
        C._plan_call(plan, w, spikes[i % 10], A.BE_SPIKE_BOOL, out, parts=parts)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 40
    ms = (ctypes.c_float * 40)(); nrec = pr(ctypes.cast(ms, ctypes.c_void_p), 40); pe(0)
    print(f"{'homo' if homo else 'hetero'} layout {('u16','d8','h8')[plan.layout]} {plan.n_slices} x {plan.slice_width} x {parts} parts, "
          f"{plan.nnz / plan.m / plan.n_slices:.0f}/block, hint {plan.block_hint}: kernel {np.mean(ms[:nrec])*1e3:.1f} us, call {dt*1e6:.1f} us", flush=True)
    del plan; torch.cuda.empty_cache()
