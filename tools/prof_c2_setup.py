#!/usr/bin/env python3
"""Where the setup of the C2 plan goes (1M x 1M, 1e10 entries, d8 layout): matrix generation, plan build, weight refresh,
fixed-point exponent — wall clock per stage (rocprofv3 --kernel-trace --stats around this script gives the kernels)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
from brainevent_amd import _csr as C
from bench import gen_csr_on_device
dev = torch.device('cuda', 0)
n = int(os.environ.get('N', 1_000_000)); nc = int(os.environ.get('NC', 10_000))
def tick(msg, t0):
    torch.cuda.synchronize(); t = time.perf_counter(); print(f'{msg}: {(t - t0) * 1e3:.0f} ms', flush=True); return t
t = time.perf_counter()
w, idx, ptr = gen_csr_on_device(n, n, nc, False, 1234, dev); t = tick('generate matrix', t)
plan = C.ScatterPlan.build(w, idx, ptr, shape=(n, n)); t = tick('ScatterPlan.build (count + fill + exponent)', t)
e = C.fixed_point_exponent(w, idx, n); t = tick(f"be_fixed_point_exponent (global atomics over the entries; the plan no longer calls it: e = {e}, plan e = {plan.scale_exp})", t)
w.mul_(0.5); torch.cuda.synchronize(); t = time.perf_counter()
plan.refresh_weights(w, idx, ptr); t = tick('refresh_weights after an in-place update (order not kept: the rows are sorted again)', t)
del plan; torch.cuda.synchronize(); t = time.perf_counter()
plan = C.ScatterPlan.build(w, idx, ptr, shape=(n, n), keep_order=True); t = tick('ScatterPlan.build(keep_order=True)', t)
w.mul_(0.5); torch.cuda.synchronize(); t = time.perf_counter()
plan.refresh_weights(w, idx, ptr); t = tick('refresh_weights through the kept order (gather-copy + column statistics from the plan)', t)
