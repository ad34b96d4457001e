#!/usr/bin/env python3
"""One rank of an 8-way post-slice cut (C2 or C4) on one GPU with a one-rank native exchange: run N steps of ONE schedule, nothing
else on the device — the program to put under `rocprofv3 --hip-trace --kernel-trace` for a timeline, or to time as it is.

  python tools/rank_step_lab.py [--fcn] --schedule seq|ahead|ahead_ids|ahead_auto [--steps N]
"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
from brainevent_amd import _csr as C, _array as A
from brainevent_amd._dist import NativeSpikeExchange, RankStep
from bench import gen_csr_shard_on_device

ap = argparse.ArgumentParser()
ap.add_argument('--fcn', action='store_true'); ap.add_argument('--schedule', default='seq')
ap.add_argument('--steps', type=int, default=300); ap.add_argument('--warmup', type=int, default=50)
ap.add_argument('--world', type=int, default=8); ap.add_argument('--check', action='store_true')
ap.add_argument('--pg', action='store_true', help='initialise a one-rank torch.distributed NCCL group first (as bench.py does)')
a = ap.parse_args()
dev = torch.device('cuda', 0); torch.cuda.set_device(0)
if a.pg:
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29577')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    dist.barrier(); torch.cuda.synchronize()
if a.fcn:
    n_pre, n_post, n_conn = 10_000_000, 10_000_000, 1000
else:
    n_pre, n_post, n_conn = 1_000_000, 1_000_000, 10_000
w, idx, ptr, shape, _ = gen_csr_shard_on_device(n_pre, n_post, n_conn, False, 1234, dev, a.world, 0)
csr = be.CSR((w, idx, ptr), shape=shape, check_structure=False)
if a.fcn:
    csr.prepare()
else:
    csr.buffers['scatter_plan'] = C.ScatterPlan.build(w, idx, ptr, shape=shape)
ex = NativeSpikeExchange(n_pre, 1, 0, NativeSpikeExchange.unique_id(), device=dev)
rs = RankStep(ex, csr)
assert rs._fast is not None
g = torch.Generator(device=dev); g.manual_seed(999)
nb = 20
spikes = torch.rand((nb, n_pre), device=dev, generator=g) < 0.01
events = [A.PackedSpikes(be.bitpack(spikes[b], 0).reshape(-1), n_pre) for b in range(nb)]
torch.cuda.synchronize()
sched = a.schedule


def run(n, first):
    out = None
    if sched == 'seq':
        for i in range(n):
            out = rs(events[(first + i) % nb])
    elif sched in ('ahead', 'ahead_ids', 'ahead_auto'):
        ids = {'ahead': False, 'ahead_ids': True, 'ahead_auto': None}[sched]
        for i in range(n):
            out = rs.ahead(events[(first + i + 1) % nb], ids=ids)
    else:
        for i in range(n):
            out = graphed(events[(first + i + 1) % nb] if 'ahead' in sched else events[(first + i) % nb])
    return out


graphed = None
if sched.startswith('graph'):
    graphed = rs.graphed(schedule=sched[len('graph_'):], static=True)
if 'ahead' in sched and not sched.startswith('graph'):
    rs.post(events[0], ids={'ahead': False, 'ahead_ids': True, 'ahead_auto': None}[sched])
run(a.warmup, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
out = run(a.steps, a.warmup)
torch.cuda.synchronize()
us = (time.perf_counter() - t0) / a.steps * 1e6
if 'ahead' in sched and not sched.startswith('graph'):
    rs.drain()
torch.cuda.synchronize()
msg = f"{'C4' if a.fcn else 'C2'} rank 0 of {a.world}, schedule {sched}: {us:.2f} us/step over {a.steps} steps"
if a.check:
    last = (a.warmup + a.steps - 1) % nb
    ref = be.BinaryArray(spikes[last]) @ csr
    msg += f", last output equals the operator's: {bool(torch.equal(out, ref))}"
print(msg, flush=True)
ex.close()
