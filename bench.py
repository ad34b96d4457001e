#!/usr/bin/env python3
"""bench.py — synaptic updates/s of the event-driven scatter  BinaryArray(spikes) @ CSR  on MI355X.

Headline workload (BASELINE.json configs[1], "C2"): CSR f32, N = 1M pre x 1M post, 1 % connectivity
(10 000 stored synapses per row, heterogeneous weights), Bernoulli 1 % firing, synthetic data generated
on the device with the reference's own generator family (brainevent/_csr/binary.py:771-777:
indptr = arange(n+1) * n_conn, indices ~ U{0..n_post-1}, weights ~ U[0,1)).

One "step" = one  spk @ csr  (spike compaction + LDS-accumulating scatter + partial reduction) on one
fresh spike vector (a batch of 100 pre-generated vectors is cycled, as brainevent/_csr/initialize.py:115-125
does).  Inputs are resident in HBM when the timed region starts.

Multi-GPU (launched by torch.distributed.run, one rank per GPU, RCCL): the matrix is partitioned by
post-neuron slice; every step each rank contributes the spikes of its 1/G of the pre population and one
all-gather rebuilds the full spike vector on every rank (the only exchange of the path); outputs are
disjoint, no reduction.  Default is weak scaling: every rank owns a full C2-sized post slice
(n_pre x n_post_per_gpu), so per-GPU work is fixed; `--scaling strong` splits the 1M posts across ranks.

Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--n', '--neurons', dest='n', type=int, default=1_000_000, help='pre = post population per GPU shard '
                    '(--neurons: the spelling to use under torch.distributed.run, whose own parser trips over --n)')
    ap.add_argument('--conn', type=float, default=0.01)
    ap.add_argument('--fire', type=float, default=0.01)
    ap.add_argument('--homo', action='store_true', help='homogeneous weight (4 B/update) instead of hetero f32')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='weak')
    ap.add_argument('--jit-shard', type=int, default=1, help='jitc workload: run rank 0 of an N-way walk-class partition')
    ap.add_argument('--exchange', choices=['bits', 'bytes'], default='bits', help='payload of the per-step spike all-gather (N > 1)')
    ap.add_argument('--exchange-ahead', type=int, default=0, help='N > 1: 1 = post the all-gather of step t+1 before scattering '
                    'step t (synaptic delay >= 2 steps); default 0 = exchange and scatter strictly in sequence (on one rank the '
                    'pipelined schedule measured 16 us/step slower: the cross-stream waits cost more than the local copy hides)')
    ap.add_argument('--route', choices=['plan', 'direct'], default='plan')
    ap.add_argument('--parts', type=int, default=0)
    ap.add_argument('--shift', type=int, default=0)
    ap.add_argument('--width', type=int, default=0, help='columns per slice of the scatter plan (0 = balanced automatically)')
    ap.add_argument('--layout', choices=['u16', 'd8', 'h8'], default=None, help='block layout of the scatter plan (default: '
                    'the delta layout that applies)')
    ap.add_argument('--workload', choices=['csr', 'jitc', 'fcn', 'dense'], default='csr',
                    help='csr = the headline C2 config; the others are the secondary BASELINE.json configs (single GPU)')
    ap.add_argument('--jit-gather', action='store_true', help='jitc: time the gather orientation instead of the scatter')
    ap.add_argument('--batch', type=int, default=32, help='dense: batch rows')
    ap.add_argument('--k', type=int, default=1000, help='fcn: synapses per pre neuron (stored on this GPU)')
    ap.add_argument('--n-post', type=int, default=0, help='fcn: post population on this GPU (default: n); with --k 125 '
                    '--n-post 1250000 this is one rank of the 8-way post-sliced N=10M, K=1000 config')
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline leg')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    return ap.parse_args()


def gen_csr_on_device(n_pre, n_post, n_conn, homo, seed, dev):
    """Reference generator family, on the device, in row blocks (nnz can exceed 2^31)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    nnz = n_pre * n_conn
    indices = torch.empty(nnz, dtype=torch.int32, device=dev)
    weights = torch.ones(1, dtype=torch.float32, device=dev) if homo else torch.empty(nnz, dtype=torch.float32, device=dev)
    blk = max(1, (1 << 28) // max(n_conn, 1)) * n_conn
    for lo in range(0, nnz, blk):
        hi = min(nnz, lo + blk)
        indices[lo:hi] = torch.randint(0, n_post, (hi - lo,), dtype=torch.int32, device=dev, generator=g)
        if not homo:
            weights[lo:hi].uniform_(0.0, 1.0, generator=g)
    indptr = torch.arange(n_pre + 1, dtype=torch.int64, device=dev) * n_conn
    return weights, indices, indptr


def cpu_baseline(args, n_post, n_conn):
    """The oracle's C restatement of the reference's serial numba scatter loop
    (brainevent/_csr/binary.py:446-451), 1 thread, on a bounded row sample of the same workload:
    same n_post, same row length, same firing rate, fewer pre rows (only active rows are ever touched,
    so the per-update cost is that of the full problem)."""
    from oracle import oracle_c
    oracle_c.build()
    rows = max(1000, min(args.n, int(2.5e8 // max(n_conn, 1))))   # <= 2.5e8 stored synapses (2 GB)
    rng = np.random.default_rng(0)
    indices = rng.integers(0, n_post, rows * n_conn, dtype=np.int32)
    indptr = np.arange(rows + 1, dtype=np.int64) * n_conn
    w = np.ones(1, np.float32) if args.homo else rng.random(rows * n_conn, dtype=np.float32)
    spikes = [(rng.random(rows) < args.fire) for _ in range(8)]
    upd = 0
    t_used = 0.0
    steps = 0
    oracle_c.csrmv_f32(w, indices, indptr, spikes[0], (rows, n_post), True)   # warm
    while t_used < args.cpu_seconds and steps < 10000:
        s = spikes[steps % len(spikes)]
        t0 = time.perf_counter()
        oracle_c.csrmv_f32(w, indices, indptr, s, (rows, n_post), True)
        t_used += time.perf_counter() - t0
        upd += int(s.sum()) * n_conn
        steps += 1
    res = {
        'value': upd / t_used / 1e9, 'unit': 'Geff/s', 'cores': 1, 'kind': 'port',
        'sample': f'{rows} of {args.n} pre rows (n_post={n_post}, {n_conn} synapses/row, fire={args.fire}), '
                  f'{steps} steps, C port of the reference numba loop _csr/binary.py:446-451, gcc -O3 -march=native',
        'host_cpus': os.cpu_count(),
    }
    # the all-cores variant next to it (OpenMP over the active rows + atomic adds).  NOT the reference's
    # algorithm (its scatter is serial by construction) — and on this host slower than the serial loop: float atomics on a
    # shared 4 MB vector bounce cache lines between cores.  A few seconds only.
    try:
        avail = len(os.sched_getaffinity(0))
        best = None
        for n_thr in sorted({min(avail, t) for t in (4, 16, 64, avail)}):     # the best thread count is the bound
            oracle_c.csrmv_t_f32_parallel(w, indices, indptr, spikes[0], (rows, n_post), n_thr)
            upd_p, t_p, steps_p = 0, 0.0, 0
            while t_p < 1.0 and steps_p < 10000:
                s = spikes[steps_p % len(spikes)]
                t0 = time.perf_counter()
                oracle_c.csrmv_t_f32_parallel(w, indices, indptr, s, (rows, n_post), n_thr)
                t_p += time.perf_counter() - t0
                upd_p += int(s.sum()) * n_conn
                steps_p += 1
            if best is None or upd_p / t_p > best[0]:
                best = (upd_p / t_p, n_thr)
        res['parallel_atomics_variant'] = {'value': best[0] / 1e9, 'unit': 'Geff/s', 'cores': best[1], 'cores_available': avail,
                                        'what': 'same loop, OpenMP over active rows + atomic adds, best of 4 / 16 / 64 / all '
                                                'threads (not the reference algorithm)'}
    except Exception as e:
        res['parallel_atomics_variant'] = {'error': repr(e)}
    return res


def time_steps(step, steps, warmup):
    import ctypes as ct
    from brainevent_amd import _lib
    for i in range(warmup):
        out = step(i)
    prof_enable = _lib.fn('be_profile_enable', ct.c_int, [ct.c_int])
    prof_read = _lib.fn('be_profile_read', ct.c_int, [ct.c_void_p, ct.c_int])
    _lib.check(prof_enable(steps), 'be_profile_enable')
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        out = step(warmup + i)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms = (ct.c_float * steps)()
    n_rec = prof_read(ct.cast(ms, ct.c_void_p), steps)
    prof_enable(0)
    return elapsed, (float(np.mean(ms[:n_rec])) if n_rec > 0 else None), out


def secondary(args):
    """Secondary single-GPU workloads (BASELINE.json configs[2..4]); same JSON shape, own metric strings."""
    import brainevent_amd as be
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    n_batch = 20
    cfg, roof = {}, None
    if args.workload == 'jitc':
        n = args.n if args.n != 1_000_000 else 4_000_000
        prob = args.conn if args.conn != 0.01 else 0.001
        M = be.JITCScalarR((np.float32(1.0), prob, 42), shape=(n, n), corder=not args.jit_gather)
        spikes = torch.rand((n_batch, n), device=dev, generator=g) < args.fire
        target = M.scatter_shard(args.jit_shard, 0) if args.jit_shard > 1 else M     # rank 0's walk classes of an N-way split
        step = lambda i: be.BinaryArray(spikes[i % n_batch]) @ target
        elapsed, kern_ms, out = time_steps(step, args.steps, args.warmup)
        upd = float(out.double().sum().item())          # weight 1: the sum is the number of delivered edges
        value = upd * args.steps / elapsed / 1e9 if not args.jit_gather else n * n * prob * args.steps / elapsed / 1e9
        metric = 'synaptic updates/sec (Geff/s), BinaryArray @ JITCScalarR ' + ('gather: generated edges/s' if args.jit_gather else 'scatter')
        cfg = {'workload': f'BinaryArray({args.fire:g}) @ JITCScalarR w=1 prob={prob:g} seed=42 {n}x{n}, '
                           f"{'gather (corder=False matrix)' if args.jit_gather else 'scatter (corder=True matrix)'}",
               'edges_last_step': upd}
        if args.jit_shard > 1:
            cfg['shard'] = f'walk classes of rank 0 of {args.jit_shard} (no stored state; outputs of the ranks are disjoint)'
        if kern_ms:
            # no stored matrix: HBM is not the bound; report the bandwidth a stored CSR would have needed (8 B/update)
            roof = {'bound': 'valu+lds (no HBM matrix traffic)', 'achieved': None, 'peak': None, 'unit': 'GB/s', 'frac': None,
                    'traffic': None, 'kernel_ms': round(kern_ms, 5),
                    'equivalent_stored_matrix_GBps': round(8 * upd / (kern_ms * 1e-3) / 1e9, 1)}
    elif args.workload == 'fcn':
        n = args.n if args.n != 1_000_000 else 10_000_000
        K = args.k
        n_post = args.n_post or n
        idx = torch.empty((n, K), dtype=torch.int32, device=dev)
        for lo in range(0, n, 200_000):
            hi = min(n, lo + 200_000)
            idx[lo:hi] = torch.randint(0, n_post, (hi - lo, K), dtype=torch.int32, device=dev, generator=g)
        w = torch.ones(1, device=dev) if args.homo else torch.empty((n, K), device=dev).uniform_(0, 1, generator=g)
        conn = be.FixedNumPerPre((w, idx), shape=(n, n_post), check_indices=False)
        conn.prepare()
        spikes = torch.rand((n_batch, n), device=dev, generator=g) < args.fire
        act = spikes.sum(dim=1).cpu().numpy()
        step = lambda i: be.BinaryArray(spikes[i % n_batch]) @ conn
        elapsed, kern_ms, out = time_steps(step, args.steps, args.warmup)
        upd = sum(int(act[(args.warmup + i) % n_batch]) for i in range(args.steps)) * K
        value = upd / elapsed / 1e9
        metric = 'synaptic updates/sec (Geff/s), BinaryArray @ FixedNumPerPre scatter'
        cfg = {'workload': f"BinaryArray({args.fire:g}) @ FixedNumPerPre K={K} {n} pre x {n_post} post "
                           f"{'homo' if args.homo else 'hetero'} f32, 1 GPU",
               'route': type(conn.buffers.get('scatter_plan')).__name__ if conn.buffers.get('scatter_plan') is not None
               else 'direct (global atomics)'}
        if kern_ms:
            alg = (4 if args.homo else 8) * float(np.mean(act)) * K + n + 4 * n_post
            roof = {'bound': 'hbm', 'achieved': round(alg / (kern_ms * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(alg / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), 'traffic': None, 'kernel_ms': round(kern_ms, 5)}
    else:   # dense
        n = args.n if args.n != 1_000_000 else 65536
        W = torch.empty((n, n), dtype=torch.float16, device=dev).normal_(0, 1, generator=g)
        spikes = torch.rand((n_batch, args.batch, n), device=dev, generator=g) < args.fire
        step = lambda i: be.BinaryArray(spikes[i % n_batch]) @ W
        elapsed, kern_ms, out = time_steps(step, args.steps, args.warmup)
        pairs = float(spikes.sum().item()) / n_batch
        value = pairs * n * args.steps / elapsed / 1e9
        metric = 'synaptic updates/sec (Geff/s), batched BinaryArray @ dense fp16'
        union = float(spikes.any(dim=1).sum().item()) / n_batch
        cfg = {'workload': f'BinaryArray({args.fire:g}) [{args.batch},{n}] @ dense fp16 [{n},{n}]', 'union_rows': union,
               'active_pairs': pairs}
        if kern_ms:
            alg = union * n * 2 + args.batch * n * 2
            roof = {'bound': 'hbm', 'achieved': round(alg / (kern_ms * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(alg / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), 'traffic': None, 'kernel_ms': round(kern_ms, 5),
                    'algorithmic_bytes_per_launch': int(alg)}
    line = {'metric': metric, 'value': round(value, 3), 'unit': 'Geff/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 5), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f16' if args.workload == 'dense' else 'f32', 'data': 'synthetic', 'config': cfg, 'roofline': roof}
    print(json.dumps(line), flush=True)


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    multi = world > 1 or os.environ.get('BENCH_FORCE_DIST') == '1'
    is_fcn = args.workload == 'fcn'
    if args.workload != 'csr' and not (is_fcn and multi):
        return secondary(args)       # jitc / dense, and fcn on one GPU without the exchange
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world and world > 1:
        raise SystemExit(f'--gpus {args.gpus} != WORLD_SIZE {world}')
    if args.gpus > 1 and world == 1:
        raise SystemExit('launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N')
    n_dev = max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank % n_dev)          # (rehearsals put several ranks on one card; the driver has one GPU per rank)
    dev = torch.device('cuda', local_rank % n_dev)
    dist = None
    # BENCH_FORCE_DIST=1 runs the multi-rank code path (process group, all-gather, max-reduce) with a single rank:
    # the only way to exercise it on a one-GPU box
    use_dist = world > 1 or os.environ.get('BENCH_FORCE_DIST') == '1'
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        backend = os.environ.get('BENCH_BACKEND', 'nccl')      # nccl = RCCL; 'gloo' only to rehearse several ranks on one card
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    import brainevent_amd as be
    from brainevent_amd import _csr as C, _lib

    plan, plan_bytes = None, 0
    t_setup = time.perf_counter()
    if is_fcn:
        # C4: FixedNumPerPre K = 1000, N = 10M, post-sliced (strong scaling: the problem is fixed, a rank holds the K / world
        # synapses per row that land in its N / world outputs).  The shard of a uniform random matrix is generated directly:
        # K / world targets per row, uniform over the rank's slice (the exact shard is ragged around that, §4 of DESIGN.md).
        n_pre = args.n if args.n != 1_000_000 else 10_000_000
        n_post_total = args.n_post or n_pre
        n_post = n_post_total // world
        n_conn = max(1, args.k // world)
        args.scaling = 'strong'
        g0 = torch.Generator(device=dev)
        g0.manual_seed(4321 + rank)
        idx = torch.empty((n_pre, n_conn), dtype=torch.int32, device=dev)
        for lo in range(0, n_pre, 200_000):
            hi = min(n_pre, lo + 200_000)
            idx[lo:hi] = torch.randint(0, n_post, (hi - lo, n_conn), dtype=torch.int32, device=dev, generator=g0)
        wts = torch.ones(1, device=dev) if args.homo else torch.empty((n_pre, n_conn), device=dev).uniform_(0, 1, generator=g0)
        csr = be.FixedNumPerPre((wts, idx), shape=(n_pre, n_post), check_indices=False).prepare()
        ws_obj = csr.buffers.get('scatter_plan')
        args.route = type(ws_obj).__name__ if ws_obj is not None else 'direct'
        if isinstance(ws_obj, C.ScatterPlan):
            plan, plan_bytes = ws_obj, ws_obj.nbytes()
    else:
        n_pre = args.n
        n_post_total = args.n * world if args.scaling == 'weak' else args.n
        n_post = n_post_total // world                     # this rank's post slice
        n_conn = max(1, int(n_post * args.conn))           # stored synapses per (row, shard)
        weights, indices, indptr = gen_csr_on_device(n_pre, n_post, n_conn, args.homo, 1234 + rank, dev)
        csr = be.CSR((weights, indices, indptr), shape=(n_pre, n_post), check_structure=False)
    if not is_fcn and args.route == 'plan':
        # default: LDS-filling accumulator capacity and slices balanced over the 256 CUs; --shift forces full-capacity slices
        csr.buffers['scatter_plan'] = C.ScatterPlan.build(weights, indices, indptr, shape=(n_pre, n_post),
                                                          slice_shift=args.shift or None, slice_width=args.width or None,
                                                          layout=args.layout or None)
        plan = csr.buffers['scatter_plan']
        plan_bytes = plan.nbytes()
        if args.parts:
            plan.default_parts = lambda: args.parts
    elif not is_fcn:                      # --route direct: global atomics
        csr.buffers['scatter_plan'] = None
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup

    # spike batch: each rank draws the spikes of its own 1/world of the pre population
    n_batch = 100
    g = torch.Generator(device=dev)
    g.manual_seed(999 + rank)
    n_local = n_pre // world
    if use_dist:
        from brainevent_amd._dist import SpikeExchange
        # one all-gather per step (RCCL over xGMI); bit-packed by default: 1/8 of the bytes, consumed packed
        exchange = SpikeExchange(n_pre, packed=(args.exchange == 'bits'), device=dev)
        n_local = exchange.hi - exchange.lo
    local_spikes = (torch.rand((n_batch, n_local), device=dev, generator=g) < args.fire)
    if use_dist:
        counts = torch.empty(n_batch, dtype=torch.int64, device=dev)
        for b in range(n_batch):
            counts[b] = exchange.gather(local_spikes[b]).sum()
        active_per_vec = counts.cpu().numpy()
    else:
        active_per_vec = local_spikes.sum(dim=1).cpu().numpy()

    ahead = use_dist and args.exchange == 'bits' and args.exchange_ahead
    ticket = [exchange.post(local_spikes[0])] if ahead else None

    def step(i):
        s = local_spikes[i % n_batch]
        if ahead:
            # step i's spikes were posted during step i - 1: post step i + 1's now (the collective overlaps with the
            # scatter below), then consume step i's.  Every timed step still issues one exchange and one scatter.
            nxt = exchange.post(local_spikes[(i + 1) % n_batch])
            ev = exchange.wait_events(ticket[0])
            ticket[0] = nxt
            return ev @ csr
        if use_dist:
            return exchange.gather_events(s) @ csr
        return be.BinaryArray(s) @ csr

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        out = step(i)
    prof_enable = _lib.fn('be_profile_enable', ctypes.c_int, [ctypes.c_int])
    prof_read = _lib.fn('be_profile_read', ctypes.c_int, [ctypes.c_void_p, ctypes.c_int])
    _lib.check(prof_enable(args.steps), 'be_profile_enable')
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    if ahead:
        ticket[0][1].wait()      # the exchange posted by the last step (never consumed)
    ms = (ctypes.c_float * args.steps)()
    n_rec = prof_read(ctypes.cast(ms, ctypes.c_void_p), args.steps)
    prof_enable(0)
    kern_ms = float(np.mean(ms[:n_rec])) if n_rec > 0 else None
    kern_ms_median = float(np.median(ms[:n_rec])) if n_rec > 0 else None

    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # exact work done in the timed steps (all ranks see the same full spike vector)
    upd_per_rank = sum(int(active_per_vec[(args.warmup + i) % n_batch]) for i in range(args.steps)) * n_conn
    total_upd = upd_per_rank * world
    value = total_upd / elapsed / 1e9
    checksum = float(out.double().sum().item())

    copy_gbps = None
    if rank == 0:
        try:
            src = torch.empty(1 << 29, dtype=torch.float32, device=dev)       # 2 GiB
            dst = torch.empty_like(src)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            dst.copy_(src)
            e0.record()
            for _ in range(5):
                dst.copy_(src)
            e1.record()
            torch.cuda.synchronize()
            copy_gbps = round(5 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
            del src, dst
        except Exception:
            copy_gbps = None
    if rank == 0:
        bytes_per_upd = 4 if args.homo else 8           # int32 index (+ f32 weight)
        mean_active = float(np.mean([active_per_vec[(args.warmup + i) % n_batch] for i in range(args.steps)]))
        alg_bytes = bytes_per_upd * mean_active * n_conn + n_pre * 1 + n_post * 4 + 16 * mean_active
        roof = None
        if kern_ms:
            achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
            traffic = None
            tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    # PMC figures exist for the default workload only (density 1 %, firing 1 %), per block layout
                    lay = {1: 'd8', 2: 'h8'}.get(getattr(plan, 'layout', 0) if args.route == 'plan' else 0, 'u16')
                    key = f"{'homo' if args.homo else 'hetero'}_{lay}_n{args.n}"
                    if args.conn == 0.01 and args.fire == 0.01 and args.route == 'plan' and world == 1:
                        traffic = tj.get(key, {}).get('hbm_bytes_per_launch')
                except Exception:
                    traffic = None
            plan_kernel = {1: 'k_plan_accumulate_d8', 2: 'k_plan_accumulate_h8'}.get(getattr(plan, 'layout', 0), 'k_plan_accumulate')
            kernel_name = {'plan': plan_kernel, 'ScatterPlan': plan_kernel,
                           'BinnedScatter': 'k_bin_rows'}.get(args.route, 'k_csrmv_t_direct')
            roof = {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic,
                    # the same run's device-copy ceiling (bytes read + written per second of a 2 GiB
                    # device-to-device copy) next to the nominal peak, and the kernel's real traffic rate against it
                    'device_copy_GBps': copy_gbps,
                    'traffic_GBps': (round(traffic / (kern_ms * 1e-3) / 1e9, 1) if traffic else None),
                    'kernel': kernel_name,
                    'kernel_ms': round(kern_ms, 5), 'kernel_ms_median': round(kern_ms_median, 5),
                    'algorithmic_bytes_per_launch': int(alg_bytes)}
        line = {
            'metric': 'synaptic updates/sec (Geff/s), BinaryArray @ ' + ('FixedNumPerPre scatter' if is_fcn else 'CSR scatter'),
            'value': round(value, 3), 'unit': 'Geff/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 5), 'higher_is_better': True, 'scaling': args.scaling,
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': (f"BinaryArray({args.fire:g} fire) @ FixedNumPerPre K={args.k} f32 "
                                    f"{'homo' if args.homo else 'hetero'}, {n_pre} pre x {n_post_total} post "
                                    f"({n_conn} synapses/row/shard), route={args.route}" if is_fcn else
                                    f"BinaryArray({args.fire:g} fire) @ CSR f32 {'homo' if args.homo else 'hetero'}, "
                                    f"{n_pre} pre x {n_post_total} post, {args.conn:g} density "
                                    f"({n_conn} synapses/row/shard), route={args.route}"),
                       'n_pre': n_pre, 'n_post': n_post_total, 'n_post_per_gpu': n_post, 'n_conn': n_conn,
                       'parallelism': f'post-slice x{world}' + (f' + spike all-gather ({args.exchange}' + (', posted one step ahead' if ahead else '') + ')' if use_dist else ''),
                       'plan_GB': round(plan_bytes / 1e9, 2), 'setup_s': round(t_setup, 2),
                       'plan_slices': (f'{plan.n_slices} x {plan.slice_width} columns x {plan.default_parts()} parts, '
                                       f"layout {({1: 'd8 (5 B/entry)', 2: 'h8 (1 B/entry)'}.get(plan.layout, 'u16'))}" if plan is not None else None),
                       'mean_active_rows': mean_active, 'checksum': checksum},
            'roofline': roof,
        }
        if world == 1 and not args.no_cpu:
            try:
                line['cpu_baseline'] = cpu_baseline(args, n_post, n_conn)
            except Exception as e:   # the CPU leg must never sink the GPU number
                line['cpu_baseline'] = {'error': repr(e)}
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
