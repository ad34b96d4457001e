#!/usr/bin/env python3
"""bench.py — synaptic updates/s of the event-driven scatter  BinaryArray(spikes) @ CSR  on MI355X.

Headline workload (BASELINE.json configs[1], "C2"): CSR f32, N = 1M pre x 1M post, 1 % connectivity
(10 000 stored synapses per row, heterogeneous weights), Bernoulli 1 % firing, synthetic data generated
on the device with the reference's own generator family (brainevent/_csr/binary.py:771-777:
indptr = arange(n+1) * n_conn, indices ~ U{0..n_post-1}, weights ~ U[0,1)).

One "step" = one  spk @ csr  (spike compaction + LDS-accumulating scatter + partial reduction) on one
fresh spike vector (a batch of 100 pre-generated vectors is cycled, as brainevent/_csr/initialize.py:115-125
does).  Inputs are resident in HBM when the timed region starts.

Multi-GPU (launched by torch.distributed.run, one rank per GPU, RCCL): ONE global matrix — the named N = 1M problem,
the same seed on every rank — is partitioned by post-neuron slice: rank g keeps the synapses that land in its 1/G of
the outputs (generated in row blocks and cut on the fly, so no rank ever holds more than its shard).  Every step each
rank contributes the spikes of its 1/G of the pre population and one all-gather rebuilds the full spike vector on every
rank (the only exchange of the path); outputs are disjoint, no reduction.  That is STRONG scaling (`"scaling": "strong"`,
the default: total work is fixed as N grows).  `--scaling weak` keeps a full C2-sized post slice per rank instead
(1M pre x N·1M post: a different, N times larger problem — per-GPU work fixed).

Prints ONE JSON line (rank 0).  With the default arguments on one GPU the line also carries `secondary`: the other
single-GPU BASELINE.json configs (C3 JITC scatter, C4 FixedNumPerPre N = 10M K = 1000, C5 batched dense fp16), each with
its own value / ms_per_step / kernel_ms / roofline, timed after the headline.
"""
import argparse
import math
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

_STATE = {}              # process-wide notes of this run (exchange init timed out, which extra leg is running)
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
TRAFFIC_SOURCE = 'profiles/traffic.json (rocprofv3 --pmc, separate pass, guide corrections applied; not this run)'


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--n', '--neurons', dest='n', type=int, default=1_000_000, help='pre = post population '
                    '(--neurons: the spelling to use under torch.distributed.run, whose own parser trips over --n)')
    ap.add_argument('--conn', type=float, default=0.01)
    ap.add_argument('--fire', type=float, default=0.01)
    ap.add_argument('--exact-active', action='store_true', help='csr: every spike vector has exactly round(n * fire) active neurons '
                    '(the reference tuner\'s generator, brainevent/_csr/initialize.py:115-125) instead of Bernoulli(fire)')
    ap.add_argument('--homo', action='store_true', help='homogeneous weight (4 B/update) instead of hetero f32')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='strong',
                    help='N > 1: strong = the named problem cut into N post slices (default); weak = N times the problem')
    ap.add_argument('--emulate-world', type=int, default=0, help='one process: hold the shard rank 0 of a W-way strong split '
                    'would hold and run the exchange path with a one-rank group (what one rank of W does per step)')
    ap.add_argument('--jit-shard', type=int, default=1, help='jitc workload: run rank 0 of an N-way walk-class partition')
    ap.add_argument('--exchange', choices=['bits', 'bytes'], default='bits', help='payload of the per-step spike all-gather (N > 1)')
    ap.add_argument('--producer', choices=['words', 'bytes'], default='words', help='N > 1, --exchange bits: the form a rank\'s own '
                    'spikes arrive in.  words = bit-packed by the producer (what be_lif_coba_step_packed writes): the exchange '
                    'gathers them from where they lie; bytes = one byte per neuron, packed by the exchange every step')
    ap.add_argument('--exchange-impl', choices=['native', 'torch'], default='native', help='native = the library\'s own RCCL '
                    'all-gather (be_exchange_*: one C call per step); torch = torch.distributed.all_gather_into_tensor '
                    '(about 30 us of host time per call: a 1-of-8 shard step is host-bound behind it)')
    ap.add_argument('--exchange-ahead', type=int, default=0, help='N > 1: 1 = post the all-gather of step t+1 before scattering '
                    'step t (synaptic delay >= 2 steps); default 0 = exchange and scatter strictly in sequence')
    ap.add_argument('--route', choices=['plan', 'direct', 'auto'], default='plan')
    ap.add_argument('--parts', type=int, default=0)
    ap.add_argument('--shift', type=int, default=0)
    ap.add_argument('--width', type=int, default=0, help='columns per slice of the scatter plan (0 = balanced automatically)')
    ap.add_argument('--layout', choices=['u16', 'd8', 'h8'], default=None, help='block layout of the scatter plan (default: '
                    'the delta layout that applies)')
    ap.add_argument('--workload', choices=['csr', 'jitc', 'fcn', 'dense', 'gather_mirror'], default='csr',
                    help='csr = the headline C2 config; the others are the secondary BASELINE.json configs')
    ap.add_argument('--jit-gather', action='store_true', help='jitc: time the gather orientation instead of the scatter')
    ap.add_argument('--batch', type=int, default=32, help='dense: batch rows')
    ap.add_argument('--k', type=int, default=1000, help='fcn: synapses per pre neuron')
    ap.add_argument('--acc32', action='store_true', help='fcn: force the 32-bit fixed-point sums of the binned route (opt-in: outputs good '
                    'to rtol = atol = 1e-5, not to 1e-5 relative per output — never the default for weights like U[0,1))')
    ap.add_argument('--n-post', type=int, default=0, help='fcn: post population (default: n)')
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline leg')
    ap.add_argument('--no-secondary', action='store_true', help='skip the secondary configs after the headline')
    ap.add_argument('--secondary-steps', type=int, default=60)
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--full-line', action='store_true', help='print the verbose line (every prose field per entry) instead of the compact '
                    'one; the verbose line is always also written to --full-line-file')
    ap.add_argument('--full-line-file', default=os.path.join('gpurun_out', 'bench_full_line.json'),
                    help="where rank 0 writes the verbose line ('' = nowhere)")
    return ap.parse_args(argv)


# =====================================================================================================================
# synthetic inputs
# =====================================================================================================================
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


def gen_csr_shard_on_device(n_pre, n_post_total, n_conn, homo, seed, dev, world, rank):
    """Post slice `rank` of `world` of the SAME global matrix :func:`gen_csr_on_device` draws (same seed, same generator
    calls in the same order on every rank): each row block is generated, cut to the columns [lo, hi) with local ids, and
    appended — a rank never holds more than its shard plus one block.  Rows of the shard are ragged
    (Binomial(n_conn, 1/world) entries).  Returns (weights, indices, indptr int64, (n_pre, hi - lo), global nnz)."""
    from brainevent_amd._dist import post_slice_bounds
    lo, hi = post_slice_bounds(n_post_total, world, rank)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    rows_blk = max(1, (1 << 28) // max(n_conn, 1))
    idx_parts, w_parts, cnt_parts = [], [], []
    for r0 in range(0, n_pre, rows_blk):
        r1 = min(n_pre, r0 + rows_blk)
        blk = torch.randint(0, n_post_total, ((r1 - r0) * n_conn,), dtype=torch.int32, device=dev, generator=g)
        wb = None if homo else torch.empty((r1 - r0) * n_conn, dtype=torch.float32, device=dev).uniform_(0.0, 1.0, generator=g)
        keep = (blk >= lo) & (blk < hi)
        cnt_parts.append(keep.view(r1 - r0, n_conn).sum(dim=1))
        idx_parts.append((blk[keep] - lo).to(torch.int32))
        if not homo:
            w_parts.append(wb[keep])
        del blk, wb, keep
    indices = torch.cat(idx_parts)
    weights = torch.ones(1, dtype=torch.float32, device=dev) if homo else torch.cat(w_parts)
    del idx_parts, w_parts
    indptr = torch.zeros(n_pre + 1, dtype=torch.int64, device=dev)
    torch.cumsum(torch.cat(cnt_parts), 0, out=indptr[1:])
    return weights, indices, indptr, (n_pre, hi - lo), n_pre * n_conn


def gen_fixed_num_on_device(n, K, n_post, homo, dev, g):
    idx = torch.empty((n, K), dtype=torch.int32, device=dev)
    for lo in range(0, n, 200_000):
        hi = min(n, lo + 200_000)
        idx[lo:hi] = torch.randint(0, n_post, (hi - lo, K), dtype=torch.int32, device=dev, generator=g)
    w = torch.ones(1, device=dev) if homo else torch.empty((n, K), device=dev).uniform_(0, 1, generator=g)
    return w, idx


# =====================================================================================================================
# CPU baseline (the oracle's C port of the reference's numba loop; rank 0, N = 1 only)
# =====================================================================================================================
def cpu_baseline(args, n_post, n_conn, parallel=True):
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
    if not parallel:
        return res
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


def cpu_baseline_jitc(n, prob, fire, seconds):
    """The oracle's C restatement of the reference's serial numba scatter walk (brainevent/_jit_scalar/binary.py:381-416), 1 thread:
    the same matrix (shape, prob, seed 42), the walks of a bounded sample of active rows (work is proportional to them)."""
    from oracle import oracle_c
    oracle_c.build()
    rng = np.random.default_rng(0)
    n_act = 4000                                      # ~4000 x n x prob edges per call (C3: 1.6e7): a fraction of a second
    upd, t_used, calls = 0.0, 0.0, 0
    while t_used < seconds and calls < 1000:
        v = np.zeros(n, np.uint8)
        v[rng.choice(n, n_act, replace=False)] = 1
        t0 = time.perf_counter()
        out = oracle_c.jitmv('s', 1.0, 0.0, prob, v, 42, shape=(n, n), transpose=True, corder=False)
        t_used += time.perf_counter() - t0
        upd += float(out.sum())
        calls += 1
    return {'value': upd / t_used / 1e9, 'unit': 'Geff/s', 'cores': 1, 'kind': 'port',
            'sample': f'{n_act} active rows per call of the same {n} x {n} matrix (prob={prob:g}, seed 42; the full vector fires '
                      f'{int(n * fire)}), {calls} calls, C port of the reference numba walk _jit_scalar/binary.py:381-416, gcc -O3 -march=native',
            'host_cpus': os.cpu_count()}


def cpu_baseline_dense(n, batch, fire, seconds):
    """The oracle's C restatement of the reference's numba row-accumulate loop (brainevent/_dense/binary.py:579-606, f32 on the
    CPU), 1 thread: S[batch, k] @ W[k, n] on a bounded block of weight rows (only rows with a spike are ever read)."""
    from oracle import oracle_c
    oracle_c.build()
    rng = np.random.default_rng(0)
    rows = max(64, min(n, int(1.0e8 // n)))           # <= 400 MB of f32 weights (their values do not matter to the timing)
    W = np.full((rows, n), 0.5, dtype=np.float32)
    S = rng.random((batch, rows)) < fire
    pairs, t_used, calls = 0, 0.0, 0
    while t_used < seconds and calls < 1000:
        t0 = time.perf_counter()
        for b in range(batch):
            oracle_c.densemv_f32(W, S[b], True)
        t_used += time.perf_counter() - t0
        pairs += int(S.sum())
        calls += 1
    return {'value': pairs * n / t_used / 1e9, 'unit': 'Geff/s', 'cores': 1, 'kind': 'port',
            'sample': f'{rows} of {n} weight rows (f32 on the CPU), {batch} spike vectors, fire={fire:g}, {calls} passes, C port of the '
                      f'reference numba loop _dense/binary.py:579-606, gcc -O3 -march=native',
            'host_cpus': os.cpu_count()}


# =====================================================================================================================
# timing
# =====================================================================================================================
def time_steps(step, steps, warmup, fence=None):
    """W untimed steps, then exactly K steps bracketed by `fence` (synchronize [+ barrier]) on both sides — nothing else is
    issued inside that region: no profiling events, no per-step markers.  `value` / `ms_per_step` come from it.
    A SECOND pass over the same K steps (same inputs, outside the timed region) then carries the instrumentation: the
    in-library HIP events around each op's dominant kernel (on the op's own stream) and one HIP event between steps on the
    stream the calls are issued on (whole step: compaction + kernels + reduce).
    Returns (wall seconds of the K timed steps, per-step dominant-kernel ms, per-step whole-step ms, last output)."""
    from brainevent_amd import _lib
    fence = fence or torch.cuda.synchronize
    out = None
    for i in range(warmup):
        out = step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(steps):
        out = step(warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    # ---- instrumented passes (not part of `value`): each kind of event in a pass of its own, so that neither perturbs the other
    # (a marker per step costs ~7 us of step time on this runtime: 0.121 ms/step uninstrumented vs 0.128 with both kinds armed)
    prof_enable = _lib.fn('be_profile_enable', ctypes.c_int, [ctypes.c_int])
    prof_read = _lib.fn('be_profile_read', ctypes.c_int, [ctypes.c_void_p, ctypes.c_int])
    _lib.check(prof_enable(steps), 'be_profile_enable')      # (i) HIP events around each op's dominant kernel, inside the C ABI
    for i in range(steps):
        out = step(warmup + i)
    fence()
    ms = (ctypes.c_float * steps)()
    n_rec = prof_read(ctypes.cast(ms, ctypes.c_void_p), steps)
    prof_enable(0)
    kern = np.array(ms[:n_rec], dtype=np.float64) if n_rec > 0 else None
    # (ii) whole steps: one HIP event every `blk` steps on the stream the calls are issued on (torch's current stream,
    # brainevent_amd._array.stream_ptr); the per-step figure of a block is its duration / blk
    blk = 10 if steps >= 40 else 1
    n_blk = steps // blk
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(n_blk + 1)]
    marks[0].record()
    for i in range(n_blk * blk):
        out = step(warmup + i)
        if (i + 1) % blk == 0:
            marks[(i + 1) // blk].record()
    for i in range(n_blk * blk, steps):      # (the last output stays the one of step warmup + steps - 1)
        out = step(warmup + i)
    fence()
    step_ms = np.array([marks[i].elapsed_time(marks[i + 1]) / blk for i in range(n_blk)], dtype=np.float64)
    return elapsed, kern, step_ms, out


def read_ceiling_gbps(dev, gib=2, repeats=5):
    """Read-only streaming rate of this device, in this run (GB/s), or None."""
    try:
        from brainevent_amd import _lib, _array as A
        buf = torch.empty(gib << 28, dtype=torch.float32, device=dev).fill_(1.0)
        sink = torch.zeros(1, dtype=torch.int32, device=dev)
        ms = ctypes.c_float(0.0)
        f = _lib.fn('be_diag_stream_read', ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p,
                                                          ctypes.POINTER(ctypes.c_float), ctypes.c_void_p])
        torch.cuda.synchronize()
        _lib.check(f(buf.data_ptr(), buf.numel() * 4, repeats, sink.data_ptr(), ctypes.byref(ms), A.stream_ptr()), 'be_diag_stream_read')
        return round(buf.numel() * 4 / (ms.value * 1e-3) / 1e9, 1) if ms.value > 0 else None
    except Exception as e:      # a diagnostic must never sink the number
        print(f'[bench] read ceiling not measured: {e!r}', file=sys.stderr, flush=True)
        return None


def _stats(a):
    return None if a is None or len(a) == 0 else {'mean': round(float(np.mean(a)), 5), 'median': round(float(np.median(a)), 5)}


def hbm_roofline(alg_bytes, kern_ms, traffic=None, kernel=None, extra=None):
    """The roofline object of an HBM-bound kernel.  `achieved` / `frac` are SURVEY.md §8(d)'s algorithmic bytes over the
    kernel's measured duration — unless the layout stores fewer bytes than §8(d) counts and that figure would exceed the
    peak: then `achieved` / `frac` are the real (PMC) traffic rate and the algorithmic rate is reported as `effective_GBps`
    (a fraction of an HBM peak above 1 is not a bandwidth)."""
    if not kern_ms:
        return None
    sec = kern_ms * 1e-3
    alg_rate = alg_bytes / sec / 1e9
    roof = {'bound': 'hbm', 'achieved': round(alg_rate, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(alg_rate / HBM_PEAK_GBS, 4), 'traffic': traffic, 'basis': 'algorithmic bytes (SURVEY 8d)'}
    if traffic:
        roof['traffic_GBps'] = round(traffic / sec / 1e9, 1)
        roof['traffic_source'] = TRAFFIC_SOURCE
        if traffic < 0.97 * alg_bytes:
            # the layout stores fewer bytes than SURVEY 8d counts per update (d8: 5 B, h8: 1 B per entry): `achieved` / `frac` are an
            # EFFECTIVE rate on the 8 B (4 B) per update basis; what the kernel really moves is `traffic_GBps`
            roof['effective'] = True
            roof['real_frac_of_peak'] = round(traffic / sec / 1e9 / HBM_PEAK_GBS, 4)
    if alg_rate > HBM_PEAK_GBS:
        roof['effective_GBps'] = round(alg_rate, 1)
        if traffic:
            roof.update(achieved=round(traffic / sec / 1e9, 1), frac=round(traffic / sec / 1e9 / HBM_PEAK_GBS, 4),
                        basis='measured HBM traffic: the layout stores fewer bytes per update than SURVEY 8d counts')
        else:
            roof.update(achieved=None, frac=None, basis='algorithmic rate exceeds the peak (compressed layout) and no PMC '
                                                        'traffic figure exists for this configuration: see effective_GBps')
    if kernel:
        roof['kernel'] = kernel
    roof['kernel_ms'] = round(kern_ms, 5)
    roof['algorithmic_bytes_per_launch'] = int(alg_bytes)
    if extra:
        roof.update(extra)
    return roof


def traffic_lookup(key):
    """HBM bytes per launch / step of a configuration from profiles/traffic.json (separate rocprofv3 --pmc passes), or None."""
    try:
        return json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json'))).get(key, {}).get('hbm_bytes_per_launch')
    except Exception:
        return None


def sq_lookup(key):
    """SQ counters per launch of a JITC walk from profiles/sq_counters.json (separate rocprofv3 --pmc passes), or None."""
    try:
        return json.load(open(os.path.join(ROOT, 'profiles', 'sq_counters.json'))).get(key)
    except Exception:
        return None


def plan_traffic(args, plan, world):
    """PMC bytes per launch of the dominant kernel, looked up (they come from a separate rocprofv3 --pmc pass)."""
    tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
    if not (os.path.exists(tpath) and plan is not None and args.conn == 0.01 and args.fire == 0.01 and world == 1):
        return None
    try:
        tj = json.load(open(tpath))
        lay = {1: 'd8', 2: 'h8'}.get(getattr(plan, 'layout', 0), 'u16')
        return tj.get(f"{'homo' if args.homo else 'hetero'}_{lay}_n{args.n}", {}).get('hbm_bytes_per_launch')
    except Exception:
        return None


# =====================================================================================================================
# secondary single-GPU workloads (BASELINE.json configs[2..4]); same JSON shape, own metric strings
# =====================================================================================================================
def run_jitc(args, dev, g):
    import brainevent_amd as be
    n_batch = 20
    n = args.n if args.n != 1_000_000 else 4_000_000
    prob = args.conn if args.conn != 0.01 else 0.001
    M = be.JITCScalarR((np.float32(1.0), prob, 42), shape=(n, n), corder=not args.jit_gather)
    spikes = torch.rand((n_batch, n), device=dev, generator=g) < args.fire
    target = M.scatter_shard(args.jit_shard, 0) if args.jit_shard > 1 else M     # rank 0's walk classes of an N-way split
    step = lambda i: be.BinaryArray(spikes[i % n_batch]) @ target
    elapsed, kern, step_ms, out = time_steps(step, args.steps, args.warmup)
    kern_ms = float(np.mean(kern)) if kern is not None else None
    upd = float(out.double().sum().item())          # weight 1: the sum is the number of delivered edges
    value = upd * args.steps / elapsed / 1e9 if not args.jit_gather else n * n * prob * args.steps / elapsed / 1e9
    metric = 'synaptic updates/sec (Geff/s), BinaryArray @ JITCScalarR ' + ('gather: generated edges/s' if args.jit_gather else 'scatter')
    cfg = {'workload': f'BinaryArray({args.fire:g}) @ JITCScalarR w=1 prob={prob:g} seed=42 {n}x{n}, '
                       f"{'gather (corder=False matrix)' if args.jit_gather else 'scatter (corder=True matrix)'}",
           'edges_last_step': upd}
    if args.jit_gather:
        cfg['delivered_updates_Geff_per_s'] = round(upd * args.steps / elapsed / 1e9, 3)      # (`value` counts every generated edge)
    if args.jit_shard > 1:
        cfg['shard'] = f'walk classes of rank 0 of {args.jit_shard} (no stored state; outputs of the ranks are disjoint)'
    roof = None
    if kern_ms:
        # No stored matrix: HBM is not the bound, the walk's vector ALU is.  The fraction comes from the SQ counters of the walk
        # kernel (separate rocprofv3 --pmc passes, profiles/sq_counters.json + profiles/r05_c3_sq_counters.txt): SQ_ACTIVE_INST_VALU
        # counts the quad-cycles a SIMD spends issuing vector instructions; x 4 = busy cycles per launch, against the
        # 256 CUs x 4 SIMDs x 2.4 GHz the chip offers during the kernel's measured time.  (Rounds 3-4 priced a hand count of 13
        # issue slots per generated edge instead; it is kept as `valu_issue_slots_per_edge_isa` for comparison.)
        key = 'c3_gather' if args.jit_gather else 'c3'
        default_c3 = n == 4_000_000 and prob == 0.001 and args.fire == 0.01 and args.jit_shard <= 1
        sq = sq_lookup(key) if default_c3 else None
        edges = (n * n * prob) if args.jit_gather else upd
        peak = 256 * 4 * 2.4e9 / 1e9                                      # G SIMD-cycles / s
        kname = 'k_jit_mv_gather' if args.jit_gather else 'k_jit_mv_scatter'
        if sq:
            busy = 4.0 * sq['SQ_ACTIVE_INST_VALU']
            ach = busy / (kern_ms * 1e-3) / 1e9
            roof = {'bound': 'valu', 'achieved': round(ach, 1), 'peak': round(peak, 1), 'unit': 'G VALU-busy SIMD-cycles/s',
                    'frac': round(ach / peak, 4), 'traffic': None, 'kernel': kname, 'kernel_ms': round(kern_ms, 5),
                    'basis': 'SQ_ACTIVE_INST_VALU (quad-cycles, PMC, per launch) x 4 / kernel time, against 1024 SIMDs x 2.4 GHz',
                    'counters_per_launch': {k: sq[k] for k in ('SQ_INSTS_VALU', 'SQ_ACTIVE_INST_VALU', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES')
                                            if k in sq},
                    'counters_source': 'profiles/sq_counters.json (rocprofv3 --pmc, separate passes; not this run)',
                    'lane_slots_per_edge': round(64.0 * sq['SQ_INSTS_VALU'] / edges, 2), 'valu_issue_slots_per_edge_isa': 17 if args.jit_gather else 13,
                    'edges_per_launch': edges, 'equivalent_stored_matrix_GBps': round(8 * upd / (kern_ms * 1e-3) / 1e9, 1)}
        else:
            roof = {'bound': 'valu', 'achieved': None, 'peak': round(peak, 1), 'unit': 'G VALU-busy SIMD-cycles/s', 'frac': None,
                    'traffic': None, 'kernel': kname, 'kernel_ms': round(kern_ms, 5), 'edges_per_launch': edges,
                    'basis': 'no SQ counter pass exists for this configuration (profiles/sq_counters.json covers C3 at its default size)',
                    'equivalent_stored_matrix_GBps': round(8 * upd / (kern_ms * 1e-3) / 1e9, 1)}
    line = _line(metric, value, args, elapsed, 'f32', cfg, roof, kern, step_ms)
    if args.jit_shard <= 1:
        try:
            line['parity_check'] = jitc_parity(M, spikes[(args.warmup + args.steps - 1) % n_batch], out, args.jit_gather)
        except Exception as e:
            line['parity_check'] = {'error': None, 'ok': False, 'what': 'check failed to run: ' + repr(e)[:200]}
    if not args.no_cpu and not args.jit_gather:
        try:
            line['cpu_baseline'] = cpu_baseline_jitc(n, prob, args.fire, min(args.cpu_seconds, 6.0))
        except Exception as e:
            line['cpu_baseline'] = {'error': repr(e)}
    return line


def jitc_parity(M, spk, out, gather):
    """One number for the last timed JITC step (weight 1: outputs are edge counts, exact in f32).
    scatter (corder=True: the walk owners are the rows of M): sum(out) == the entries of the active rows, counted by the
    materialisation's count pass (be_jitc_csr_count) — no matrix is stored.
    gather (corder=False: M materialises column-wise): every one of 4096 sampled outputs == the number of active rows among that
    column's materialised entries (be_jitc_csr_count + be_jitc_csr_fill: the 1.6e10-entry CSC of C3 is 64 GB, dropped afterwards)."""
    if not gather:
        cnt = M.owner_counts('mv')
        want = int(cnt[spk].to(torch.int64).sum().item())
        got = float(out.double().sum().item())
        return {'what': 'sum(out) vs edges of the active rows from be_jitc_csr_count (exact)', 'error': abs(got - want), 'ok': got == want,
                'edges': want}
    csc = M.materialize('mv')
    ptr, idx = csc.indptr, csc.indices
    gsel = torch.Generator(device=out.device)
    gsel.manual_seed(5)
    cols = torch.randint(0, out.numel(), (4096,), device=out.device, generator=gsel)
    b, ln = ptr[cols], ptr[cols + 1] - ptr[cols]
    tot = int(ln.sum().item())
    off = torch.repeat_interleave(b - torch.cumsum(ln, 0) + ln, ln) + torch.arange(tot, device=out.device)
    hits = spk[idx[off].long()].to(torch.float64)
    ref = torch.zeros(cols.numel(), dtype=torch.float64, device=out.device)
    ref.index_add_(0, torch.repeat_interleave(torch.arange(cols.numel(), device=out.device), ln), hits)
    err = float((out[cols].double() - ref).abs().max().item())
    del csc, ptr, idx
    return {'what': '4096 sampled outputs vs active rows among the materialised column entries (exact)', 'error': err, 'ok': err == 0.0}


def run_fcn(args, dev, g):
    import brainevent_amd as be
    n_batch = 20
    n = args.n if args.n != 1_000_000 else 10_000_000
    K = args.k
    n_post = args.n_post or n
    w, idx = gen_fixed_num_on_device(n, K, n_post, args.homo, dev, g)
    conn = be.FixedNumPerPre((w, idx), shape=(n, n_post), check_indices=False)
    torch.cuda.synchronize()
    t_setup = time.perf_counter()
    if args.acc32 and not args.homo:
        from brainevent_amd import _csr as C
        conn.buffers['scatter_plan'] = C.BinnedScatter(w, n, n_post, n * K, indices=idx, row_len=K, acc32=True)
    conn.prepare()
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup
    spikes = torch.rand((n_batch, n), device=dev, generator=g) < args.fire
    act = spikes.sum(dim=1).cpu().numpy()
    step = lambda i: be.BinaryArray(spikes[i % n_batch]) @ conn
    elapsed, kern, step_ms, out = time_steps(step, args.steps, args.warmup)
    kern_ms = float(np.mean(kern)) if kern is not None else None
    ws_chk = conn.buffers.get('scatter_plan')
    if hasattr(ws_chk, 'check_status'):      # binned route: sticky give-up flag + conservation counters over every step above (raises)
        ws_chk.check_status()
    upd = sum(int(act[(args.warmup + i) % n_batch]) for i in range(args.steps)) * K
    value = upd / elapsed / 1e9
    metric = 'synaptic updates/sec (Geff/s), BinaryArray @ FixedNumPerPre scatter'
    ws = conn.buffers.get('scatter_plan')
    cfg = {'workload': f"BinaryArray({args.fire:g}) @ FixedNumPerPre K={K} {n} pre x {n_post} post "
                       f"{'homo' if args.homo else 'hetero'} f32, 1 GPU",
           'route': (type(ws).__name__ + (' (32-bit fixed-point sums, BE_BINNED_ACC32)' if getattr(ws, 'acc32', False) else ''))
                    if ws is not None else 'direct (global atomics)',
           'setup_s': round(t_setup, 3)}
    alg = (4 if args.homo else 8) * float(np.mean(act)) * K + n + 4 * n_post
    # the step is several kernels of similar weight on this route: the whole-step HIP-event time is the honest divisor
    whole = float(np.median(step_ms))
    default_c4 = n == 10_000_000 and K == 1000 and n_post == n and args.fire == 0.01
    roof = hbm_roofline(alg, whole, traffic=traffic_lookup('c4_homo' if args.homo else 'c4_hetero') if default_c4 else None,
                        kernel='whole step (HIP events): compaction + ' + cfg['route'] + ' kernels',
                        extra={'dominant_kernel_ms': round(kern_ms, 5) if kern_ms else None})
    try:          # last timed step vs f64 index_add of the active rows' entries (homo: the exact integer histogram)
        rows = torch.nonzero(spikes[(args.warmup + args.steps - 1) % n_batch]).flatten()
        ref = torch.zeros(n_post, dtype=torch.int64 if args.homo else torch.float64, device=dev)
        for lo in range(0, rows.numel(), 20000):
            r = rows[lo:lo + 20000]
            if args.homo:
                ref += torch.bincount(idx[r].flatten().long(), minlength=n_post)
            else:
                ref.index_add_(0, idx[r].flatten().long(), w[r].flatten().double())
        if args.homo:
            err = float((out.double() - ref.double() * float(w.flatten()[0])).abs().max().item())
        else:
            err = float(((out.double() - ref).abs() / ref.abs().clamp_min(1e-30)).max().item())
        parity = {'what': 'last timed step vs ' + ('integer histogram (max abs diff)' if args.homo else 'f64 index_add (max rel err)'),
                  'error': err, 'ok': bool(err == 0.0 if args.homo else err <= 1e-5)}
        del ref, rows
    except Exception as e:
        parity = {'error': None, 'ok': False, 'what': 'check failed to run: ' + repr(e)[:200]}
    del conn, w, idx
    line = _line(metric, value, args, elapsed, 'f32', cfg, roof, kern, step_ms)
    line['parity_check'] = parity
    if not args.no_cpu:
        try:      # FixedNumPerPre is a CSR of equal rows: the reference's loop (_fcn/binary.py:167-200) is the CSR scatter loop
            torch.cuda.empty_cache()
            a2 = argparse.Namespace(**vars(args))
            a2.n, a2.cpu_seconds = n, min(args.cpu_seconds, 6.0)
            line['cpu_baseline'] = cpu_baseline(a2, n_post, K, parallel=False)
        except Exception as e:
            line['cpu_baseline'] = {'error': repr(e)}
    return line


def run_dense(args, dev, g):
    import brainevent_amd as be
    n_batch = 20
    n = args.n if args.n != 1_000_000 else 65536
    W = torch.empty((n, n), dtype=torch.float16, device=dev).normal_(0, 1, generator=g)
    spikes = torch.rand((n_batch, args.batch, n), device=dev, generator=g) < args.fire
    step = lambda i: be.BinaryArray(spikes[i % n_batch]) @ W
    elapsed, kern, step_ms, out = time_steps(step, args.steps, args.warmup)
    kern_ms = float(np.mean(kern)) if kern is not None else None
    pairs = float(spikes.sum().item()) / n_batch
    value = pairs * n * args.steps / elapsed / 1e9
    metric = 'synaptic updates/sec (Geff/s), batched BinaryArray @ dense fp16'
    union = float(spikes.any(dim=1).sum().item()) / n_batch
    cfg = {'workload': f'BinaryArray({args.fire:g}) [{args.batch},{n}] @ dense fp16 [{n},{n}]', 'union_rows': union,
           'active_pairs': pairs}
    roof = hbm_roofline(union * n * 2 + args.batch * n * 2, kern_ms, kernel='k_densemm_mfma',
                        traffic=traffic_lookup('c5') if (n == 65536 and args.batch == 32 and args.fire == 0.01) else None) if kern_ms else None
    try:          # last timed step on 64 sampled output columns vs an f64 matmul of the same operands
        gsel = torch.Generator(device=dev)
        gsel.manual_seed(5)
        cols = torch.randint(0, n, (64,), device=dev, generator=gsel)
        S = spikes[(args.warmup + args.steps - 1) % n_batch]
        ref = S.double() @ W[:, cols].double()
        o = out if isinstance(out, torch.Tensor) else torch.as_tensor(out, device=dev)
        err = float(((o[:, cols].double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item())
        parity = {'what': 'max abs err / max |ref| over 64 sampled output columns vs f64 matmul (f16 outputs: bar 2e-3)', 'error': err,
                  'ok': bool(err <= 2e-3)}
    except Exception as e:
        parity = {'error': None, 'ok': False, 'what': 'check failed to run: ' + repr(e)[:200]}
    del W
    line = _line(metric, value, args, elapsed, 'f16', cfg, roof, kern, step_ms)
    line['parity_check'] = parity
    if not args.no_cpu:
        try:
            line['cpu_baseline'] = cpu_baseline_dense(n, args.batch, args.fire, min(args.cpu_seconds, 6.0))
        except Exception as e:
            line['cpu_baseline'] = {'error': repr(e)}
    return line


def run_gather_mirror(args, dev, g):
    """SURVEY.md 8 f1 at the headline's size: `CSR @ BinaryArray` (the unfavourable direction of the C2 matrix) evaluated
    event-driven through the mirror — built by the column-block kernels from the 1e10-entry CSR behind its int64 indptr —
    next to the gather kernel that streams the whole matrix (reference: brainevent/_csr/main.py:1647-1654)."""
    import brainevent_amd as be
    from brainevent_amd import _csr as C
    n = args.n
    n_conn = max(1, int(n * args.conn))
    w, idx, ptr = gen_csr_on_device(n, n, n_conn, args.homo, 1234, dev)
    csr = be.CSR((w, idx, ptr), shape=(n, n), check_structure=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mr = csr.build_mirror()
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t0
    n_batch = 20
    spikes = torch.rand((n_batch, n), device=dev, generator=g) < args.fire
    upd_per_vec = torch.stack([mr.counts[spikes[b]].sum() for b in range(n_batch)]).cpu().numpy()
    act = spikes.sum(dim=1).cpu().numpy()
    step = lambda i: csr @ be.BinaryArray(spikes[i % n_batch])
    elapsed, kern, step_ms, out = time_steps(step, args.steps, args.warmup)
    kern_ms = float(np.mean(kern)) if kern is not None else None
    timed = [(args.warmup + i) % n_batch for i in range(args.steps)]
    value = float(upd_per_vec[timed].sum()) / elapsed / 1e9
    last = timed[-1]
    # the checker: the gather kernel over all stored entries (one pass of the matrix per call)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ref = be.binary_csrmv(w, idx, ptr, spikes[last], shape=(n, n), transpose=False)
    e0.record()
    for _ in range(3):
        ref = be.binary_csrmv(w, idx, ptr, spikes[last], shape=(n, n), transpose=False)
    e1.record()
    torch.cuda.synchronize()
    gather_ms = e0.elapsed_time(e1) / 3
    if args.homo:
        err = float((out.double() - ref.double()).abs().max().item())
    else:
        err = float(((out.double() - ref.double()).abs() / ref.double().abs().clamp_min(1e-30)).max().item())
    plan = mr.plan if isinstance(mr.plan, C.ScatterPlan) else None
    bytes_per_upd = 4 if args.homo else 8
    alg = bytes_per_upd * float(np.mean(upd_per_vec[timed])) + n + 4 * n + 16 * float(np.mean(act[timed]))
    kname = {1: 'k_plan_accumulate_d8', 2: 'k_plan_accumulate_h8'}.get(getattr(plan, 'layout', 0), 'k_plan_accumulate') if plan else 'k_bin_stream'
    traffic = traffic_lookup(f"gather_mirror_{'homo' if args.homo else 'hetero'}_n{n}") if (args.conn == 0.01 and args.fire == 0.01) else None
    roof = hbm_roofline(alg, kern_ms, traffic=traffic, kernel=kname + ' (over the mirror)')
    cfg = {'workload': f"CSR f32 {'homo' if args.homo else 'hetero'} @ BinaryArray({args.fire:g} fire), {n} x {n}, {n_conn} synapses/row: the "
                       f"gather direction through the event-driven mirror (route={type(mr.plan).__name__})",
           'mirror_build_s': round(t_setup, 2), 'mirror_GB': round(mr.nbytes() / 1e9, 2), 'mirror_raw_arrays_released': mr.released,
           'plan_slices': (f'{plan.n_slices} x {plan.slice_width} x {plan.default_parts()} parts' if plan else None),
           'gather_kernel_ms_per_step': round(gather_ms, 3),
           'speedup_over_gather_kernel': round(gather_ms / (elapsed / args.steps * 1e3), 1)}
    line = _line('synaptic updates/sec (Geff/s), CSR @ BinaryArray through the CSC mirror', value, args, elapsed, 'f32', cfg, roof, kern,
                 step_ms)
    line['parity_check'] = {'what': 'last timed step vs the gather kernel over all stored entries (max '
                                    + ('abs diff)' if args.homo else 'rel err)'), 'error': err,
                            'ok': bool(err == 0.0 if args.homo else err <= 1e-5)}
    del csr, mr, w, idx, ptr
    return line


def _line(metric, value, args, elapsed, dtype, cfg, roof, kern, step_ms):
    return {'metric': metric, 'value': round(value, 3), 'unit': 'Geff/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 5), 'higher_is_better': True, 'scaling': args.scaling,
            'vs_baseline': None, 'dtype': dtype, 'data': 'synthetic', 'config': cfg, 'roofline': roof,
            'kernel_ms': _stats(kern), 'step_ms_hip_events': _stats(step_ms)}


# seconds per 1e5 steps the reference's examples print for themselves (examples/COBA_2005.py:98-125, examples/CUBA_2005.py:96-123),
# keyed by scale: (A6000, Ryzen 7 7840HS); firing rates 50.6 Hz (COBA) and 24-25 Hz (CUBA)
REFERENCE_TABLE = {'C1_coba': {1: (2.66, 4.44), 10: (3.17, 27.81), 100: (11.70, 215.45), 'rate': (50.6, 1.0)},
                   'C1_cuba': {1: (2.64, 1.17), 10: (3.04, 16.45), 100: (11.41, 145.35), 'rate': (24.5, 2.5)}}


def network_sweep(name, fname, scales=(1, 10, 100), steps=100_000, unroll=None):
    """The reference's own size sweep of its COBA / CUBA example on this build: seconds per 1e5 time steps (the duration the
    reference simulates) and the firing rate at each scale, beside the published figures."""
    import importlib.util
    unroll = unroll or int(os.environ.get('BENCH_NETWORK_UNROLL', 10))
    path = os.path.join(ROOT, 'examples', fname)
    spec = importlib.util.spec_from_file_location(fname[:-3] + '_example', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ref = REFERENCE_TABLE[name]
    sweep, ok = [], True
    first = None
    for sc in scales:
        n, el, rate, _, _ = mod.run_fused(float(sc), steps, graph=True, unroll=unroll, combined=True)
        first = first or (n, el, rate)
        ok = ok and abs(rate - ref['rate'][0]) <= ref['rate'][1]
        sweep.append({'scale': sc, 'neurons': n, 's_per_1e5_steps': round(el * 1e5 / steps, 3), 'rate_hz': round(rate, 2),
                      'ref_a6000_s': ref[sc][0], 'ref_ryzen_s': ref[sc][1]})
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    n, el, rate = first
    return {'metric': f'time per 0.1-ms step of the 4000-neuron network of examples/{fname} (ONE BinaryArray @ CSR scatter over both '
                      f'projections stacked n x 2n + the fused neuron step applying the two weights; HIP graph of {unroll} steps per replay)', 'value': round(el / steps * 1e6, 2), 'unit': 'us/step',
            'higher_is_better': False, 'steps': steps, 'neurons': n, 'firing_rate_hz': round(rate, 2),
            'reference_firing_rate_hz': ref['rate'][0], 'sweep': sweep,
            'parity_check': {'what': f"firing rate within {ref['rate'][1]} Hz of the reference's {ref['rate'][0]} Hz at every scale",
                             'error': round(max(abs(r['rate_hz'] - ref['rate'][0]) for r in sweep), 3), 'ok': bool(ok)}}


SECONDARY = {'jitc': run_jitc, 'fcn': run_fcn, 'dense': run_dense, 'gather_mirror': run_gather_mirror}


def secondary(args, workload=None):
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    return SECONDARY[workload or args.workload](args, dev, g)


def secondary_configs(base):
    """C3 / C4 / C5 of BASELINE.json after the headline (one GPU): compact entries for the `secondary` object."""
    out = {}
    for name, wl, extra in (('C2_gather_mirror', 'gather_mirror', ['--no-cpu']), ('C3', 'jitc', []),
                            # the reference's DEFAULT orientation of the same object (corder=False: brainevent/_jit_scalar/main.py:190-252,
                            # gather walk _jit_scalar/binary.py:353-377): every edge of the matrix is regenerated per step
                            ('C3_gather', 'jitc', ['--jit-gather', '--no-cpu', '--steps', '12', '--warmup', '3']),
                            ('C4', 'fcn', []), ('C4_homo', 'fcn', ['--homo', '--cpu-seconds', '3']), ('C5', 'dense', [])):
        a = parse(['--workload', wl, '--steps', str(base.secondary_steps), '--warmup', '10'] + extra + (['--no-cpu'] if base.no_cpu else []))
        try:
            ln = secondary(a, wl)
            out[name] = {k: ln[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'warmup', 'dtype', 'kernel_ms',
                                            'step_ms_hip_events', 'roofline')}
            out[name]['config'] = ln['config']
            if 'parity_check' in ln:
                out[name]['parity_check'] = ln['parity_check']
            if 'cpu_baseline' in ln:
                out[name]['cpu_baseline'] = ln['cpu_baseline']
        except Exception as e:       # a secondary leg must never sink the headline
            out[name] = {'error': repr(e)}
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    # SURVEY 8d's secondary list on the headline op: C2 with one shared weight (4 B / update), and the ONE operating point the
    # reference's own tuner times (brainevent/_csr/initialize.py:227-241: n_pre = n_post = 500 000, 2000 synapses per row, one
    # weight, exactly n / 250 = 2000 active rows per vector, 100 vectors, 50 warm-up + 200 timed calls) — the number a maintainer
    # of the reference can put beside their own `per_call_us`
    for name, extra in (('C2_homo', ['--homo', '--steps', '100', '--warmup', '20']),
                        ('ref_tuner_point', ['--n', '500000', '--conn', '0.004', '--fire', '0.004', '--homo', '--exact-active',
                                             '--steps', '200', '--warmup', '50'])):
        a = parse(['--no-cpu', '--no-secondary'] + extra)
        try:
            ln = run_scatter(a)
            out[name] = {k: ln[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'warmup', 'step_ms_hip_events',
                                            'parity_check', 'roofline')}
            out[name]['config'] = {k: ln['config'][k] for k in ('workload', 'plan_slices', 'plan_GB', 'setup_s', 'mean_active_rows')}
            if name == 'ref_tuner_point':
                out[name]['per_call_us'] = round(ln['ms_per_step'] * 1e3, 2)
                out[name]['reference'] = 'brainevent/_csr/initialize.py:227-241 run_benchmark defaults (per_call_us of binary_csrmv transpose)'
        except Exception as e:
            out[name] = {'error': repr(e)}
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    # what ONE rank of the 8-way strong split does per step, measured on this GPU (shard rank 0 of 8 of the same global matrix
    # + the exchange path on a one-rank RCCL group): the single-GPU evidence for the multi-GPU projection in DESIGN.md section 4
    for name, extra in (('C2_rank_of_8', []), ('C4_rank_of_8', ['--workload', 'fcn'])):
        a = parse(['--emulate-world', '8', '--steps', '100', '--warmup', '20', '--no-cpu', '--no-secondary'] + extra)
        try:
            ln = run_scatter(a)
            out[name] = {k: ln[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'warmup', 'scaling', 'step_ms_hip_events',
                                            'parity_check', 'roofline', 'rank_breakdown') if k in ln}
            out[name]['config'] = {k: ln['config'][k] for k in ('workload', 'parallelism', 'n_post_per_gpu', 'synapses_per_row_per_shard',
                                                                'plan_slices', 'setup_s')}
            out[name]['note'] = ('value = updates THIS rank delivers per second; an 8-rank job delivers 8x that if every rank keeps this step '
                                 'time with 8 real ranks in the all-gather')
        except Exception as e:
            out[name] = {'error': repr(e)}
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    # C1 of BASELINE.json (the reference's own CPU-runnable case) and the only table the reference publishes for this path: the
    # COBA / CUBA networks of examples/*_2005.py at scale 1 / 10 / 100 (4000 / 40 000 / 400 000 neurons), 1e5 steps of 0.1 ms each
    # as the reference runs them — one `spikes @ CSR` scatter over both projections (stacked n x 2n, weight 1) + the fused neuron step
    # that applies the two weights per time step, replayed as a HIP graph (bit-identical to the two-projection formulation:
    # tests/test_graph_capture_gpu.py)
    for name, fname in (('C1_coba', 'coba_2005.py'), ('C1_cuba', 'cuba_2005.py')):
        try:
            out[name] = network_sweep(name, fname)
        except Exception as e:
            out[name] = {'error': repr(e)}
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return out


# =====================================================================================================================
# headline
# =====================================================================================================================
def reference_for_shard(weights, indices, indptr, spk, n_post, homo):
    """Size-independent check of one step (SURVEY.md §8d "parity at scale"): exact integer histogram of the active rows'
    columns (homogeneous weight) or a float64 index_add of their (column, weight) pairs, on the device."""
    rows = torch.nonzero(spk).flatten()
    ptr = indptr.to(torch.int64)
    ref = torch.zeros(n_post, dtype=torch.int64 if homo else torch.float64, device=indices.device)
    for c0 in range(0, rows.numel(), 2000):
        r = rows[c0:c0 + 2000]
        b, e = ptr[r], ptr[r + 1]
        ln = e - b
        tot = int(ln.sum().item())
        if tot == 0:
            continue
        off = torch.repeat_interleave(b - torch.cumsum(ln, 0) + ln, ln) + torch.arange(tot, device=indices.device)
        cols = indices[off].long()
        if homo:
            ref += torch.bincount(cols, minlength=n_post)
        else:
            ref.index_add_(0, cols, weights[off].double())
    return ref


def run_scatter(args):
    """The scatter workload (csr, or fcn behind the exchange) on this process's rank; returns the JSON line on rank 0."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    force_dist = os.environ.get('BENCH_FORCE_DIST') == '1' or args.emulate_world > 1
    multi = world > 1 or force_dist
    is_fcn = args.workload == 'fcn'
    if args.workload != 'csr' and not (is_fcn and multi):
        return secondary(args)       # jitc / dense, and fcn on one GPU without the exchange
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world and world > 1:
        raise SystemExit(f'--gpus {args.gpus} != WORLD_SIZE {world}')
    # BENCH_MOCK_STEP=1: rehearsal of the launcher / partition / exchange / reduction / JSON plumbing on CPU tensors over gloo
    # (tests/test_bench_launch_cpu.py).  The scatter itself is replaced by a torch index_add: NOT a measurement — the line says
    # "mock_step": true and its value means nothing.
    mock = os.environ.get('BENCH_MOCK_STEP') == '1'
    if mock:
        os.environ['BENCH_BACKEND'] = 'gloo'
        dev = torch.device('cpu')
    else:
        n_dev = max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank % n_dev)      # (rehearsals put several ranks on one card; the driver has one GPU per rank)
        dev = torch.device('cuda', local_rank % n_dev)
    dist = None
    # BENCH_FORCE_DIST=1 / --emulate-world W run the multi-rank code path (process group, all-gather, max-reduce) with a
    # single rank: the only way to exercise it on a one-GPU box
    use_dist = multi
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        backend = os.environ.get('BENCH_BACKEND', 'nccl')      # nccl = RCCL; 'gloo' only to rehearse several ranks on one card
        if not dist.is_initialized():          # (a multi-rank job runs several legs over one process group: main())
            if backend == 'nccl':
                dist.init_process_group('nccl', device_id=dev)
            else:
                dist.init_process_group(backend)

    import brainevent_amd as be
    from brainevent_amd import _csr as C

    # the partition this process holds: rank `p_rank` of `p_world` post slices (emulation: rank 0 of W on one process)
    p_world, p_rank = (args.emulate_world, 0) if args.emulate_world > 1 else (world, rank)
    plan, plan_bytes = None, 0
    t_setup = time.perf_counter()
    if is_fcn:          # C4: FixedNumPerPre N = 10M, K = 1000 — as a global matrix it is a CSR with rows of K entries
        n_pre = args.n if args.n != 1_000_000 else 10_000_000
        n_post_total = args.n_post or n_pre
        n_conn_global = args.k
        args.scaling = 'strong'
    else:
        n_pre = args.n
        n_post_total = args.n * p_world if args.scaling == 'weak' else args.n
        n_conn_global = max(1, int(n_post_total * args.conn))
    if p_world == 1:
        n_post, n_conn = n_post_total, n_conn_global
        weights, indices, indptr = gen_csr_on_device(n_pre, n_post, n_conn, args.homo, 1234, dev)
        shape = (n_pre, n_post)
    else:
        weights, indices, indptr, shape, _ = gen_csr_shard_on_device(n_pre, n_post_total, n_conn_global, args.homo, 1234, dev,
                                                                      p_world, p_rank)
        n_post = shape[1]
        n_conn = n_conn_global / p_world               # mean stored synapses per (row, shard); the rows are ragged
    nnz_local = int(indices.numel())
    if not mock:
        torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_setup          # synthetic data (torch generators): not the library's setup
    t_setup = time.perf_counter()
    csr = None if mock else be.CSR((weights, indices, indptr), shape=shape, check_structure=False)
    if mock:
        pass
    elif args.route == 'plan' and not is_fcn:
        # default: LDS-filling accumulator capacity and slices balanced over the 256 CUs; --shift forces full-capacity slices
        csr.buffers['scatter_plan'] = C.ScatterPlan.build(weights, indices, indptr, shape=shape,
                                                          slice_shift=args.shift or None, slice_width=args.width or None,
                                                          layout=args.layout or None)
        if args.parts:
            csr.buffers['scatter_plan'].default_parts = lambda: args.parts
    elif args.route == 'direct':                      # global atomics
        csr.buffers['scatter_plan'] = None
    else:                                             # the container's own choice (plan / binned / direct)
        csr.prepare()
    ws_obj = None if mock else csr.buffers.get('scatter_plan')
    route = 'MOCK (torch index_add on CPU tensors)' if mock else (type(ws_obj).__name__ if ws_obj is not None else 'direct')
    if isinstance(ws_obj, C.ScatterPlan):
        plan, plan_bytes = ws_obj, ws_obj.nbytes()
    if not mock:
        torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup

    # spike batch: each rank draws the spikes of its own 1/world of the pre population
    n_batch = 100
    g = torch.Generator(device=dev)
    g.manual_seed(999 + rank)
    n_local = n_pre
    native = False
    if use_dist:
        from brainevent_amd._dist import SpikeExchange, NativeSpikeExchange
        # one all-gather per step (RCCL over xGMI); bit-packed by default: 1/8 of the bytes, consumed packed
        native = (args.exchange_impl == 'native' and args.exchange == 'bits' and os.environ.get('BENCH_BACKEND', 'nccl') == 'nccl'
                  and not mock)
        exchange = None
        if native:      # the library's own communicator; the id travels over the (already initialised) process group
            box = [None]
            try:
                box = [NativeSpikeExchange.unique_id() if rank == 0 else None]
            except Exception as e:
                print(f'[bench] rank {rank}: native exchange unavailable ({e!r})', file=sys.stderr, flush=True)
            dist.broadcast_object_list(box, src=0)
            ok = 0.0
            if box[0] is not None:
                # ncclCommInitRank blocks until every rank has joined: run it beside a watchdog so that a rendezvous that
                # never completes ends the job with a message instead of hanging it
                import threading
                res, res_lock = {}, threading.Lock()

                def _init():
                    ex_, err_ = None, None
                    try:
                        torch.cuda.set_device(dev)      # the current HIP device is per thread
                        ex_ = NativeSpikeExchange(n_pre, world, rank, box[0], device=dev)
                    except Exception as e:          # noqa: BLE001 - reported below, every rank then falls back together
                        err_ = e
                    with res_lock:                  # publish under the lock; a result nobody waits for any more is closed right here
                        if res.get('abandoned'):
                            if ex_ is not None:
                                ex_.close()
                            res['late'] = True
                        else:
                            res['ex'], res['err'] = ex_, err_
                th = threading.Thread(target=_init, daemon=True)
                th.start()
                # 60 s: a healthy 8-rank ncclCommInitRank on one node takes 2-6 s; a bootstrap that has not finished within ten times
                # that will not (the 90 s of round 4 only delayed the fallback)
                th.join(timeout=float(os.environ.get('BENCH_NATIVE_INIT_TIMEOUT', 60)))
                with res_lock:
                    if 'ex' not in res and 'err' not in res:
                        # ncclCommInitRank has not returned: its bootstrap (sockets between the ranks) is stuck, on this rank or on a
                        # peer.  The job goes on with torch.distributed's own communicator — every rank takes the same path through
                        # the MIN below; the thread closes its handle itself should it ever get one.  The line records it, and a rank
                        # whose thread is still inside the call when its work is done leaves with os._exit (main()).
                        res['abandoned'] = True
                        res['err'] = TimeoutError('be_exchange_init timed out')
                        _STATE['exchange_init_timed_out'] = True
                        _STATE['init_thread'] = th
                        print(f'[bench] rank {rank}: be_exchange_init did not return within BENCH_NATIVE_INIT_TIMEOUT; continuing with '
                              f'torch.distributed (as --exchange-impl torch would)', file=sys.stderr, flush=True)
                    if res.get('ex') is not None:
                        exchange, ok = res['ex'], 1.0
                    else:           # an error every rank can recover from together (the MIN below): torch.distributed instead
                        print(f'[bench] rank {rank}: be_exchange_init failed ({res.get("err")!r})', file=sys.stderr, flush=True)
            flag = torch.tensor([ok], dtype=torch.float64, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)          # every rank takes the same path
            if flag.item() < 1.0:
                if exchange is not None:
                    exchange.close()
                exchange, native = None, False
        if exchange is None:
            exchange = SpikeExchange(n_pre, packed=(args.exchange == 'bits'), device=dev)
        n_local = exchange.hi - exchange.lo
    if args.exact_active:
        local_spikes = torch.zeros((n_batch, n_local), dtype=torch.bool, device=dev)
        k_act = min(n_local, max(0, int(round(n_local * args.fire))))
        for b in range(n_batch):
            local_spikes[b, torch.randperm(n_local, device=dev, generator=g)[:k_act]] = True
    else:
        local_spikes = (torch.rand((n_batch, n_local), device=dev, generator=g) < args.fire)
    row_len = (indptr[1:] - indptr[:-1]).to(torch.int64)
    if use_dist:
        full = [exchange.gather(local_spikes[b]).clone() for b in range(n_batch)]
    else:
        full = [local_spikes[b] for b in range(n_batch)]
    # exact work per spike vector on THIS rank: stored synapses of the active rows
    upd_per_vec = torch.stack([row_len[f].sum() for f in full]).cpu().numpy()
    active_per_vec = torch.stack([f.sum() for f in full]).cpu().numpy()

    # the local spikes as the producer hands them over: bit-packed words (packed once here, outside the timed steps — the
    # neuron step writes them that way, be_lif_coba_step_packed) or bytes (the exchange then packs them every step)
    local_events = [local_spikes[b] for b in range(n_batch)]
    producer = 'bytes'
    if use_dist and args.exchange == 'bits' and args.producer == 'words' and not mock:
        from brainevent_amd import _array as A
        local_events = [A.PackedSpikes(be.bitpack(local_spikes[b], 0).reshape(-1), n_local) for b in range(n_batch)]
        producer = 'words'
    ahead = use_dist and args.exchange == 'bits' and args.exchange_ahead
    # the rank step with the host path cut to two C calls (brainevent_amd._dist.RankStep; BENCH_RANK_STEP=0: the operator surface,
    # `exchange.gather_events(s) @ csr`, as in rounds 1-2 — same kernels, ~25 us more host time per step)
    rank_step = None
    if use_dist and native and not mock and os.environ.get('BENCH_RANK_STEP', '1') != '0':
        from brainevent_amd._dist import RankStep
        rank_step = RankStep(exchange, csr)
    fast_ahead = bool(ahead and rank_step is not None and rank_step._fast is not None)
    ticket = None
    if fast_ahead:
        rank_step.post(local_events[0])
    elif ahead:
        ticket = [exchange.post(local_events[0])]

    def mock_scatter(full_spikes):
        ref = reference_for_shard(weights, indices, indptr, full_spikes, n_post, args.homo)
        return ref.to(torch.float32) * float(weights[0]) if args.homo else ref.to(torch.float32)

    def step(i):
        s = local_spikes[i % n_batch]
        if mock:
            return mock_scatter(exchange.gather(s) if use_dist else s)
        if fast_ahead:      # the pipelined schedule through RankStep: be_exchange_post (step i + 1), be_exchange_wait + scatter (step i)
            return rank_step.ahead(local_events[(i + 1) % n_batch])
        if ahead:
            # step i's spikes were posted during step i - 1: post step i + 1's now (the collective overlaps with the
            # scatter below), then consume step i's.  Every timed step still issues one exchange and one scatter.
            nxt = exchange.post(local_events[(i + 1) % n_batch])
            ev = exchange.wait_events(ticket[0])
            ticket[0] = nxt
            return ev @ csr
        if rank_step is not None:
            return rank_step(local_events[i % n_batch])
        if use_dist:
            return exchange.gather_events(local_events[i % n_batch]) @ csr
        return be.BinaryArray(s) @ csr

    def fence():
        if not mock:
            torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            if not mock:
                torch.cuda.synchronize()

    # The parity reference of the LAST timed step needs only the inputs (its spike vector and the matrix): it is computed here, in
    # front of the W warm-up steps, and compared with that step's output afterwards — the check stays out of the way of the per-rank
    # measurements that follow the timed region.  (It does not change the number: at the driver's flags — W = 5, K = 20, a 2.5-ms timed
    # region — three A/B pairs read 776-810 Geff/s either way; 200-step runs read 850-870: the short region is simply colder.)
    last = (args.warmup + args.steps - 1) % n_batch
    ref = reference_for_shard(weights, indices, indptr, full[last], n_post, args.homo)
    if mock:
        for i in range(args.warmup):
            out = step(i)
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = step(args.warmup + i)
        fence()
        elapsed, kern, step_ms = time.perf_counter() - t0, None, np.zeros(0)
    else:
        elapsed, kern, step_ms, out = time_steps(step, args.steps, args.warmup, fence)
    if fast_ahead:
        rank_step.drain()
    elif ahead:      # the exchange posted by the last step (never consumed)
        exchange.wait_events(ticket[0]) if native else ticket[0][1].wait()
    kern_ms = float(np.mean(kern)) if kern is not None else None

    # ---- where a rank's step goes (N > 1 / --emulate-world; after the timed region, same inputs): the exchange alone, the local
    #      scatter alone, and the OTHER schedule of the two (sequential <-> exchange posted one step ahead), each over `steps`
    #      steps between fences.  Every rank measures; min / max over the ranks are reported, so that the first run on a real
    #      multi-GPU node says whether a rank waits in the all-gather (exchange_us spread), in its scatter, or for the slowest rank.
    breakdown = None
    fast = use_dist and rank_step is not None and rank_step._fast is not None
    if use_dist and not mock:
        # the two host paths below issue different numbers of collectives: every rank must take the same one (a rank whose shard fell
        # back to the direct route has no fast path) — agreed with a MIN, like the native init result
        agree = torch.tensor([1.0 if fast else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)
        fast = bool(agree.item() >= 1.0)
    if use_dist and not mock:
        def timed(fn_i, prime=None, finish=None):
            if prime:
                prime()
            for i in range(min(args.warmup, 10)):
                fn_i(i)
            fence()
            t0 = time.perf_counter()
            for i in range(args.steps):
                fn_i(args.warmup + i)
            fence()
            el = (time.perf_counter() - t0) / args.steps * 1e6
            if finish:
                finish()
            return el
        if fast:
            ex_us = timed(lambda i: rank_step.exchange_only(local_events[i % n_batch]))
            sc_us = timed(lambda i: rank_step.scatter_only())
            if ahead:
                other_name, other_us = 'sequential', timed(lambda i: rank_step(local_events[i % n_batch]))
            else:
                other_name = 'exchange_ahead_1'
                other_us = timed(lambda i: rank_step.ahead(local_events[(i + 1) % n_batch]),
                                 prime=lambda: rank_step.post(local_events[0]), finish=rank_step.drain)
            # one process emulating a rank of W: the one-rank "all-gather" is a 4-us device copy.  Both schedules once more against an
            # exchange of realistic LENGTH (be_exchange_emulate_latency_us: a spin kernel behind every all-gather, on its stream — the
            # latency of a real 8-rank all-gather stood in for, not RCCL): what the pipelined schedule hides shows here
            emulated = None
            if args.emulate_world > 1 and world == 1:
                try:
                    from brainevent_amd import _lib as L
                    emu = L.fn('be_exchange_emulate_latency_us', ctypes.c_int, [ctypes.c_double])
                    add_us = float(os.environ.get('BENCH_EMULATE_EXCHANGE_US', 9.0))
                    L.check(emu(add_us), 'be_exchange_emulate_latency_us')
                    try:
                        e_seq = timed(lambda i: rank_step(local_events[i % n_batch]))
                        e_ahead = timed(lambda i: rank_step.ahead(local_events[(i + 1) % n_batch]),
                                        prime=lambda: rank_step.post(local_events[0]), finish=rank_step.drain)
                    finally:
                        emu(0.0)
                    emulated = {'exchange_plus_us': add_us, 'sequential_us': round(e_seq, 2), 'exchange_ahead_1_us': round(e_ahead, 2)}
                except Exception as e:          # a diagnostic must never sink the line
                    emulated = {'error': repr(e)[:120]}
        else:           # the operator surface (torch.distributed exchange, or a shard without a fixed-point workspace)
            ex_us = timed(lambda i: exchange.gather_events(local_events[i % n_batch]))
            ev0 = exchange.gather_events(local_events[0])
            sc_us = timed(lambda i: ev0 @ csr)
            other_name, other_us = ('sequential' if ahead else 'exchange_ahead_1'), -1.0       # (not measured on this host path)
            emulated = None
        v = torch.tensor([ex_us, sc_us, other_us, elapsed / args.steps * 1e6], dtype=torch.float64, device=dev)
        vmax, vmin = v.clone(), v.clone()
        dist.all_reduce(vmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(vmin, op=dist.ReduceOp.MIN)
        mx, mn = vmax.cpu().tolist(), vmin.cpu().tolist()
        this_name = 'exchange_ahead_1' if ahead else 'sequential'
        breakdown = {'what': 'us per step on each rank over the same steps, measured after the timed region; {min, max} over the ranks',
                     'exchange_only_us': {'min': round(mn[0], 2), 'max': round(mx[0], 2)},
                     'scatter_only_us': {'min': round(mn[1], 2), 'max': round(mx[1], 2)},
                     'step_us': {'schedule': this_name, 'min': round(mn[3], 2), 'max': round(mx[3], 2)},
                     'not_in_kernels_us': round(mx[3] - mx[0] - mx[1], 2), 'host_path': 'RankStep (two C calls per step)' if fast else 'operator surface',
                     'other_schedule': {'schedule': other_name,
                                        'step_us': {'min': round(mn[2], 2), 'max': round(mx[2], 2)} if mx[2] >= 0 else None,
                                        'note': 'exchange_ahead_1: the all-gather of step t + 1 is posted before step t is scattered '
                                                '(valid for synaptic delays >= 2 steps); whole-job value at this schedule = value x '
                                                'step_us.max / other step_us.max'}}
        if emulated is not None:
            breakdown['emulated'] = emulated

    if hasattr(ws_obj, 'check_status'):       # binned route: sticky give-up flag + conservation counters over every step above (raises)
        ws_obj.check_status()
    # one-step parity check of what was timed (every rank checks its own slice; rank 0 reports the worst)
    if args.homo:
        err = float((out.to(torch.float64) - ref.to(torch.float64) * float(weights[0])).abs().max().item())
    else:
        err = float(((out.double() - ref).abs() / ref.abs().clamp_min(1e-30)).max().item())
    stats = torch.tensor([elapsed, err], dtype=torch.float64, device=dev)
    nnz_sum = torch.tensor([float(nnz_local), float(upd_per_vec[[(args.warmup + i) % n_batch for i in range(args.steps)]].sum())],
                           dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
        dist.all_reduce(nnz_sum, op=dist.ReduceOp.SUM)
    elapsed, err = float(stats[0].item()), float(stats[1].item())
    total_nnz, total_upd = float(nnz_sum[0].item()), float(nnz_sum[1].item())
    value = total_upd / elapsed / 1e9
    checksum = float(out.double().sum().item())

    read_gbps = None
    line = None
    if rank == 0 and args.emulate_world <= 1 and not mock:
        read_gbps = read_ceiling_gbps(dev)
    if rank == 0:
        timed = [(args.warmup + i) % n_batch for i in range(args.steps)]
        bytes_per_upd = 4 if args.homo else 8           # int32 index (+ f32 weight)
        mean_active = float(np.mean(active_per_vec[timed]))
        mean_upd = float(np.mean(upd_per_vec[timed]))
        alg_bytes = bytes_per_upd * mean_upd + n_pre * 1 + n_post * 4 + 16 * mean_active
        plan_kernel = {1: 'k_plan_accumulate_d8', 2: 'k_plan_accumulate_h8'}.get(getattr(plan, 'layout', 0), 'k_plan_accumulate')
        kernel_name = {'ScatterPlan': plan_kernel, 'BinnedScatter': 'k_bin_stream'}.get(route, 'k_csrmv_t_direct')
        traffic = plan_traffic(args, plan, p_world)
        if traffic is None and args.emulate_world == 8 and args.fire == 0.01 and not args.homo:
            if is_fcn and n_pre == 10_000_000 and args.k == 1000:
                traffic = traffic_lookup('c4_rank_of_8')
            elif not is_fcn and args.n == 1_000_000 and args.conn == 0.01:
                traffic = traffic_lookup('c2_rank_of_8')
        roof = hbm_roofline(alg_bytes, kern_ms, traffic=traffic, kernel=kernel_name,
                            extra={'kernel_ms_median': round(float(np.median(kern)), 5) if kern is not None else None,
                                   # the same run's measured ceiling next to the nominal peak — a READ-ONLY
                                   # stream (be_diag_stream_read: 2 GiB, 16 B per lane), the ceiling of a read-dominated kernel
                                   # (a device copy reads AND writes: 4.9 TB/s, below what the accumulate kernel itself moves)
                                   'read_ceiling_GBps': read_gbps})
        if roof and read_gbps and roof.get('traffic') and kern_ms:
            roof['real_frac_of_read_ceiling'] = round(roof['traffic'] / (kern_ms * 1e-3) / 1e9 / read_gbps, 4)
        what = 'FixedNumPerPre' if is_fcn else 'CSR'
        scaling = args.scaling      # how `--gpus N` partitions: strong = this same problem cut into N post slices
        line = {
            'metric': f'synaptic updates/sec (Geff/s), BinaryArray @ {what} scatter'
                      + (f' [{scaling} scaling over {max(world, args.emulate_world)} post slices]'
                         if world > 1 or args.emulate_world > 1 else ''),
            'value': round(value, 3), 'unit': 'Geff/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 5), 'higher_is_better': True, 'scaling': scaling,
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': (f"BinaryArray({args.fire:g} fire) @ {what} f32 {'homo' if args.homo else 'hetero'}, "
                                    f"{n_pre} pre x {n_post_total} post, {n_conn_global} synapses/row "
                                    f"({n_conn_global / n_post_total:g} density), route={route}"),
                       'n_pre': n_pre, 'n_post': n_post_total, 'n_post_per_gpu': n_post, 'n_conn': n_conn_global,
                       'synapses_per_row_per_shard': n_conn, 'stored_synapses_total': total_nnz,
                       'parallelism': f'post-slice x{p_world}' + (f' + spike all-gather ({args.exchange}, '
                                                                  + ('be_exchange_* / RCCL' if native else 'torch.distributed')
                                                                  + (', posted one step ahead' if ahead else '')
                                                                  + (', step issued as two C calls (RankStep)' if rank_step is not None else '')
                                                                  + (', local spikes arrive as packed words' if producer == 'words'
                                                                     else ', local spikes arrive as bytes and are packed per step')
                                                                  + ')' if use_dist else '')
                                      + (f' [one process emulating rank 0 of {args.emulate_world}]' if args.emulate_world > 1 else ''),
                       'plan_GB': round(plan_bytes / 1e9, 2), 'setup_s': round(t_setup, 2), 'data_gen_s': round(t_gen, 2),
                       'plan_slices': (f'{plan.n_slices} x {plan.slice_width} columns x {plan.default_parts()} parts, '
                                       f"layout {({1: 'd8 (5 B/entry)', 2: 'h8 (1 B/entry)'}.get(plan.layout, 'u16'))}" if plan is not None else None),
                       'mean_active_rows': mean_active, 'checksum': checksum},
            'step_ms_hip_events': _stats(step_ms),
            'timing': ('value / ms_per_step: wall clock over exactly `steps` uninstrumented steps between synchronize (+ barrier) fences; '
                       'roofline.kernel_ms and step_ms_hip_events: HIP events in two separate passes over the same steps afterwards '
                       '(in-library events around the dominant kernel; one event per 10 steps on the issuing stream)'),
            'parity_check': {'what': 'last timed step vs ' + ('exact integer histogram of the active rows\' columns (max abs diff)'
                                                              if args.homo else 'float64 index_add of the active rows\' entries (max rel err)')
                                     + ', every rank on its own slice, worst rank reported',
                             'error': err, 'ok': bool(err == 0.0 if args.homo else err <= 1e-5),
                             'stored_synapses_all_ranks': total_nnz,
                             'expected_stored_synapses': float(n_pre) * n_conn_global if args.emulate_world <= 1 else None},
            'roofline': roof,
        }
        if breakdown is not None:
            line['rank_breakdown'] = breakdown
        if _STATE.get('exchange_init_timed_out'):
            line['exchange_init_timed_out'] = True
        if mock:
            line['mock_step'] = True
            line['data'] = 'synthetic (MOCK STEP: plumbing rehearsal on CPU tensors, not a measurement)'
        default_cfg = (args.n == 1_000_000 and args.conn == 0.01 and args.fire == 0.01 and not args.homo and args.route == 'plan')
        if world == 1 and not force_dist and not args.no_cpu and not mock:
            try:
                line['cpu_baseline'] = cpu_baseline(args, n_post, int(n_conn_global))
            except Exception as e:   # the CPU leg must never sink the GPU number
                line['cpu_baseline'] = {'error': repr(e)}
        if world == 1 and not force_dist and not args.no_secondary and default_cfg and args.workload == 'csr' and not mock:
            line['secondary'] = 'pending'      # filled in by main() once this call's process group and buffers are gone
    if use_dist:
        if native:
            exchange.close()
        dist.barrier()
        if not getattr(args, 'keep_group', False):
            dist.destroy_process_group()
    return line


# =====================================================================================================================
# the printed line: numbers per entry, prose once (the driver's record keeps an 8 KB tail of stdout — the whole line must fit)
# =====================================================================================================================
LINE_BYTE_BOUND = 8000

LEGEND = {
    'timing': 'value, ms_per_step: wall clock over exactly `steps` bare steps between synchronize(+barrier) fences, inputs in HBM; kernel_ms '
              '(dominant kernel: in-library HIP events on its stream) and step_ms (HIP events, issuing stream): medians of separate '
              'passes over the same steps',
    'roofline': 'hbm: frac = SURVEY 8d algorithmic bytes per launch / kernel_ms / 8000 GB/s; real = PMC HBM bytes per launch '
                '(profiles/traffic.json, separate rocprofv3 --pmc passes) / kernel_ms / peak; real_rd = same / read_ceiling_GBps '
                '(be_diag_stream_read, 2 GiB, 16 B/lane, this run); C4*: kernel_ms = whole step. valu (C3*): 4 x SQ_ACTIVE_INST_VALU per '
                'launch (profiles/sq_counters.json) / kernel_ms / (1024 SIMDs x 2.4 GHz)',
    'parity': '[error, ok], last timed step. headline, C2_homo, ref_tuner_point, C4*, *_rank_of_8: max rel err vs f64 index_add of the '
              'active rows (homo: max abs diff vs integer histogram); C2_gather_mirror: vs the full-matrix gather kernel; C3: |sum(out) '
              '- edges of active rows from be_jitc_csr_count|; C3_gather: 4096 sampled outputs vs the materialised columns; C5: max abs '
              'err / max|ref| on 64 sampled columns vs f64 matmul (bar 2e-3); C1_*: max |rate - reference rate| Hz over the sweep',
    'cpu_Geff_s': 'oracle C port of the reference numba loop, 1 thread, bounded sample, this host',
    'rank_breakdown': 'us per step [min, max over ranks]: ex = exchange alone, sc = scatter alone, step = the timed schedule, other = '
                      'the other schedule (seq <-> ahead: all-gather of step t+1 posted before step t scatters); emul = [us, seq, ahead]: '
                      'both schedules with every all-gather lengthened by `us` (spin kernel on its stream: the latency of a real 8-rank '
                      'all-gather stood in for on one GPU)',
    'sweep': '[scale, neurons, s per 1e5 steps here, rate Hz, reference A6000 s, reference Ryzen 7840HS s] (examples/COBA_2005.py:98-125, '
             'CUBA_2005.py:96-123)',
    'full': 'verbose line (kernel names, bytes, samples): --full-line-file, default gpurun_out/bench_full_line.json',
}

WORKLOADS = {
    'C2_gather_mirror': 'CSR @ spk via the CSC mirror, headline matrix', 'C3': 'spk @ JITCScalarR 4M^2 p=1e-3, scatter',
    'C3_gather': 'same, corder=False (reference default); value = generated Gedges/s', 'C4': 'spk @ FixedNumPerPre K=1000 N=1e7 hetero',
    'C4_homo': 'same, one weight', 'C5': 'spk[32,64k] @ dense f16 64k^2', 'C2_homo': 'headline, one weight',
    'ref_tuner_point': '_csr/initialize.py:227-241: n=5e5, 2000/row, homo, 2000 active', 'C2_rank_of_8': 'rank 0 of the 8-way '
    'post-slice cut of the headline + one-rank RCCL exchange', 'C4_rank_of_8': 'same for C4', 'C1_coba': 'examples/coba_2005.py, HIP '
    'graph', 'C1_cuba': 'examples/cuba_2005.py', 'C4_strong': 'C4 post-sliced over the N ranks', 'C2_weak': 'a headline slice per rank',
}


def _sig(x, n=4):
    if isinstance(x, bool) or x is None or isinstance(x, str):
        return x
    if isinstance(x, (int, np.integer)):
        return int(x)
    x = float(x)
    if not math.isfinite(x):
        return None
    if x == 0.0:
        return 0
    r = float(f'{x:.{n}g}')
    return int(r) if r == int(r) and abs(r) < 1e15 else r


def _compact_roof(roof, read_ceiling=None):
    if not roof:
        return None
    out = {k: roof.get(k) for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')}
    out['achieved'], out['frac'] = _sig(out['achieved'], 5), _sig(out['frac'])
    for k_in, k_out in (('kernel', 'kernel'), ('kernel_ms', 'kernel_ms'), ('real_frac_of_peak', 'real'), ('effective_GBps', 'effective_GBps'),
                        ('lane_slots_per_edge', 'lane_slots_per_edge')):
        if roof.get(k_in) is not None:
            out[k_out] = _sig(roof[k_in], 5) if not isinstance(roof[k_in], str) else roof[k_in].split(' (')[0].split(':')[0]
    if roof.get('traffic') and roof.get('kernel_ms') and 'real' not in out and roof.get('bound') == 'hbm':
        out['real'] = _sig(roof['traffic'] / (roof['kernel_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS)
    rc = roof.get('read_ceiling_GBps') or read_ceiling
    if rc and roof.get('traffic') and roof.get('kernel_ms') and roof.get('bound') == 'hbm':
        out['real_rd'] = _sig(roof['traffic'] / (roof['kernel_ms'] * 1e-3) / 1e9 / rc)
    return out


def _compact_breakdown(rb):
    if not rb:
        return None
    mm = lambda d: None if not d else [_sig(d['min'], 5), _sig(d['max'], 5)]
    o = rb.get('other_schedule') or {}
    out = {'schedule': rb['step_us']['schedule'], 'ex': mm(rb['exchange_only_us']), 'sc': mm(rb['scatter_only_us']), 'step': mm(rb['step_us']),
           'other_schedule': o.get('schedule'), 'other': mm(o.get('step_us'))}
    for k in ('host_path', 'graph', 'other_graph'):
        if rb.get(k):
            out[k] = rb[k].split(' (')[0] if isinstance(rb[k], str) else rb[k]
    em = rb.get('emulated')
    if em and 'error' not in em:       # [exchange lengthened by us, sequential us, pipelined us]
        out['emul'] = [_sig(em['exchange_plus_us']), _sig(em['sequential_us'], 5), _sig(em['exchange_ahead_1_us'], 5)]
    return out


def _compact_entry(ln, read_ceiling=None, top=False):
    if 'error' in ln and 'value' not in ln:
        return {'error': str(ln['error'])[:160]}
    e = {'value': _sig(ln['value'], 5)}
    if ln.get('unit', 'Geff/s') != 'Geff/s':
        e['unit'] = ln['unit']
    for k in ('ms_per_step', 'n_gpus'):
        if k in ln:
            e[k] = _sig(ln[k], 5)
    if ln.get('step_ms_hip_events'):
        e['step_ms'] = _sig(ln['step_ms_hip_events']['median'], 5)
    r = _compact_roof(ln.get('roofline'), read_ceiling)
    if r:
        e['roofline'] = {k: r[k] for k in ('bound', 'frac', 'kernel_ms', 'real', 'real_rd', 'effective_GBps', 'lane_slots_per_edge')
                         if r.get(k) is not None}
    pc = ln.get('parity_check')
    if pc:
        e['parity'] = [_sig(pc.get('error'), 3), bool(pc.get('ok'))]
    cb = ln.get('cpu_baseline')
    if cb and 'value' in cb:
        e['cpu_Geff_s'] = _sig(cb['value'])
    rb = _compact_breakdown(ln.get('rank_breakdown'))
    if rb:
        e['rank_breakdown'] = rb
    for k in ('per_call_us', 'firing_rate_hz', 'reference_firing_rate_hz', 'setup_s'):
        if k in ln:
            e[k] = ln[k]
    if 'sweep' in ln:          # rows: [scale, neurons, s per 1e5 steps, rate Hz, reference A6000 s, reference Ryzen s]
        e['sweep'] = [[r['scale'], r['neurons'], r['s_per_1e5_steps'], r['rate_hz'], r['ref_a6000_s'], r['ref_ryzen_s']] for r in ln['sweep']]
    cfg = ln.get('config') or {}
    for k in ('setup_s', 'mirror_build_s', 'speedup_over_gather_kernel'):
        if cfg.get(k) is not None and k not in e:
            e[k] = cfg[k] if isinstance(cfg[k], str) else _sig(cfg[k])
    return e


def compact_line(line):
    """The line as printed: the driver contract's keys in full on the headline, numbers only per secondary, every prose string once in
    `legend` (LEGEND / WORKLOADS).  The verbose form (what run_* return) goes to --full-line-file."""
    roof = line.get('roofline') or {}
    rc = roof.get('read_ceiling_GBps')
    out = {k: line.get(k) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                                    'vs_baseline', 'dtype', 'data')}
    cfg = line.get('config') or {}
    out['config'] = {k: (_sig(cfg[k], 12) if not isinstance(cfg[k], str) else cfg[k]) for k in
                     ('workload', 'parallelism', 'n_post_per_gpu', 'n_conn', 'stored_synapses_total', 'plan_slices', 'plan_GB', 'setup_s',
                      'mean_active_rows', 'checksum') if cfg.get(k) is not None}
    if line.get('step_ms_hip_events'):
        out['step_ms'] = _sig(line['step_ms_hip_events']['median'], 5)
    pc = line.get('parity_check')
    if pc:
        out['parity_check'] = {k: _sig(pc[k], 12) for k in ('error', 'ok', 'stored_synapses_all_ranks', 'expected_stored_synapses') if k in pc}
    r = _compact_roof(roof)
    if r:
        for k in ('algorithmic_bytes_per_launch', 'traffic_GBps', 'read_ceiling_GBps', 'kernel_ms_median'):
            if roof.get(k) is not None:
                r[k] = _sig(roof[k], 10 if k.startswith('alg') else 5)
    out['roofline'] = r
    cb = line.get('cpu_baseline')
    if cb:
        c = {k: (_sig(cb[k]) if not isinstance(cb[k], str) else cb[k]) for k in ('value', 'unit', 'cores', 'kind', 'sample', 'host_cpus', 'error')
             if k in cb}
        pv = cb.get('parallel_atomics_variant')
        if pv and 'value' in pv:
            c['omp_atomics_variant'] = {'value': _sig(pv['value']), 'cores': pv['cores']}
        out['cpu_baseline'] = c
    rb = _compact_breakdown(line.get('rank_breakdown'))
    if rb:
        out['rank_breakdown'] = rb
    for k in ('mock_step', 'exchange_init_timed_out', 'extras'):
        if k in line:
            out[k] = line[k]
    sec = line.get('secondary')
    if isinstance(sec, dict):
        out['secondary'] = {k: (_compact_entry(v, rc) if isinstance(v, dict) else v) for k, v in sec.items()}
        wl = {k: WORKLOADS[k] for k in sec if k in WORKLOADS}
    else:
        wl = {}
    out['legend'] = dict(LEGEND, workloads=wl) if isinstance(sec, dict) else {k: LEGEND[k] for k in ('timing', 'roofline', 'full')}
    has_rb = bool(line.get('rank_breakdown')) or (isinstance(sec, dict) and any(isinstance(v, dict) and v.get('rank_breakdown')
                                                                                  for v in sec.values()))
    if has_rb:
        out['legend']['rank_breakdown'] = LEGEND['rank_breakdown']
    else:
        out['legend'].pop('rank_breakdown', None)
    if not (isinstance(sec, dict) and any(isinstance(v, dict) and 'sweep' in v for v in sec.values())):
        out['legend'].pop('sweep', None)
    return out


def emit(line, args):
    """Rank 0: the verbose line to --full-line-file, the compact one (or the verbose one with --full-line) on stdout."""
    full = json.dumps(_finite(line), allow_nan=False)
    path = getattr(args, 'full_line_file', '')
    if path:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            with open(path, 'w') as f:
                f.write(full + '\n')
        except OSError:
            pass
    if getattr(args, 'full_line', False):
        print(full, flush=True)
        return
    txt = json.dumps(_finite(compact_line(line)), allow_nan=False, separators=(',', ':'))
    if len(txt) > LINE_BYTE_BOUND:          # never silently outgrow the driver's record: drop the legend's workload map, then say so
        c = compact_line(line)
        c['legend'] = {'see': 'bench.py LEGEND / WORKLOADS', 'line_bytes_before_trim': len(txt)}
        txt = json.dumps(_finite(c), allow_nan=False, separators=(',', ':'))
    print(txt, flush=True)


def _finite(o):
    """Strict JSON: a non-finite float (an error of inf against an all-zero reference, ...) becomes null."""
    if isinstance(o, float):
        return o if math.isfinite(o) else None
    if isinstance(o, dict):
        return {k: _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    if isinstance(o, np.generic):
        return _finite(o.item())
    return o


def multi_rank_extras(args, line, world):
    """After the headline of a multi-rank job (the default `--gpus N` run): the SAME N ranks also measure, over the same process group,
      * `C4_strong` — BASELINE config C4 as it is defined: FixedNumPerPre K = 1000, N = 10M, post-sliced over the N GPUs with the
        spike all-gather (`--workload fcn`), and
      * `C2_weak`   — the headline problem with every rank holding a full 1M x 1M slice (`--scaling weak`: per-rank work fixed, only
        the exchange is added),
    and rank 0 attaches both to `secondary` of the ONE line it prints.  A watchdog bounds the legs together
    (BENCH_EXTRAS_SECONDS, default 240 s): when it expires every rank stops where it is — rank 0 prints the headline line it
    already has, with whatever legs finished and `extras` = {incomplete: why, leg, rank} — and the process ends with a NON-ZERO status
    (3 = stalled, 4 = a leg raised on this rank; the line is on stdout by then).  The watchdog stays armed through the closing
    barrier, and a rank whose leg raises prints / leaves before it could enter another collective: a failure or a stall in a leg can
    cost the legs and the exit status, never the headline.  (`value` of a leg = synaptic updates of all ranks per second.)"""
    import threading
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    done = threading.Event()
    out = {}
    state = {'leg': None}
    if line is not None:
        line['secondary'] = out

    def emit_and_exit(why, code):
        """Rank 0 prints the line as it stands (a snapshot of the legs: the main thread may be filling `out` right now), every rank
        then leaves at once with `code` — nothing runs interpreter teardown over a collective that is still in flight."""
        try:
            if rank == 0 and line is not None:
                snap = {k: v for k, v in list(out.items())}
                ln = {k: v for k, v in line.items() if k != 'secondary'}
                ln['secondary'] = snap
                ln['extras'] = {'incomplete': why, 'leg': state['leg'], 'rank': rank}
                try:
                    emit(ln, args)
                except Exception:
                    ln['secondary'] = {}
                    emit(ln, args)
        finally:
            sys.stdout.flush()
            os._exit(code)

    def watchdog():
        # armed until the process group is gone: a rank that waits in the closing barrier for a peer that never arrives is ended here too
        if not done.wait(float(os.environ.get('BENCH_EXTRAS_SECONDS', 240))):
            print(f'[bench] rank {rank}: extra leg {state["leg"]} did not finish within BENCH_EXTRAS_SECONDS', file=sys.stderr, flush=True)
            emit_and_exit(f'timed out in leg {state["leg"]} (BENCH_EXTRAS_SECONDS)', 3)
    threading.Thread(target=watchdog, daemon=True).start()
    # A launcher that sees one rank fail sends SIGTERM to the others (torch.distributed.run does) — while their main thread sits inside a
    # collective, where no Python-level handler runs.  The C-level handler writes the signal number to a pipe at once; a thread reads it
    # and prints / leaves as the watchdog would.
    try:
        import signal
        r_fd, w_fd = os.pipe()
        os.set_blocking(w_fd, False)
        signal.signal(signal.SIGTERM, lambda *_: None)
        signal.set_wakeup_fd(w_fd, warn_on_full_buffer=False)

        def on_signal():
            os.read(r_fd, 1)
            if not done.is_set():
                emit_and_exit(f'SIGTERM from the launcher during leg {state["leg"]} (a peer rank failed)', 3)
        threading.Thread(target=on_signal, daemon=True).start()
    except Exception as e:          # (not the main thread, or no signals on this platform: the watchdog alone then)
        print(f'[bench] rank {rank}: no SIGTERM relay ({e!r})', file=sys.stderr, flush=True)
    mock = os.environ.get('BENCH_MOCK_STEP') == '1'
    legs = (('C4_strong', ['--workload', 'fcn', '--steps', '100', '--warmup', '20']),
            ('C2_weak', ['--scaling', 'weak', '--steps', '100', '--warmup', '20']))
    for i, (name, extra) in enumerate(legs):
        a = parse(['--gpus', str(args.gpus), '--no-cpu', '--no-secondary', '--exchange', args.exchange, '--exchange-impl', args.exchange_impl]
                  + extra + os.environ.get('BENCH_EXTRAS_ARGS', '').split())        # (BENCH_EXTRAS_ARGS: smaller sizes for rehearsals)
        a.keep_group = True
        state['leg'] = name
        try:
            if not mock:
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
            if os.environ.get('BENCH_EXTRAS_FAIL') == f'{name}:{rank}':      # rehearsal hook (tests/test_bench_launch_cpu.py)
                raise RuntimeError('injected failure')
            ln = run_scatter(a)
            if ln is not None:
                out[name] = {k: ln[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'ms_per_step', 'steps', 'warmup', 'scaling',
                                                'parity_check', 'roofline', 'rank_breakdown', 'step_ms_hip_events') if k in ln}
                out[name]['config'] = {k: ln['config'][k] for k in ('workload', 'parallelism', 'n_post_per_gpu', 'setup_s', 'data_gen_s')
                                       if k in ln['config']}
        except Exception as e:
            # This rank is out of step with the others (they are inside the leg's collectives, or will be): it must not enter another
            # collective.  Rank 0 prints what it has at once; every rank that fails leaves with a non-zero status — the peers are then
            # ended by their communicator's abort or by their own watchdog, whichever comes first.
            out[name] = {'error': repr(e)[:300], 'rank': rank}
            print(f'[bench] rank {rank}: leg {name} failed: {e!r}', file=sys.stderr, flush=True)
            emit_and_exit(f'leg {name} raised on rank {rank}: {e!r}'[:200], 4)
    state['leg'] = 'closing barrier'
    try:
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        pass
    done.set()
    return line


def launch_ranks(args, argv):
    """`python bench.py --gpus N` (N > 1) without a launcher: start the N ranks as a CHILD process (torch.distributed.run,
    one rank per GPU, rendezvous on 127.0.0.1), relay what rank 0 prints and return the child's exit code.  Runs before
    anything in this process has touched the GPU, and never replaces this process (no exec)."""
    import socket
    import subprocess
    with socket.socket() as sk:                       # a free rendezvous port
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC: RCCL between processes needs it on this driver
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ and args.emulate_world <= 1:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    if os.environ.get('BENCH_FAULT_TIMEOUT'):        # rehearsals: where is every rank if the run has not finished by then
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ['BENCH_FAULT_TIMEOUT']), exit=False, file=sys.stderr)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    extra_legs = (world > 1 and args.workload == 'csr' and args.scaling == 'strong' and not args.no_secondary and args.emulate_world <= 1
                  and (os.environ.get('BENCH_MOCK_STEP') != '1' or os.environ.get('BENCH_EXTRAS_FORCE') == '1')
                  and ((args.n == 1_000_000 and args.conn == 0.01 and args.fire == 0.01) or os.environ.get('BENCH_EXTRAS_FORCE') == '1'))
    args.keep_group = extra_legs
    line = run_scatter(args)
    if extra_legs:
        line = multi_rank_extras(args, line, world)
    if line is not None and line.get('secondary') == 'pending':
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        line['secondary'] = secondary_configs(args)
    if line is not None:
        emit(line, args)
    th = _STATE.get('init_thread')
    if th is not None and th.is_alive():
        # a thread of this process is still inside ncclCommInitRank (be_exchange_init never returned): no interpreter teardown over it
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(5)


if __name__ == '__main__':
    main()
