"""The multi-GPU step on the device with a one-rank RCCL process group (the only multi-rank configuration a one-GPU box
can run): bit-packed exchange -> packed events -> planned scatter, sequential and pipelined schedules, JITC walk-class
shard.  The partition logic itself is covered under gloo with two ranks in test_dist_cpu.py."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _init_one_rank_group():
    """A one-rank RCCL group on a free local port (the port found by bind(0) can be taken again before the store listens on it:
    seen once as EADDRINUSE — try another)."""
    import torch.distributed as dist
    last = None
    for _ in range(5):
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        try:
            dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
            return dist
        except Exception as e:           # DistNetworkError (address in use)
            last = e
    raise last


@pytest.fixture(scope='module')
def one_rank_group():
    dist = _init_one_rank_group()
    yield dist
    dist.destroy_process_group()


def test_exchange_and_scatter_on_device(one_rank_group):
    import brainevent_amd as be
    from brainevent_amd import _dist as D
    from brainevent_amd._csr import ScatterPlan
    from oracle import oracle_np as O
    rng = np.random.default_rng(0)
    dev = torch.device('cuda', 0)
    n_pre, n_post = 5003, 30011
    lens = rng.integers(0, 60, n_pre)
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    idx = rng.integers(0, n_post, ptr[-1]).astype(np.int32)
    w = rng.random(ptr[-1]).astype(np.float32)
    sw, si, sp, sshape = D.shard_csr_by_post(torch.from_numpy(w).to(dev), torch.from_numpy(idx).to(dev),
                                             torch.from_numpy(ptr).to(dev), (n_pre, n_post), 1, 0)
    shard = be.CSR((sw, si, sp), shape=sshape, check_structure=False)
    shard.buffers['scatter_plan'] = ScatterPlan.build(sw, si, sp, shape=sshape)
    for packed in (True, False):
        ds = D.DistributedScatter(shard, n_pre, packed=packed, device=dev)
        assert (ds.exchange.lo, ds.exchange.hi) == (0, n_pre)
        for step in range(3):
            s = rng.random(n_pre) < 0.1
            out = ds.step(torch.from_numpy(s).to(dev))
            ref = O.binary_csrmv(w, idx, ptr, s, (n_pre, n_post), True)
            np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
            np.testing.assert_array_equal(ds.exchange.gather(torch.from_numpy(s).to(dev)).cpu().numpy(), s)
    # pipelined schedule: post step t + 1, consume step t
    ex = D.SpikeExchange(n_pre, packed=True, device=dev)
    spikes = [rng.random(n_pre) < 0.2 for _ in range(5)]
    ticket = ex.post(torch.from_numpy(spikes[0]).to(dev))
    for t in range(5):
        nxt = ex.post(torch.from_numpy(spikes[t + 1]).to(dev)) if t + 1 < 5 else None
        out = ex.wait_events(ticket) @ shard
        np.testing.assert_allclose(out.cpu().numpy(), O.binary_csrmv(w, idx, ptr, spikes[t], (n_pre, n_post), True),
                                   rtol=1e-5, atol=1e-5)
        ticket = nxt


def test_jitc_shard_behind_the_exchange(one_rank_group):
    import brainevent_amd as be
    from brainevent_amd import _dist as D
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(1)
    n = 4000
    M = be.JITCScalarR((np.float32(1.0), 0.02, 9), shape=(n, n), corder=True)
    ds = D.DistributedScatter(M.scatter_shard(1, 0), n, packed=True, device=dev)
    s = rng.random(n) < 0.1
    got = ds.step(torch.from_numpy(s).to(dev))
    np.testing.assert_array_equal(got.cpu().numpy(), be.BinaryArray(s) @ M)


def test_native_exchange_through_the_c_abi():
    """``be_exchange_*`` (RCCL loaded by the library, one-rank communicator — all a one-GPU box can run): the gathered words
    are the bit-packed full vector, consumed packed by the planned scatter; slices follow ``word_aligned_bounds``."""
    import brainevent_amd as be
    from brainevent_amd import _dist as D
    from brainevent_amd._csr import ScatterPlan
    from oracle import oracle_np as O
    rng = np.random.default_rng(5)
    dev = torch.device('cuda', 0)
    n_pre, n_post = 70001, 30011                        # n_pre not a multiple of 32: the last word is partial
    uid = D.NativeSpikeExchange.unique_id()
    assert len(uid) == 128
    ex = D.NativeSpikeExchange(n_pre, 1, 0, uid, device=dev)
    assert (ex.lo, ex.hi) == (0, n_pre)
    lens = rng.integers(0, 40, n_pre)
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    idx = rng.integers(0, n_post, ptr[-1]).astype(np.int32)
    w = rng.random(ptr[-1]).astype(np.float32)
    csr = be.CSR((torch.from_numpy(w).to(dev), torch.from_numpy(idx).to(dev), torch.from_numpy(ptr).to(dev)), shape=(n_pre, n_post))
    csr.buffers['scatter_plan'] = ScatterPlan.build(csr.data, csr.indices, csr.indptr, shape=(n_pre, n_post))
    for kind in ('bool', 'float'):
        s = rng.random(n_pre) < 0.05
        local = torch.from_numpy(s if kind == 'bool' else np.where(s, 1.5, -1.0).astype(np.float32)).to(dev)
        ev = ex.gather_events(local)
        np.testing.assert_array_equal(ev.value.cpu().numpy(), s)
        out = ev @ csr
        np.testing.assert_allclose(out.cpu().numpy(), O.binary_csrmv(w, idx, ptr, s, (n_pre, n_post), True), rtol=1e-5, atol=1e-5)
    # pipelined schedule on the exchange's own stream: post step t + 1, consume step t
    spikes = [rng.random(n_pre) < 0.1 for _ in range(4)]
    ticket = ex.post(torch.from_numpy(spikes[0]).to(dev))
    for t in range(4):
        nxt = ex.post(torch.from_numpy(spikes[t + 1]).to(dev)) if t + 1 < 4 else None
        out = ex.wait_events(ticket) @ csr
        np.testing.assert_allclose(out.cpu().numpy(), O.binary_csrmv(w, idx, ptr, spikes[t], (n_pre, n_post), True), rtol=1e-5, atol=1e-5)
        ticket = nxt
    ex.close()


def test_packed_producer_feeds_the_exchange_without_a_pack_launch(one_rank_group):
    """A step loop that keeps its spikes as words: `lif_coba_step(..., spike_bits=...)` writes the bit-packed spikes (equal to
    `be_pack_spikes` of its byte spikes), both exchanges take the packed slice as it is (`BE_SPIKE_BITS`: gathered from where it
    lies) and the scatter consumes the gathered words — same result as the byte path and as the oracle."""
    import brainevent_amd as be
    from brainevent_amd import _dist as D, _array as A
    from brainevent_amd._csr import ScatterPlan
    from oracle import oracle_np as O
    rng = np.random.default_rng(11)
    dev = torch.device('cuda', 0)
    for n in (70016, 70001, 37):                       # whole words, a partial last word, less than two words
        V = torch.from_numpy(rng.uniform(-62, -49.5, n).astype(np.float32)).to(dev)
        ge, gi = torch.rand(n, device=dev), torch.rand(n, device=dev)
        refr = torch.from_numpy(np.where(rng.random(n) < 0.2, 1.0, 0.0).astype(np.float32)).to(dev)
        inp = torch.rand(n, device=dev) * 30
        st = [t.clone() for t in (V, ge, gi, refr)]
        spk = torch.zeros(n, dtype=torch.bool, device=dev)
        bits = torch.full(((n + 31) // 32,), -1, dtype=torch.int32, device=dev)
        be.lif_coba_step(*st, inp, inp, spk, spike_bits=bits)
        s = spk.cpu().numpy()
        assert 0 < s.sum() < n
        np.testing.assert_array_equal(be.BitPackedBinary.from_packed(bits, n).value.cpu().numpy(), s)
        np.testing.assert_array_equal(bits.cpu().numpy(), be.BitPackedBinary(spk).packed[0].cpu().numpy().view(np.int32))
        st2 = [t.clone() for t in (V, ge, gi, refr)]
        bits2 = torch.zeros_like(bits)
        be.lif_coba_step(*st2, inp, inp, None, spike_bits=bits2)          # words only
        assert torch.equal(bits2, bits) and all(torch.equal(a, b) for a, b in zip(st, st2))
        n_post = 3001
        lens = rng.integers(0, 30, n)
        ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        idx = rng.integers(0, n_post, ptr[-1]).astype(np.int32)
        w = rng.random(ptr[-1]).astype(np.float32)
        csr = be.CSR((torch.from_numpy(w).to(dev), torch.from_numpy(idx).to(dev), torch.from_numpy(ptr).to(dev)), shape=(n, n_post))
        ref = O.binary_csrmv(w, idx, ptr, s, (n, n_post), True)
        local = be.BitPackedBinary.from_packed(bits, n)
        ex = D.NativeSpikeExchange(n, 1, 0, D.NativeSpikeExchange.unique_id(), device=dev)
        ev = ex.gather_events(local)
        np.testing.assert_array_equal(ev.value.cpu().numpy(), s)
        np.testing.assert_allclose((ev @ csr).cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
        ev = ex.wait_events(ex.post(A.PackedSpikes(bits, n)))
        np.testing.assert_allclose((ev @ csr).cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
        ex.close()
        tex = D.SpikeExchange(n, packed=True, device=dev)
        np.testing.assert_array_equal(tex.gather(local).cpu().numpy(), s)
        np.testing.assert_allclose((tex.wait_events(tex.post(local)) @ csr).cpu().numpy(), ref, rtol=1e-5, atol=1e-5)


def test_rank_step_fast_path_equals_the_operator_surface():
    """`RankStep(exchange, shard)(local)` issues the exchange and the planned / binned step as two C calls; it must give the bits of
    `exchange.gather_events(local) @ shard`, follow an in-place weight update (it falls back, the general path refreshes the
    workspace) and serve byte, float and packed local spikes."""
    import brainevent_amd as be
    from brainevent_amd import _dist as D, _array as A, _csr as C
    rng = np.random.default_rng(17)
    dev = torch.device('cuda', 0)
    n_pre, n_post = 40_000, 300_000
    lens = rng.integers(20, 60, n_pre)
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    idx = rng.integers(0, n_post, ptr[-1]).astype(np.int32)
    w = rng.random(ptr[-1]).astype(np.float32)
    ex = D.NativeSpikeExchange(n_pre, 1, 0, D.NativeSpikeExchange.unique_id(), device=dev)
    for route in ('plan', 'binned'):
        wd = torch.from_numpy(w.copy()).to(dev)
        csr = be.CSR((wd, torch.from_numpy(idx).to(dev), torch.from_numpy(ptr).to(dev)), shape=(n_pre, n_post))
        if route == 'plan':
            csr.buffers['scatter_plan'] = C.ScatterPlan.build(csr.data, csr.indices, csr.indptr, shape=(n_pre, n_post))
        else:
            csr.buffers['scatter_plan'] = C.BinnedScatter(csr.data, n_pre, n_post, int(ptr[-1]), indices=csr.indices)
        step = D.RankStep(ex, csr)
        assert step._fast is not None
        for kind in ('bool', 'float', 'words'):
            s = rng.random(n_pre) < 0.05
            if kind == 'bool':
                local = torch.from_numpy(s).to(dev)
            elif kind == 'float':
                local = torch.from_numpy(np.where(s, 2.0, 0.0).astype(np.float32)).to(dev)
            else:
                local = A.PackedSpikes(be.bitpack(torch.from_numpy(s).to(dev), 0).reshape(-1), n_pre)
            fast = step(local)
            slow = ex.gather_events(local) @ csr
            assert torch.equal(fast, slow), (route, kind)
        # the whole rank step (exchange + scatter) as one replayed HIP graph over a static local buffer: same bits
        static = torch.zeros(n_pre, dtype=torch.bool, device=dev)
        graphed = step.captured(static)
        for _ in range(3):
            s = rng.random(n_pre) < 0.05
            static.copy_(torch.from_numpy(s).to(dev))
            assert torch.equal(graphed(), step(static)), route
        del graphed
        # a caller that REBINDS shard.data (instead of updating it in place) must not get the old weights through the raw pointers
        old_data = csr.data
        csr.data = (old_data * 3.0).contiguous()
        assert torch.equal(step(static), ex.gather_events(static) @ csr), route
        csr.data = old_data
        csr.buffers['scatter_plan'] = C.fresh_scatter_workspace(csr.buffers['scatter_plan'], csr.data, csr.indices, csr.indptr)
        step = D.RankStep(ex, csr)
        csr.data.mul_(0.5)                                   # in-place update: the fast path notices and hands over
        local = torch.from_numpy(s).to(dev)
        got = step(local)
        from oracle import oracle_np as O
        np.testing.assert_allclose(got.cpu().numpy(), O.binary_csrmv(w * 0.5, idx, ptr, s, (n_pre, n_post), True), rtol=1e-5, atol=1e-5)
        assert torch.equal(step(local), got)                 # ... and is fast again on the refreshed workspace, same bits
    # anything else takes the general path
    dense_like = be.CSR((torch.ones(1, device=dev), torch.from_numpy(idx[:100]).to(dev), torch.tensor([0, 100], dtype=torch.int32, device=dev)),
                        shape=(1, n_post))
    assert D.RankStep(ex, dense_like)._fast is None
    ex.close()


@pytest.mark.parametrize('route', ['plan', 'binned'])
def test_rank_step_pipelined_schedule_is_bit_identical_to_the_sequential_one(one_rank_group, route):
    """`RankStep.post / step_posted / ahead / drain` — the exchange of step t + 1 posted on the exchange's own stream before step t
    is scattered (`bench.py --exchange-ahead 1`) — against `RankStep.__call__` (exchange, then scatter, on one stream): the same
    bits for every step, on a planned and on a binned shard, with byte and packed-word producers; `exchange_only` + `scatter_only`
    are the two halves of the sequential step."""
    import brainevent_amd as be
    from brainevent_amd import _dist as D, _array as A
    from brainevent_amd import _csr as C
    rng = np.random.default_rng(21)
    dev = torch.device('cuda', 0)
    n_pre, n_post = (40000, 3000) if route == 'plan' else (30000, 400_000)
    uid = D.NativeSpikeExchange.unique_id()
    ex = D.NativeSpikeExchange(n_pre, 1, 0, uid, device=dev)
    lens = rng.integers(20, 60, n_pre)
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    idx = torch.from_numpy(rng.integers(0, n_post, ptr[-1]).astype(np.int32)).to(dev)
    w = torch.from_numpy(rng.uniform(0.1, 1.0, ptr[-1]).astype(np.float32)).to(dev)
    tptr = torch.from_numpy(ptr).to(dev)
    csr = be.CSR((w, idx, tptr), shape=(n_pre, n_post))
    if route == 'plan':
        csr.buffers['scatter_plan'] = C.ScatterPlan.build(w, idx, tptr, shape=(n_pre, n_post))
    else:
        csr.buffers['scatter_plan'] = C.BinnedScatter(w, n_pre, n_post, int(ptr[-1]), indices=idx, indptr=tptr)
    rs = D.RankStep(ex, csr)
    assert rs._fast is not None
    spikes = [torch.from_numpy(rng.random(n_pre) < 0.05).to(dev) for _ in range(6)]
    seq = [rs(s).clone() for s in spikes]
    for producer in ('bytes', 'words'):
        loc = spikes if producer == 'bytes' else [A.PackedSpikes(be.bitpack(s, 0).reshape(-1), n_pre) for s in spikes]
        rs.post(loc[0])
        got = [rs.ahead(loc[t + 1]) if t + 1 < 6 else rs.step_posted() for t in range(6)]
        rs.drain()
        for t in range(6):
            assert torch.equal(got[t], seq[t]), (producer, t)
        # two exchanges in flight (both buffers of the exchange), consumed in order
        rs.post(loc[2]); rs.post(loc[3])
        assert torch.equal(rs.step_posted(), seq[2]) and torch.equal(rs.step_posted(), seq[3])
        # the exchange's stream also compacts what it gathered (be_exchange_post_ids) and the scatter takes that list
        # (BE_SPIKE_IDS: no spike-list build on the scattering stream): the same bits, mixed freely with plain posts
        rs.post(loc[0], ids=True)
        got = [rs.ahead(loc[t + 1], ids=(t % 3 != 1)) if t + 1 < 6 else rs.step_posted() for t in range(6)]
        rs.drain()
        for t in range(6):
            assert torch.equal(got[t], seq[t]), (producer, 'ids', t)
    rs.exchange_only(spikes[4])
    assert torch.equal(rs.scatter_only(), seq[4])
    rs.check_status()                                 # binned shard: sticky flag + conservation counters; planned shard: nothing to check
    ref = (be.BinaryArray(spikes[4]) @ csr)
    assert torch.equal(ref, seq[4])
    ex.close()


def test_pipelined_exchange_is_correct_when_no_queue_of_its_own_exists():
    """The exchange's side stream is probed onto a hardware queue of its own (be_exchange.hip: exchange_pick_side_stream).  With ONE
    hardware queue for the whole process (GPU_MAX_HW_QUEUES=1) no candidate can overlap: the probe keeps the best it saw and the
    pipelined schedule — posts, waits, the id lists — is still bit-identical to the sequential one (run in a child process: the
    queue count is read when the runtime starts).  BE_EXCHANGE_PROBE=0 (no probe at all) likewise."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in ({'GPU_MAX_HW_QUEUES': '1'}, {'BE_EXCHANGE_PROBE': '0'}):
        r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(root, 'tests', 'test_dist_gpu.py'), '-q', '-x', '-k',
                            'pipelined_schedule_is_bit_identical'], env=dict(os.environ, **extra), capture_output=True, text=True,
                           timeout=600, cwd=root)
        assert r.returncode == 0 and '2 passed' in r.stdout, (extra, r.stdout[-1500:], r.stderr[-1500:])
