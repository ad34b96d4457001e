"""world_size-2 gloo tests (CPU) of the multi-GPU path: post-slice partition + spike all-gather.
The local product is done by the oracle here (no GPU in this container); on MI355X the same DistributedScatter
runs the HIP scatter on every rank (bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, packed, n_pre, n_post, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from brainevent_amd import _dist as D
        from oracle import oracle_np as O
        rng = np.random.default_rng(0)                      # same matrix on every rank
        lens = rng.integers(0, 40, n_pre)
        indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        indices = rng.integers(0, n_post, indptr[-1]).astype(np.int32)
        w = rng.random(indptr[-1]).astype(np.float32)
        sw, si, sp, sshape = D.shard_csr_by_post(torch.from_numpy(w), torch.from_numpy(indices), torch.from_numpy(indptr),
                                                 (n_pre, n_post), world, rank)
        lo, hi = D.post_slice_bounds(n_post, world, rank)
        assert sshape == (n_pre, hi - lo) and int(sp[-1]) == si.numel()
        assert si.numel() == 0 or (int(si.min()) >= 0 and int(si.max()) < hi - lo)

        def matmul(full, shard):
            d, i, p, shp = shard
            return O.binary_csrmv(d.numpy(), i.numpy(), p.numpy(), full.numpy(), shp, True)

        ds = D.DistributedScatter((sw, si, sp, sshape), n_pre, packed=packed, matmul=matmul)
        outs = []
        for step in range(3):
            full_ref = np.random.default_rng(100 + step).random(n_pre) < 0.2     # what the gathered vector must be
            plo, phi = ds.exchange.lo, ds.exchange.hi
            local = torch.from_numpy(full_ref[plo:phi].copy())
            got_full = ds.exchange.gather(local)
            assert np.array_equal(got_full.numpy(), full_ref)
            outs.append(ds.step(local))
        if packed:      # pipelined schedule: post step t+1, then consume step t
            refs = [np.random.default_rng(500 + t).random(n_pre) < 0.3 for t in range(5)]
            plo, phi = ds.exchange.lo, ds.exchange.hi
            ticket = ds.exchange.post(torch.from_numpy(refs[0][plo:phi].copy()))
            for t in range(5):
                nxt = ds.exchange.post(torch.from_numpy(refs[t + 1][plo:phi].copy())) if t + 1 < 5 else None
                got = ds.exchange.wait_events(ticket).value
                assert np.array_equal(got.numpy(), refs[t]), f"pipelined exchange, step {t}"
                ticket = nxt
            # a producer that emits its slice as words (PackedSpikes / packed-only BitPackedBinary): gathered as they are
            from brainevent_amd import _array as A
            from brainevent_amd._event import BitPackedBinary
            ref = np.random.default_rng(900).random(n_pre) < 0.4
            loc = ref[plo:phi]
            words = torch.from_numpy(np.packbits(np.pad(loc, (0, (-loc.size) % 32)), bitorder='little').view(np.int32).copy())
            assert np.array_equal(ds.exchange.gather(A.PackedSpikes(words, phi - plo)).numpy(), ref)
            if phi > plo:
                got = ds.exchange.wait_events(ds.exchange.post(BitPackedBinary.from_packed(words, phi - plo))).value
                assert np.array_equal(got.numpy(), ref)
            else:
                ds.exchange.wait_events(ds.exchange.post(A.PackedSpikes(words, 0)))
        q.put((rank, lo, hi, np.stack(outs)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('packed,n_pre,n_post', [(False, 64, 50), (False, 67, 53), (True, 64, 53), (True, 67, 50), (True, 20, 9),
                                                 (True, 200, 31)])
def test_post_sliced_scatter_world2(packed, n_pre, n_post):
    _run_world(2, packed, n_pre, n_post)


def test_post_sliced_scatter_world8():
    """Eight ranks (the driver's node size) under gloo: word-aligned pre slices with uneven and empty owners."""
    _run_world(8, True, 1000, 203)


def _run_world(world, packed, n_pre, n_post):
    from oracle import oracle_np as O
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, packed, n_pre, n_post, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert res[0][1] == 0 and res[-1][2] == n_post and all(res[i][2] == res[i + 1][1] for i in range(world - 1))   # slices tile
    rng = np.random.default_rng(0)
    lens = rng.integers(0, 40, n_pre)
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    indices = rng.integers(0, n_post, indptr[-1]).astype(np.int32)
    w = rng.random(indptr[-1]).astype(np.float32)
    for step in range(3):
        spk = np.random.default_rng(100 + step).random(n_pre) < 0.2
        ref = O.binary_csrmv(w, indices, indptr, spk, (n_pre, n_post), True)
        got = np.concatenate([r[3][step] for r in res])
        np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-6)


def test_slice_bounds_and_fixed_num_shards():
    from brainevent_amd import _dist as D
    for n, w in ((10, 3), (1_000_000, 8), (7, 8), (64, 2), (65, 2), (1, 4)):
        b = [D.word_aligned_bounds(n, w, r) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        assert all(x[0] % 32 == 0 or x[0] == n for x in b) and len({(x[1] - x[0] + 31) // 32 for x in b if x[1] - x[0] == b[0][1]}) == 1
    for n, w in ((10, 3), (1_000_000, 8), (7, 8)):
        b = [D.post_slice_bounds(n, w, r) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        assert max(x[1] - x[0] for x in b) - min(x[1] - x[0] for x in b) <= 1
    rng = np.random.default_rng(1)
    n_pre, n_post, K = 20, 37, 6
    idx = torch.from_numpy(rng.integers(0, n_post, (n_pre, K)).astype(np.int32))
    data = torch.from_numpy(rng.random((n_pre, K)).astype(np.float32))
    dense = np.zeros((n_pre, n_post), np.float32)
    np.add.at(dense, (np.repeat(np.arange(n_pre), K), idx.numpy().reshape(-1)), data.numpy().reshape(-1))
    cols = []
    for r in range(3):
        d, i, p, shp = D.shard_fixed_num_by_post(data, idx, (n_pre, n_post), 3, r)
        # the blocked implementation gives the same shard whatever the block size (here: 1, 2 and many rows per block)
        for be_ in (1, 13, 50):
            flat_ptr = torch.arange(n_pre + 1, dtype=torch.int64) * K
            d2, i2, p2, shp2 = D.shard_csr_by_post(data.reshape(-1), idx.reshape(-1), flat_ptr, (n_pre, n_post), 3, r,
                                                   block_entries=be_)
            assert shp2 == shp and torch.equal(i2, i) and torch.equal(p2.long(), p.long()) and torch.equal(d2, d)
        blk = np.zeros(shp, np.float32)
        np.add.at(blk, (np.repeat(np.arange(n_pre), np.diff(p.numpy())), i.numpy()), d.numpy())
        cols.append(blk)
    np.testing.assert_allclose(np.concatenate(cols, axis=1), dense, rtol=1e-6)
    bits = torch.from_numpy(rng.random(29) < 0.5)
    assert torch.equal(D._unpack_bits(D._pack_bits(bits), 29), bits)


def test_jit_walk_class_columns_tile_the_output():
    """Geometry of the JITC multi-GPU partition: the columns of the (chunk, lane) classes tile the output, and the
    oracle's edge stream of a class stays inside that class's columns."""
    from brainevent_amd._jitc import jit_scatter_class_columns
    from brainevent_amd._dist import post_slice_bounds
    from oracle import oracle_np as O
    for shape1, out_len in ((17, 17), (1000, 1000), (30, 100), (4_000, 1_003), (3, 50)):
        chunk = max(1, (shape1 + 3) // 4)
        n_cls = ((out_len + chunk - 1) // chunk) * 32
        for world in (1, 2, 8):
            seen = np.zeros(out_len, np.int32)
            for r in range(world):
                lo, hi = post_slice_bounds(n_cls, world, r)
                cols = jit_scatter_class_columns(shape1, out_len, lo, hi)
                assert cols.size == 0 or (cols.min() >= 0 and cols.max() < out_len)
                seen[cols] += 1
            assert (seen == 1).all()
    seed, clen, n_cols = 123, 10, 100
    chunk = max(1, (n_cols + 3) // 4)
    n_edges = 0
    for c in range(((n_cols + chunk - 1) // chunk) * 32):
        ch, lane = divmod(c, 32)
        cs, ce = ch * chunk, min((ch + 1) * chunk, n_cols)
        cols = set(jit_scatter_class_columns(n_cols, n_cols, c, c + 1).tolist())
        q, state = O.lr_initial_q(O.lr_init(seed, 5, ch, lane), clen)
        while cs + lane + 32 * q < ce:           # the whole stream of (row 5, chunk, lane) stays inside the class's columns
            assert cs + lane + 32 * q in cols
            n_edges += 1
            state = O.lr_next(state)
            q = q + 1 + O.lr_bounded(state, clen - 1)
    assert n_edges > 0


def test_native_exchange_slice_arithmetic_matches_the_python_partition():
    """ranks > 0 of the library's own exchange binding had never executed.  Its partition is a pure function now
    (``be_exchange_slice_for``: what init / slice / allgather / post all go through); here it is checked for worlds 1 ... 8,
    even, uneven and degenerate populations (owners of nothing) against ``word_aligned_bounds`` and against the invariants
    the bit-packed gather rests on: the slices tile [0, n), each starts on a word boundary, every rank contributes the same
    number of words, and the words gathered are exactly ceil-padded."""
    import ctypes
    from brainevent_amd import _lib
    from brainevent_amd._dist import word_aligned_bounds
    f = _lib.fn('be_exchange_slice_for', ctypes.c_int,
                [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                 ctypes.POINTER(ctypes.c_int64)])
    for world in range(1, 9):
        for n in (0, 1, 31, 32, 33, 255, 256, 257, 1000, 4096, 1_000_000, 10_000_000, 10_000_019, 32 * world, 32 * world + 1,
                  32 * (world - 1) + 5):
            cover, wprs = 0, set()
            for rank in range(world):
                lo, hi, wpr = ctypes.c_int64(-1), ctypes.c_int64(-1), ctypes.c_int64(-1)
                assert f(n, world, rank, ctypes.byref(lo), ctypes.byref(hi), ctypes.byref(wpr)) == 0
                assert (lo.value, hi.value) == word_aligned_bounds(n, world, rank), (n, world, rank)
                assert lo.value == cover and lo.value <= hi.value <= n            # tiles in rank order, may be empty
                assert lo.value % 32 == 0 or lo.value == n
                assert hi.value - lo.value <= wpr.value * 32
                cover = hi.value
                wprs.add(wpr.value)
            assert cover == n and len(wprs) == 1
            wpr = wprs.pop()
            assert wpr * world * 32 >= n and (wpr - 1) * world * 32 < max(n, 1) + 32 * world
    assert f(10, 0, 0, None, None, None) < 0 and f(10, 2, 2, None, None, None) < 0 and f(-1, 1, 0, None, None, None) < 0
    assert f(10, 2, 1, None, None, None) == 0                                     # outputs are optional
