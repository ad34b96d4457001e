"""Experiment (round 4): the C4 step differs by up to 20 % between processes on the same box (pass B 345 ... 415 us).  Does the
placement of the workspace (the regions pass B writes) or of the matrix decide it?  One process: the same matrix with the workspace
re-allocated behind dummy allocations of different sizes, then the matrix itself re-allocated."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
from brainevent_amd import _csr as C
from bench import gen_fixed_num_on_device

n, K = 10_000_000, 1000
dev = torch.device('cuda', 0)


def step_ms(conn, spk, steps=100):
    for i in range(20):
        spk[i % len(spk)] @ conn
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        spk[i % len(spk)] @ conn
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


g = torch.Generator(device=dev); g.manual_seed(3)
spk = [be.BinaryArray(torch.rand(n, device=dev, generator=g) < 0.01) for _ in range(20)]
keep = []
for trial in range(3):
    g.manual_seed(3)
    w, idx = gen_fixed_num_on_device(n, K, n, False, dev, g)
    print(f'matrix allocation {trial}: weights at 0x{w.data_ptr():x}, indices at 0x{idx.data_ptr():x}', flush=True)
    for shift_mb in (0, 3, 130, 1000):
        dummy = torch.empty(shift_mb << 20, dtype=torch.uint8, device=dev) if shift_mb else None
        ws = C.BinnedScatter(w, n, n, n * K, indices=idx, row_len=K)
        conn = be.FixedNumPerPre((w, idx), shape=(n, n), check_indices=False)
        conn.buffers['scatter_plan'] = ws
        print(f'   workspace at 0x{ws.ws.data_ptr():x} ({ws.ws.numel() >> 20} MiB, behind a {shift_mb} MiB dummy): {step_ms(conn, spk):.3f} ms per step', flush=True)
        del conn, ws, dummy
        torch.cuda.empty_cache()
    keep.append(torch.empty((17 + 29 * trial) << 20, dtype=torch.uint8, device=dev))       # shifts the next matrix
    del w, idx
    torch.cuda.empty_cache()
