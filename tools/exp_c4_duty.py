"""Experiment (round 4): the C4 step takes 0.59 ms in a back-to-back loop and 0.52 ms under rocprofv3 --kernel-trace (same kernels:
pass B 354 instead of ~410 us).  Is it the duty cycle?  Per-step GPU time (event pairs) back to back, and with the device left
idle for 0.5 / 2 / 10 ms between steps."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
from bench import gen_fixed_num_on_device

n, K = 10_000_000, 1000
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(3)
w, idx = gen_fixed_num_on_device(n, K, n, False, dev, g)
conn = be.FixedNumPerPre((w, idx), shape=(n, n), check_indices=False).prepare()
spk = [be.BinaryArray(torch.rand(n, device=dev, generator=g) < 0.01) for _ in range(20)]
for i in range(30):
    spk[i % 20] @ conn
torch.cuda.synchronize()


def run(gap_s, steps=80):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for i in range(steps):
        ev[i][0].record()
        spk[i % 20] @ conn
        ev[i][1].record()
        if gap_s:
            torch.cuda.synchronize()
            time.sleep(gap_s)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[len(ms) // 2], ms[len(ms) // 10], wall / steps * 1e3


for gap in (0.0, 0.0005, 0.002, 0.010, 0.0):
    med, p10, wall = run(gap)
    print(f'idle {gap * 1e3:5.1f} ms between steps: GPU time per step median {med:.3f} ms, p10 {p10:.3f}; wall per step {wall:.3f} ms', flush=True)
