"""Stress (round 4): the gather kernel's coarse-bitmap path (more input columns than the LDS bitmap holds: what the C4 mirror test's
reference runs) against two independent evaluations of the same product — the float twin on a 0 / 1 operand and the scatter over
the transposed problem — many spike vectors, one shared weight (every result an exact integer multiple)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be

dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(1)
m, k, K = 200_000, 3_000_000, 1000
idx = torch.randint(0, k, (m, K), dtype=torch.int32, device=dev, generator=g)
w = torch.ones(1, device=dev)
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
    spk = torch.rand(k, device=dev, generator=g) < 0.01
    a = be.binary_fcnmv(w, idx, spk, shape=(m, k), transpose=False)            # gather kernel, coarse bitmap
    b = be.fcnmv(w, idx, spk.float(), shape=(m, k), transpose=False)           # float twin: sum of x[idx]
    if not torch.equal(a, b):
        bad += 1
        d = torch.nonzero(a != b).flatten()
        print(f'iter {it}: {d.numel()} rows differ; first {d[:6].tolist()} gather {a[d[:6]].tolist()} twin {b[d[:6]].tolist()}', flush=True)
print('mismatching iterations:', bad)
