"""Diagnostic (round 4): `FixedNumPerPre @ spk` through the C4 mirror against the gather kernel, many builds and steps in one
process — which outputs differ when they do, and whether the mirror's arrays or the step is at fault."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
from bench import gen_fixed_num_on_device

n, K = 10_000_000, 1000
builds, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(12)
w, idx = gen_fixed_num_on_device(n, K, n, False, dev, g)
wsum = int(w.view(torch.int32).reshape(-1).sum(dtype=torch.int64))
# per-pre-neuron checksums of the raw matrix: sum of the weight bit patterns of row r (what column r of the mirror must hold)
row_ck = w.view(torch.int32).sum(dim=1, dtype=torch.int64)
fails = 0
for b in range(builds):
    conn = be.FixedNumPerPre((w, idx), shape=(n, n), check_indices=False)
    mr = conn.build_mirror()
    ws = mr.plan
    ok_w = int(mr.data.view(torch.int32).sum(dtype=torch.int64)) == wsum
    ck = torch.zeros(n, dtype=torch.int64, device=dev)
    for lo in range(0, n * K, 1 << 29):                          # by pre neuron (the mirror's column), half a billion entries at a time
        hi = min(n * K, lo + (1 << 29))
        ck.index_add_(0, mr.indices[lo:hi].to(torch.int64), mr.data.view(torch.int32)[lo:hi].to(torch.int64))
    ok_rows = bool(torch.equal(ck, row_ck))
    print(f'build {b}: weights multiset {ok_w}, per-row checksums {ok_rows}, kind {ws.kind} exp {ws.scale_exp} bins {ws.n_slices}', flush=True)
    del ck
    for s in range(steps):
        spk = torch.rand(n, device=dev, generator=g) < 0.01
        out = conn @ be.BinaryArray(spk)
        ref = be.binary_fcnmv(w, idx, spk, shape=(n, n), transpose=False)
        rel = (out.double() - ref.double()).abs() / ref.double().abs().clamp_min(1e-30)
        if float(rel.max()) > 1e-5:
            fails += 1
            bad = torch.nonzero(rel > 1e-5).flatten()
            q = bad[:10]
            out2 = conn @ be.BinaryArray(spk)
            ref2 = be.binary_fcnmv(w, idx, spk, shape=(n, n), transpose=False)
            print(f'  build {b} step {s}: {bad.numel()} outputs off, max rel {float(rel.max()):.6e}; ids {q.tolist()}\n   out  {out[q].tolist()}\n   ref  {ref[q].tolist()}'
                  f'\n   out2 {out2[q].tolist()}\n   ref2 {ref2[q].tolist()}\n   out == out2 {bool(torch.equal(out, out2))}, ref == ref2 {bool(torch.equal(ref, ref2))}', flush=True)
            ws.check_status()
    print(f'build {b}: {steps} steps done, {fails} failing so far', flush=True)
    del conn, mr, ws
    torch.cuda.synchronize(); torch.cuda.empty_cache()
print('failures:', fails)
