"""Stress (round 4): the binned route's append protocol under repetition — the same step over ragged rows thousands of times;
fixed-point sums do not depend on the order of arrival, so every repetition must reproduce the first result bit for bit (a lost or
duplicated entry of the write-combining rings would show)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
from brainevent_amd import _csr as C

dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(2)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for m, k, mean_len, fire in ((2_000_000, 1_250_000, 125, 0.01), (1_000_000, 10_000_000, 1000, 0.01)):
    lens = torch.poisson(torch.full((m,), float(mean_len), device=dev), generator=g).to(torch.int64)
    ptr = torch.zeros(m + 1, dtype=torch.int64, device=dev); torch.cumsum(lens, 0, out=ptr[1:])
    nnz = int(ptr[-1])
    idx = torch.randint(0, k, (nnz,), dtype=torch.int32, device=dev, generator=g)
    w = torch.rand(nnz, device=dev, generator=g)
    ws = C.BinnedScatter(w, m, k, nnz, indices=idx, indptr=ptr)
    spikes = [torch.rand(m, device=dev, generator=g) < fire for _ in range(4)]
    first = [be.binary_csrmv(w, idx, ptr, s, shape=(m, k), transpose=True, workspace=ws).clone() for s in spikes]
    bad = 0
    for it in range(reps):
        out = be.binary_csrmv(w, idx, ptr, spikes[it % 4], shape=(m, k), transpose=True, workspace=ws)
        if not torch.equal(out, first[it % 4]):
            bad += 1
            d = torch.nonzero(out != first[it % 4]).flatten()
            print(f'  rep {it}: {d.numel()} outputs differ, first {d[:5].tolist()}: {out[d[:5]].tolist()} vs {first[it % 4][d[:5]].tolist()}', flush=True)
    ws.check_status()
    print(f'm={m} k={k} rows of ~{mean_len} ({nnz:.2e} entries, {ws.n_slices} bins, kind {ws.kind}): {reps} repetitions, {bad} differing', flush=True)
    del ws, w, idx, ptr, lens, first
    torch.cuda.empty_cache()
