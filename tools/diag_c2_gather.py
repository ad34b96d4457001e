"""Staged diagnostic of the C2-size gather direction (every stage synchronises and logs before the next starts)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import brainevent_amd as be
from bench import gen_csr_on_device

def log(*a):
    print(*a, flush=True)

stage = sys.argv[1] if len(sys.argv) > 1 else 'gather'
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
dev = torch.device('cuda', 0)
n, n_conn = 1_000_000, 10_000
t = time.time()
w, idx, ptr = gen_csr_on_device(rows, n, n_conn, False, 78, dev)
torch.cuda.synchronize(); log('generated', idx.numel(), ptr.dtype, f'{time.time() - t:.1f}s')
g = torch.Generator(device=dev); g.manual_seed(6)
spk = torch.rand(n, device=dev, generator=g) < 0.01
if stage in ('gather', 'both'):
    ref = be.binary_csrmv(w, idx, ptr, spk, shape=(rows, n), transpose=False)
    torch.cuda.synchronize(); log('gather kernel ok', float(ref.sum()))
if stage in ('mirror', 'both'):
    M = be.CSR((w, idx, ptr), shape=(rows, n), check_structure=False)
    t = time.time(); mr = M.build_mirror(); torch.cuda.synchronize(); log('mirror built', f'{time.time() - t:.2f}s', mr.released, type(mr.plan).__name__)
    out = M @ be.BinaryArray(spk); torch.cuda.synchronize(); log('mirror step ok', float(out.sum()))
    if stage == 'both':
        rel = ((out.double() - ref.double()).abs() / ref.double().abs().clamp_min(1e-30)).max().item()
        log('rel', rel)
