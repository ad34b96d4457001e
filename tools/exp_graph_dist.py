"""Does the per-rank step (bit-pack -> RCCL all-gather -> planned scatter) capture into one HIP graph?  One-rank group."""
import os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29544')
dev = torch.device('cuda', 0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
import brainevent_amd as be
from brainevent_amd import _csr as C
from brainevent_amd._dist import SpikeExchange
from bench import gen_csr_shard_on_device
n = 1_000_000
w, idx, ptr, shape, _ = gen_csr_shard_on_device(n, n, 10000, '--homo' in sys.argv, 1234, dev, 8, 0)
csr = be.CSR((w, idx, ptr), shape=shape, check_structure=False).prepare()
ex = SpikeExchange(n, packed=True, device=dev)
g = torch.Generator(device=dev); g.manual_seed(1)
spikes = torch.rand((16, n), device=dev, generator=g) < 0.01
static = torch.zeros(n, dtype=torch.bool, device=dev)

def step():
    return ex.gather_events(static) @ csr

for i in range(5):
    static.copy_(spikes[i]); ref = step().clone()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(200):
    static.copy_(spikes[i % 16]); out = step()
torch.cuda.synchronize()
print('eager  us/step', (time.perf_counter() - t0) / 200 * 1e6, flush=True)
gs = be.capture_step(step)
print('captured', flush=True)
static.copy_(spikes[4]); o = gs(); torch.cuda.synchronize()
print('replay equals eager:', torch.equal(o, ref), flush=True)
t0 = time.perf_counter()
for i in range(200):
    static.copy_(spikes[i % 16]); out = gs()
torch.cuda.synchronize()
print('graph  us/step', (time.perf_counter() - t0) / 200 * 1e6, flush=True)
dist.destroy_process_group()
