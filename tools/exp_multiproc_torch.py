"""Platform check (round 4): N processes sharing the one GPU of the box, torch only (no brainevent_amd): masked selects with their
device-to-host syncs, as the bench's shard generator does."""
import os, sys, time, torch
import torch.distributed as dist
dist.init_process_group('gloo')
r = dist.get_rank()
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
t0 = time.time()
g = torch.Generator(device=dev); g.manual_seed(0)
for i in range(4):
    blk = torch.randint(0, 200000, (100_000_000,), dtype=torch.int32, device=dev, generator=g)
    keep = (blk >= 10) & (blk < 50000)
    sel = blk[keep]
    print(f'rank {r} iter {i}: {sel.numel()} kept, {time.time() - t0:.1f} s', flush=True)
dist.barrier()
print(f'rank {r} done', flush=True)
