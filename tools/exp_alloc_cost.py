import torch, time
dev = torch.device('cuda', 0)
torch.zeros(1, device=dev); torch.cuda.synchronize()
for gb in (1, 20, 58, 20, 58):
    t0 = time.perf_counter(); x = torch.empty(int(gb * 1e9), dtype=torch.uint8, device=dev); torch.cuda.synchronize(); t1 = time.perf_counter()
    x.fill_(1); torch.cuda.synchronize(); t2 = time.perf_counter()
    x.fill_(2); torch.cuda.synchronize(); t3 = time.perf_counter()
    del x; torch.cuda.synchronize(); t4 = time.perf_counter()
    torch.cuda.empty_cache(); torch.cuda.synchronize(); t5 = time.perf_counter()
    print(f'{gb} GB: alloc {1e3*(t1-t0):.0f} ms, first fill {1e3*(t2-t1):.0f} ms, second fill {1e3*(t3-t2):.0f} ms, del {1e3*(t4-t3):.0f} ms, empty_cache {1e3*(t5-t4):.0f} ms', flush=True)
