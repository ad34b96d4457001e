"""Does a buffer written by one kernel and read by the next stay in the 256 MiB Infinity Cache?  (sizing question for a
chunked two-pass scatter: pass B writes binned entries, pass C reads them back)"""
import torch, time
dev = torch.device('cuda', 0)
def ev_time(f, reps=20):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        pre(); s.record(); f(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    ts.sort(); return ts[len(ts) // 2]
big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)       # evicts everything when written
for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024):
    n = (mb << 20) // 4
    x = torch.empty(n, dtype=torch.float32, device=dev); y = torch.empty_like(x)
    res = {}
    for name, prep in (('after write of x', lambda: x.fill_(1.0)), ('after 512 MB of other writes', lambda: (x.fill_(1.0), big.fill_(0))),
                       ('after read of x', lambda: x.sum())):
        pre = prep
        t = ev_time(lambda: x.sum())
        res[name] = mb / 1024 / (t * 1e-3) / 1e3
    pre = lambda: None
    tw = ev_time(lambda: x.fill_(2.0)); tc = ev_time(lambda: y.copy_(x))
    print(f'{mb:5d} MB: read ' + ' | '.join(f'{k} {v:.2f} TB/s' for k, v in res.items()) + f' | fill {mb/1024/(tw*1e-3)/1e3:.2f} TB/s | copy(r+w) {2*mb/1024/(tc*1e-3)/1e3:.2f} TB/s', flush=True)
    del x, y
