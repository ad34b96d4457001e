#!/usr/bin/env python3
"""Dense event-driven products: achieved bandwidth over the bytes each variant has to touch."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
dev = torch.device('cuda', 0)


def timeit(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for dtype, n, nb in ((torch.float32, 32768, 1), (torch.float16, 65536, 1), (torch.float16, 65536, 32), (torch.float32, 32768, 8)):
    W = torch.empty((n, n), dtype=dtype, device=dev).normal_()
    esz = W.element_size()
    for p in (0.01, 0.5):
        s1 = torch.rand(n, device=dev) < p
        S = torch.rand((nb, n), device=dev) < p
        if nb == 1:
            tT = timeit(lambda: be.BinaryArray(s1) @ W)
            tN = timeit(lambda: W @ be.BinaryArray(s1))
            act = int(s1.sum())
            print(f'{dtype} n={n} p={p}: spk@W {tT*1e3:.3f} ms ({act*n*esz/tT/1e9:.0f} GB/s of active rows) | '
                  f'W@spk {tN*1e3:.3f} ms ({n*n*esz/tN/1e9:.0f} GB/s if fully streamed)', flush=True)
        else:
            tT = timeit(lambda: be.BinaryArray(S) @ W)
            tN = timeit(lambda: W @ be.BinaryArray(S.T.contiguous()))
            union = int(S.any(dim=0).sum())
            print(f'{dtype} n={n} B={nb} p={p}: S@W {tT*1e3:.3f} ms ({union*n*esz/tT/1e9:.0f} GB/s of union rows) | '
                  f'W@S.T {tN*1e3:.3f} ms ({n*n*esz/tN/1e9:.0f} GB/s if fully streamed)', flush=True)
    del W
    torch.cuda.empty_cache()
