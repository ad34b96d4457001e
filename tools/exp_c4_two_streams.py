#!/usr/bin/env python3
"""VERDICT r4 task 3 (i): does overlapping pass C of step t with pass B of step t + 1 pay?

Two binned workspaces over the same C4 matrix, consecutive steps issued alternately on two HIP streams (step t on stream t % 2 with
workspace t % 2: its own regions, directory, output), so that a step's pass C and the next step's compaction + pass B are free to run
side by side — valid for a caller whose synapses have >= 1 step of delay, and for the rows of a csrmm batch.  Reports the sustained
time per step against the same steps on one stream.  usage: python tools/exp_c4_two_streams.py [--homo] [--n N] [--k K]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainevent_amd as be
from brainevent_amd import _csr as C
from bench import gen_fixed_num_on_device


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--homo', action='store_true')
    ap.add_argument('--n', type=int, default=10_000_000)
    ap.add_argument('--k', type=int, default=1000)
    ap.add_argument('--n-post', type=int, default=0)
    ap.add_argument('--steps', type=int, default=60)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev); g.manual_seed(7)
    n, K, n_post = a.n, a.k, a.n_post or a.n
    w, idx = gen_fixed_num_on_device(n, K, n_post, a.homo, dev, g)
    conns = []
    for _ in range(2):
        c = be.FixedNumPerPre((w, idx), shape=(n, n_post), check_indices=False)
        c.buffers['scatter_plan'] = C.BinnedScatter(w.reshape(-1), n, n_post, n * K, indices=idx.reshape(-1), row_len=K)
        conns.append(c)
    spikes = [be.BinaryArray(torch.rand(n, device=dev, generator=g) < 0.01) for _ in range(20)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()

    def run(two):
        outs = [None, None]
        for rep in range(2):              # first repetition warms up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(a.steps):
                s = i % 2 if two else 0
                with torch.cuda.stream(streams[s]):
                    outs[s] = spikes[i % 20] @ conns[s]
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
        return el / a.steps * 1e3, outs

    one, o1 = run(False)
    two, o2 = run(True)
    ref = spikes[(a.steps - 1) % 20] @ conns[0]
    same = bool(torch.equal(o2[(a.steps - 1) % 2], ref))
    for c in conns:
        c.buffers['scatter_plan'].check_status()
    print(f"C4 {'homo' if a.homo else 'hetero'} n={n} K={K} n_post={n_post}: one stream {one:.4f} ms/step, two streams (steps alternate) "
          f"{two:.4f} ms/step sustained ({one / two:.3f} x), last output equal to the one-stream step: {same}")


if __name__ == '__main__':
    main()
