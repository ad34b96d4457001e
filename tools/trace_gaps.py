#!/usr/bin/env python3
"""Per-step timeline from a rocprofv3 kernel trace CSV: kernel durations and the gaps between consecutive kernels.
usage: python tools/trace_gaps.py <p_kernel_trace.csv> [first_kernel_substring]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else 'k_pack_spikes'
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# take the last 40 steps: a step starts at a kernel whose name contains `first`
idx = [i for i, r in enumerate(rows) if first in r['Kernel_Name']]
idx = idx[-41:]
dur = collections.defaultdict(list); gap = collections.defaultdict(list); steps = []
for a, b in zip(idx[:-1], idx[1:]):
    seq = rows[a:b]
    steps.append((int(rows[b]['Start_Timestamp']) - int(seq[0]['Start_Timestamp'])) / 1e3)
    for j, r in enumerate(seq):
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:48]
        dur[(j, name)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        nxt = rows[a + j + 1]
        gap[(j, name)].append((int(nxt['Start_Timestamp']) - int(r['End_Timestamp'])) / 1e3)
print(f'steps {len(steps)}: mean {sum(steps)/len(steps):.1f} us, median {sorted(steps)[len(steps)//2]:.1f} us (start to next start)')
for k in sorted(dur):
    d, g = dur[k], gap[k]
    print(f'  {k[0]:2d} {k[1]:50s} {sum(d)/len(d):7.2f} us   then gap {sum(g)/len(g):6.2f} us   (n={len(d)})')
