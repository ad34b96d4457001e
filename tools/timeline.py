#!/usr/bin/env python3
"""Timeline of the LAST steps of a rocprofv3 --kernel-trace (+ --hip-trace) run: per step (anchored at the kernel whose name contains
<anchor>), every kernel with its queue, start offset, duration; the step period; and the host-side HIP calls per step.
usage: python tools/timeline.py <dir-with-p_kernel_trace.csv> <anchor> [n_steps=40]"""
import csv, glob, os, statistics, sys, collections
d, anchor = sys.argv[1], sys.argv[2]
n_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
kt = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0]
rows = list(csv.DictReader(open(kt)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
short = lambda s: s.replace('(anonymous namespace)::', '').split('(')[0][:44]
idx = [i for i, r in enumerate(rows) if anchor in r['Kernel_Name']][-(n_steps + 1):]
periods = [(int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3 for a, b in zip(idx[:-1], idx[1:])]
print(f'{kt}\nsteps {len(periods)}: period mean {statistics.mean(periods):.2f} us, median {statistics.median(periods):.2f} us')
per = collections.defaultdict(lambda: {'off': [], 'dur': [], 'q': set()})
for a, b in zip(idx[:-1], idx[1:]):
    t0 = int(rows[a]['Start_Timestamp'])
    seen = collections.Counter()
    for r in rows[a:b]:
        nm = short(r['Kernel_Name']); seen[nm] += 1
        k = (nm, seen[nm])
        per[k]['off'].append((int(r['Start_Timestamp']) - t0) / 1e3)
        per[k]['dur'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        per[k]['q'].add(r.get('Queue_Id', '?'))
print(f"{'kernel':46s} {'queue':>8s} {'n':>4s} {'start_us(med)':>14s} {'dur_us(med)':>12s} {'end_us':>8s}")
for k, v in sorted(per.items(), key=lambda kv: statistics.median(kv[1]['off'])):
    so, du = statistics.median(v['off']), statistics.median(v['dur'])
    print(f"{k[0]:46s} {','.join(sorted(v['q'])):>8s} {len(v['off']):4d} {so:14.2f} {du:12.2f} {so + du:8.2f}")
ht = glob.glob(os.path.join(d, '**', '*hip_api_trace.csv'), recursive=True)
if ht:
    h = list(csv.DictReader(open(ht[0])))
    h.sort(key=lambda r: int(r['Start_Timestamp']))
    # host calls issued during the wall-clock span of the selected steps
    lo, hi = int(rows[idx[0]]['Start_Timestamp']), int(rows[idx[-1]]['Start_Timestamp'])
    sel = [r for r in h if lo <= int(r['Start_Timestamp']) < hi]
    agg = collections.defaultdict(list)
    for r in sel:
        agg[r['Function']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    print(f"host HIP calls inside the span of these {len(periods)} steps (per step = count / steps):")
    for f, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"  {f:40s} {len(v) / len(periods):6.2f} per step, {statistics.mean(v):7.2f} us each, {sum(v) / len(periods):7.2f} us per step")
