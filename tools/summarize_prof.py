#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats / counter collection) into a short text table."""
import collections
import csv
import glob
import sys


def kernel_stats(path, out):
    rows = list(csv.DictReader(open(path)))
    out.write(f"# {path}\n{'kernel':80s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>7s}\n")
    for r in rows:
        name = r['Name'].replace('(anonymous namespace)::', '')
        if 'at::native' in name or 'rocclr' in name:
            name = name[:60] + '…'
        out.write(f"{name[:80]:80s} {r['Calls']:>6s} {float(r['AverageNs']) / 1e3:10.2f} {float(r['MinNs']) / 1e3:10.2f} "
                  f"{float(r['MaxNs']) / 1e3:10.2f} {float(r['Percentage']):7.2f}\n")


def kernel_trace(path, out, last=0):
    """Per kernel, in dispatch order: calls, mean, MEDIAN, and the median over the LAST `last` launches (the timed steps of a
    bench run: warm-up launches — first-touch page faults, cold caches, allocator growth — pollute a plain average by 30-60 %;
    `last` = 0: the last half of the launches), min, max.  From rocprofv3's p_kernel_trace.csv (Start/End timestamps, ns)."""
    import statistics
    per = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        per.setdefault(r['Kernel_Name'], []).append((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
    tot = sum(d for v in per.values() for _, d in v) or 1
    out.write(f"# {path}\n{'kernel':72s} {'calls':>6s} {'avg_us':>9s} {'med_us':>9s} {'med_timed':>9s} {'n_timed':>7s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}\n")
    rows = []
    for name, v in per.items():
        v.sort()
        d = [x[1] / 1e3 for x in v]
        n_t = min(len(d), last) if last > 0 else max(1, len(d) // 2)
        timed = d[-n_t:]
        short = name.replace('(anonymous namespace)::', '')
        if 'at::native' in short or 'rocclr' in short:
            short = short[:60] + '…'
        rows.append((sum(d), f"{short[:72]:72s} {len(d):6d} {sum(d) / len(d):9.2f} {statistics.median(d):9.2f} {statistics.median(timed):9.2f} "
                             f"{n_t:7d} {min(d):9.2f} {max(d):9.2f} {100 * sum(d) * 1e3 / tot:6.2f}\n"))
    for _, line in sorted(rows, reverse=True):
        out.write(line)


def counters(path, out):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(list)
    for r in rows:
        agg[(r['Kernel_Name'].replace('(anonymous namespace)::', '')[:70], r['Counter_Name'])].append(float(r['Counter_Value']))
    out.write(f"# {path}\n{'kernel':70s} {'counter':24s} {'n':>5s} {'mean':>16s}\n")
    for (k, c), v in sorted(agg.items()):
        if 'at::native' in k or 'rocclr' in k:
            continue
        out.write(f"{k:70s} {c:24s} {len(v):5d} {sum(v) / len(v):16.1f}\n")


if __name__ == '__main__':
    out = sys.stdout
    args = sys.argv[1:]
    last = 0
    if args and args[0] == '--last':          # --last N: the timed launches are the last N of every kernel
        last, args = int(args[1]), args[2:]
    for pat in args:
        for p in sorted(glob.glob(pat)):
            if p.endswith('kernel_trace.csv'):
                kernel_trace(p, out, last)
            elif p.endswith('kernel_stats.csv'):
                kernel_stats(p, out)
            elif p.endswith('counter_collection.csv'):
                counters(p, out)
