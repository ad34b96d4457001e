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
    for pat in sys.argv[1:]:
        for p in sorted(glob.glob(pat)):
            if p.endswith('kernel_stats.csv'):
                kernel_stats(p, out)
            elif p.endswith('counter_collection.csv'):
                counters(p, out)
