#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 results database (default output format of `rocprofv3 --kernel-trace`)."""
import glob
import sqlite3
import sys

for pat in sys.argv[1:]:
    for path in sorted(glob.glob(pat)):
        cur = sqlite3.connect(path).cursor()
        rows = cur.execute("select name, count(*), avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, sum(end-start) "
                           "from kernels group by name order by sum(end-start) desc").fetchall()
        total = sum(r[5] for r in rows) or 1
        print(f"# {path}\n{'kernel':84s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>7s}")
        for r in rows[:int(12)]:
            name = r[0].replace('(anonymous namespace)::', '')
            print(f"{name[:84]:84s} {r[1]:6d} {r[2]:10.2f} {r[3]:10.2f} {r[4]:10.2f} {100 * r[5] / total:7.2f}")
