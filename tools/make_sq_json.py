#!/usr/bin/env python3
"""profiles/sq_counters.json from the SQ --pmc passes of tools/run_profiles.sh (<out dir>/q_<cfg>_<counters>/): per launch of the
JITC walk kernels, median over the launches.  bench.py derives the walks' vector-ALU utilisation from it:
busy fraction = SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x clock x kernel time)  (SQ_ACTIVE_INST_* count quad-cycles,
MI355X_MICROARCH.md constants table)."""
import collections, csv, glob, json, statistics, sys

O = sys.argv[1]
out = {'_note': 'SQ counters per launch (median) from separate rocprofv3 --pmc passes: bench.py --workload jitc [--jit-gather] '
                '--steps 6 --warmup 2; SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES / SQ_BUSY_CYCLES count quad-cycles; GRBM_GUI_ACTIVE is '
                'the sum over the 8 XCDs'}
for cfg, kern in (('c3', 'k_jit_mv_scatter'), ('c3_gather', 'k_jit_mv_gather')):
    vals = collections.defaultdict(list)
    for p in glob.glob(f'{O}/q_{cfg}_S*/*counter_collection.csv'):
        if cfg == 'c3' and 'q_c3_gather' in p:
            continue
        for r in csv.DictReader(open(p)):
            if kern in r['Kernel_Name']:
                vals[r['Counter_Name']].append(float(r['Counter_Value']))
    if vals:
        out[cfg] = {'kernel': kern, **{k: statistics.median(v) for k, v in sorted(vals.items())}, 'launches': max(len(v) for v in vals.values())}
json.dump(out, sys.stdout, indent=2)
