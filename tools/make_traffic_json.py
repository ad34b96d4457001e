#!/usr/bin/env python3
"""profiles/traffic.json from the --pmc passes of tools/run_r04_profiles.sh: HBM bytes per step of every HBM-bound configuration
of the bench line.  bytes = 2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024 (MI355X_MICROARCH.md: FETCH_SIZE reports half of a
wide coalesced read on gfx950), summed over the kernels that make up one launch of the roofline's divisor: the dominant kernel
where the roofline is per kernel (C2, its mirror, C5, the C2 shard), both passes of the binned route where it is per step (C4)."""
import collections, csv, glob, json, os, statistics, sys

O = sys.argv[1]
KERNELS = {   # configuration -> (key in traffic.json, kernels whose traffic adds up to one unit)
    'c2': ('hetero_d8_n1000000', ['k_plan_accumulate_d8']),
    'c2_homo': ('homo_h8_n1000000', ['k_plan_accumulate_h8']),
    'c2_gather_mirror': ('gather_mirror_hetero_n1000000', ['k_plan_accumulate_d8']),
    # (the setup of a weighted binned workspace runs statistics steps of its own — all rows, |w| and counted — before the
    #  timed ones: the last 12 launches of each kernel are the measured loop's, see LAST)
    'c4': ('c4_hetero', ['k_bin_stream', 'k_bin_accumulate', 'k_compact_spikes']),
    'c4_homo': ('c4_homo', ['k_bin_stream', 'k_bin_accumulate', 'k_compact_spikes']),
    'c5': ('c5', ['k_densemm_mfma']),
    'c2_rank_of_8': ('c2_rank_of_8', ['k_plan_accumulate_d8']),
    'c4_rank_of_8': ('c4_rank_of_8', ['k_bin_stream', 'k_bin_accumulate', 'k_compact_bits']),
}


LAST = 12     # the --pmc passes run `--steps 12`: the last 12 launches of a kernel are steps of the measured loop


def launches(pattern):
    agg = collections.defaultdict(list)
    for p in glob.glob(pattern):
        for r in csv.DictReader(open(p)):
            name = r['Kernel_Name'].replace('(anonymous namespace)::', '')
            agg[name[5:] if name.startswith('void ') else name].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    return agg


def per_launch(agg, name):
    """(instantiation, median counter value) over the last LAST launches of kernel `name`, whatever else (statistics steps of a
    workspace setup, other instantiations) ran before them."""
    hits = sorted((d, v, k) for k, lst in agg.items() if k.startswith(name + '<') or k.startswith(name + '(') for d, v in lst)
    if not hits:
        return None, 0.0
    tail = hits[-LAST:]
    return tail[-1][2], statistics.median(v for _, v, _ in tail)


out = {'_note': 'HBM bytes per launch / step from rocprofv3 PMC passes (separate --pmc FETCH_SIZE and --pmc WRITE_SIZE runs per '
                'configuration, profiles/r04_*_pmc_all.txt), corrected as MI355X_MICROARCH.md prescribes for gfx950: bytes = '
                '2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024.  `kernels`: what one unit consists of (median over the last 12 launches of each, i.e. steps of the measured loop).'}
for cfg, (key, names) in KERNELS.items():
    f, w = launches(f'{O}/p_{cfg}_FETCH_SIZE/*counter_collection.csv'), launches(f'{O}/p_{cfg}_WRITE_SIZE/*counter_collection.csv')
    if not f or not w:
        continue
    parts, total = {}, 0.0
    for n in names:
        inst, fk = per_launch(f, n)
        _, wk = per_launch(w, n)
        if inst is None:
            continue
        parts[inst.split('(')[0][:80]] = {'FETCH_SIZE_KB': round(fk, 1), 'WRITE_SIZE_KB': round(wk, 1)}
        total += 2 * fk * 1024 + wk * 1024
    if parts:
        out[key] = {'kernels': parts, 'hbm_bytes_per_launch': int(total), 'source': f'profiles/{os.path.basename(O.rstrip("/"))}_pmc_all.txt ({cfg})'}
print(json.dumps(out, indent=2))
