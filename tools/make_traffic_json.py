#!/usr/bin/env python3
"""profiles/traffic.json from the --pmc passes of tools/run_r04_profiles.sh: HBM bytes per step of every HBM-bound configuration
of the bench line.  bytes = 2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024 (MI355X_MICROARCH.md: FETCH_SIZE reports half of a
wide coalesced read on gfx950), summed over the kernels that make up one launch of the roofline's divisor: the dominant kernel
where the roofline is per kernel (C2, its mirror, C5, the C2 shard), both passes of the binned route where it is per step (C4)."""
import collections, csv, glob, json, sys

O = sys.argv[1]
KERNELS = {   # configuration -> (key in traffic.json, kernels whose traffic adds up to one unit)
    'c2': ('hetero_d8_n1000000', ['k_plan_accumulate_d8']),
    'c2_homo': ('homo_h8_n1000000', ['k_plan_accumulate_h8']),
    'c2_gather_mirror': ('gather_mirror_hetero_n1000000', ['k_plan_accumulate_d8']),
    # (the setup of a weighted binned workspace runs statistics steps of its own — blocks of 16, |w| and counted — before the
    #  timed ones: the instantiation of the timed step is named in full)
    'c4': ('c4_hetero', ['k_bin_stream<float, false, 32, false>', 'k_bin_accumulate<false, 32, true>', 'k_compact_spikes']),
    'c4_homo': ('c4_homo', ['k_bin_stream', 'k_bin_accumulate', 'k_compact_spikes']),
    'c5': ('c5', ['k_densemm_mfma']),
    'c2_rank_of_8': ('c2_rank_of_8', ['k_plan_accumulate_d8']),
    'c4_rank_of_8': ('c4_rank_of_8', ['k_bin_stream<float, false, 32, false>', 'k_bin_accumulate<false, 32, false>', 'k_compact_bits']),
}


def means(pattern):
    agg = collections.defaultdict(list)
    for p in glob.glob(pattern):
        for r in csv.DictReader(open(p)):
            agg[r['Kernel_Name'].replace('(anonymous namespace)::', '')].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in agg.items()}


out = {'_note': 'HBM bytes per launch / step from rocprofv3 PMC passes (separate --pmc FETCH_SIZE and --pmc WRITE_SIZE runs per '
                'configuration, profiles/r04_*_pmc_all.txt), corrected as MI355X_MICROARCH.md prescribes for gfx950: bytes = '
                '2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024.  `kernels`: what one unit consists of.'}
for cfg, (key, names) in KERNELS.items():
    f, w = means(f'{O}/p_{cfg}_FETCH_SIZE/*counter_collection.csv'), means(f'{O}/p_{cfg}_WRITE_SIZE/*counter_collection.csv')
    if not f or not w:
        continue
    parts, total = {}, 0.0
    for n in names:
        match = (lambda k: n in k) if '<' in n else (lambda k: n + '<' in k or n + '(' in k)
        hit_f = [v for k, v in f.items() if match(k)]
        hit_w = [v for k, v in w.items() if match(k)]
        if not hit_f and not hit_w:
            continue
        fk, wk = (sum(hit_f) / len(hit_f) if hit_f else 0.0), (sum(hit_w) / len(hit_w) if hit_w else 0.0)
        parts[n] = {'FETCH_SIZE_KB': round(fk, 1), 'WRITE_SIZE_KB': round(wk, 1)}
        total += 2 * fk * 1024 + wk * 1024
    if parts:
        out[key] = {'kernels': parts, 'hbm_bytes_per_launch': int(total), 'source': f'profiles/r04_pmc_all.txt ({cfg})'}
print(json.dumps(out, indent=2))
