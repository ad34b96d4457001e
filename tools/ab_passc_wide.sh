# round 4 (timing experiment): would 16-byte loads help pass C?  BE_DBG_C_WIDE fetches the same bytes of the regions as 16-byte
# pieces (the results are garbage); rocprofv3 kernel averages of pass C at C4
set -e
export AB_FILE=brainevent_amd/csrc/be_csr_binned.hip
run() {
  d=/tmp/ab_wide_$RANDOM
  ( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 $GRAFT_REPO_ROOT/bench.py --workload fcn --steps 40 --warmup 10 --no-cpu --no-secondary > $d.log 2>&1 ) || true
  python3 - <<PY
import csv, glob
rows = {r['Name']: (float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, int(r['Calls'])) for p in glob.glob("$d/*kernel_stats.csv") for r in csv.DictReader(open(p))}
for k, v in rows.items():
    if 'k_bin_accumulate<false' in k or 'k_bin_stream<float, false' in k: print('  ', k[:60], 'avg %.1f min %.1f us, %d calls' % v)
PY
}
export -f run
bash tools/ab_build.sh "" "-DBE_DBG_C_WIDE" -- bash -c run
