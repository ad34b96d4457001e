# round 4: do line-aligned flush units help the binned route?  C4 hetero with 64-bit sums (96-byte blocks -> 128-byte stride) and with the
# opt-in 32-bit sums (192 -> 256), each built without / with -DBE_BLOCK_GALIGN; rocprofv3 kernel averages of pass B / pass C.
set -e
export AB_FILE=brainevent_amd/csrc/be_csr_binned.hip
run() {
  for a in "" "--acc32"; do
    d=/tmp/ab_galign_$RANDOM
    ( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 $GRAFT_REPO_ROOT/bench.py --workload fcn $a --steps 40 --warmup 10 --no-cpu > $d.log 2>&1 )
    python3 - <<PY
import json, csv, glob
line = [l for l in open("$d.log") if l.startswith("{")][-1]
d = json.loads(line)
rows = {r['Name']: float(r['AverageNs']) / 1e3 for p in glob.glob("$d/*kernel_stats.csv") for r in csv.DictReader(open(p))}
pb = [v for k, v in rows.items() if 'k_bin_stream<float, false, %s' % ('32' if '$a' else '16') in k]
pc = [v for k, v in rows.items() if 'k_bin_accumulate<false, %s' % ('32, true' if '$a' else '16, false') in k]
print('C4 hetero', '$a' or '64-bit', 'ms/step', d['ms_per_step'], 'pass B us', pb, 'pass C us', pc)
PY
  done
}
export -f run
bash tools/ab_build.sh "" "-DBE_BLOCK_GALIGN" -- bash -c run
