# HBM traffic of the binned route at C4 (FixedNumPerPre N = 10M, K = 1000): separate FETCH_SIZE / WRITE_SIZE passes
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_fcn
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/$c -o p -- python3 $R/bench.py --workload fcn --steps 10 --warmup 3 --no-cpu $BENCH_EXTRA > $O/$c.log 2>&1
  echo "$c rc=$?"
done
cd $R
python tools/summarize_prof.py "$O/FETCH_SIZE/*counter_collection.csv" "$O/WRITE_SIZE/*counter_collection.csv" | grep -i "k_bin\|compact\|kernel "
