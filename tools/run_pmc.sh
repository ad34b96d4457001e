# HBM traffic of the dominant kernel: separate --pmc passes (FETCH_SIZE, WRITE_SIZE), default bench command, few steps
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_r01f
mkdir -p $O
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu $BENCH_EXTRA > $O/fetch.log 2>&1
echo "fetch rc=$?"
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu $BENCH_EXTRA > $O/write.log 2>&1
echo "write rc=$?"
cd $R
python tools/summarize_prof.py "$O/fetch/*counter_collection.csv" "$O/write/*counter_collection.csv" | grep -i "plan_acc\|plan_red\|compact\|kernel "
