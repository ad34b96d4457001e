# C2 with one homogeneous weight (h8 layout): kernel stats + the two PMC passes for its HBM traffic
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/homo_h8
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 $R/bench.py --homo --no-cpu --steps 100 --warmup 20 > $O/stats.log 2>&1
echo "stats rc=$?"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $R/bench.py --homo --steps 20 --warmup 5 --no-cpu > $O/fetch.log 2>&1
echo "fetch rc=$?"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $R/bench.py --homo --steps 20 --warmup 5 --no-cpu > $O/write.log 2>&1
echo "write rc=$?"
cd $R
echo "## c2 homo (h8) kernel stats"
python tools/summarize_prof.py $O/stats/p_kernel_stats.csv | grep -v "at::native\|rocclr" | head -8
grep '^{"metric"' $O/stats.log | tail -1
echo "## PMC"
python tools/summarize_prof.py "$O/fetch/*counter_collection.csv" "$O/write/*counter_collection.csv" | grep -i "plan_acc\|plan_red\|compact\|kernel "
