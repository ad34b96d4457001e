set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R
# 1. default bench under rocprofv3 kernel trace (same command as the bench line)
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -o c2 -- python3 $R/bench.py > $O/bench_c2_prof.log 2>&1
echo "prof c2 rc=$?"
cd $R
timeout -k 10 400 python bench.py > $O/bench_c2.log 2>&1; echo "bench c2 rc=$?"; tail -1 $O/bench_c2.log | cut -c1-300
timeout -k 10 400 python bench.py --homo --no-cpu > $O/bench_c2_homo.log 2>&1; tail -1 $O/bench_c2_homo.log | cut -c1-300
