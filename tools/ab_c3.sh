# usage: tools/ab_c3.sh "<flags A>" "<flags B>" ...  — C3 (JITC scatter): value, ms/step, kernel ms per hipcc flag set of be_jitc.hip; on the GPU box
set -e
export AB_FILE=brainevent_amd/csrc/be_jitc.hip
run() { python bench.py --workload jitc --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('C3', d['value'], d['ms_per_step'], d['kernel_ms'])"; }
export -f run
bash tools/ab_build.sh "$@" -- bash -c run
