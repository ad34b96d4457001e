# usage: tools/ab_nt_plan.sh "<flags A>" "<flags B>" ...  — the plan kernels OTHER than the C2 headline per hipcc flag set:
# K = 1000 sweep (sub-wave u16 / d8 decoders), K = 10000 (wave per block), C2 homo (h8); on the GPU box
set -e
run() { (BE_EXP_NS=100000,350000,1000000,2500000 python tools/exp_layouts.py | grep -o "N=.*auto=.*"; BE_EXP_K=10000 BE_EXP_NS=100000,1000000 BE_EXP_LAYOUTS=None python tools/exp_layouts.py; python bench.py --homo --steps 100 --warmup 20 --no-cpu --no-secondary | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('C2 homo', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])") 2>&1 | grep -v amdgpu.ids; }
export -f run
bash tools/ab_build.sh "$@" -- bash -c run
