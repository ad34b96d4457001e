# usage: tools/ab_rank8.sh "<flags A>" "<flags B>" ...  — one rank of an 8-way post split of C2 (bench.py --emulate-world 8) per hipcc
# flag set of be_csr_plan.hip: value, us/step, kernel us; two runs each; on the GPU box
set -e
for F in "$@"; do
  touch brainevent_amd/csrc/be_csr_plan.hip
  BE_HIPCC_FLAGS="$F" python -c "from brainevent_amd import _lib; _lib.build()"
  echo "== flags: '$F'"
  for i in 1 2; do python bench.py --emulate-world 8 --steps 200 --warmup 50 --no-cpu --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d['value'], round(d['ms_per_step']*1e3,2), d['roofline'].get('kernel_ms'))"; done
done
touch brainevent_amd/csrc/be_csr_plan.hip; python -c "from brainevent_amd import _lib; _lib.build()"
