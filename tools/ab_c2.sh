# usage: tools/ab_c2.sh "<flags A>" "<flags B>" ...   — C2 headline (value, ms/step, kernel ms, parity) per hipcc flag set
# of be_csr_plan.hip (AB_FILE to rebuild another source); two runs each; on the GPU box
set -e
for F in "$@"; do
  touch ${AB_FILE:-brainevent_amd/csrc/be_csr_plan.hip}
  BE_HIPCC_FLAGS="$F" python -c "from brainevent_amd import _lib; _lib.build()"
  echo "== flags: '$F'"
  for i in 1 2; do python bench.py --steps 200 --warmup 50 --no-cpu --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['parity_check']['ok'])"; done
done
touch ${AB_FILE:-brainevent_amd/csrc/be_csr_plan.hip}; python -c "from brainevent_amd import _lib; _lib.build()"
