# cache-policy bits of the d8 block loads (BE_BLOCK_AUX): C2 headline kernel time per value; on the GPU box
set -e
for a in 0 2 3 18 16; do
  touch brainevent_amd/csrc/be_csr_plan.hip
  BE_HIPCC_FLAGS="-DBE_BLOCK_AUX=$a" python -c "from brainevent_amd import _lib; _lib.build()"
  echo "== aux=$a"
  for i in 1 2; do python bench.py --steps 200 --warmup 50 --no-cpu --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['parity_check']['ok'])"; done
done
touch brainevent_amd/csrc/be_csr_plan.hip; python -c "from brainevent_amd import _lib; _lib.build()"
