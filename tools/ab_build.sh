# usage: tools/ab_build.sh "<hipcc flags A>" "<hipcc flags B>" -- <command...>   (rebuilds be_csr_plan between runs; on the GPU box)
set -e
A="$1"; B="$2"; shift 3
for F in "$A" "$B"; do
  echo "== flags: '$F'"
  touch ${AB_FILE:-brainevent_amd/csrc/be_csr_plan.hip}
  BE_HIPCC_FLAGS="$F" python -c "from brainevent_amd import _lib; _lib.build()"
  "$@" 2>&1 | grep -v amdgpu.ids
done
touch ${AB_FILE:-brainevent_amd/csrc/be_csr_plan.hip}
python -c "from brainevent_amd import _lib; _lib.build()"
