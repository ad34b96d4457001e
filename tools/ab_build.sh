# usage: tools/ab_build.sh "<hipcc flags A>" "<hipcc flags B>" [...] -- <command...>
# rebuilds AB_FILE (default be_csr_plan.hip) with each flag set and runs the command; on the GPU box
set -e
SETS=()
while [ "$#" -gt 0 ] && [ "$1" != "--" ]; do SETS+=("$1"); shift; done
shift
for F in "${SETS[@]}"; do
  echo "== flags: '$F'"
  touch ${AB_FILE:-brainevent_amd/csrc/be_csr_plan.hip}
  BE_HIPCC_FLAGS="$F" python -c "from brainevent_amd import _lib; _lib.build()"
  "$@" 2>&1 | grep -v amdgpu.ids
done
touch ${AB_FILE:-brainevent_amd/csrc/be_csr_plan.hip}
python -c "from brainevent_amd import _lib; _lib.build()"
