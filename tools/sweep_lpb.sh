set -e
python -m pytest tests/test_csr_gpu.py tests/test_plan_contracts_gpu.py -x -q -m gpu 2>&1 | tail -3
for h in "" 1000; do
echo "== hint=${h:-auto}"
BE_EXP_HINT=$h BE_EXP_LAYOUTS=u16 BE_EXP_K=3000 BE_EXP_NS=300000,1000000 python tools/exp_layouts.py 2>&1 | grep -v amdgpu.ids
BE_EXP_HINT=$h BE_EXP_LAYOUTS=u16 BE_EXP_K=2000 BE_EXP_NS=300000,1000000 python tools/exp_layouts.py 2>&1 | grep -v amdgpu.ids
BE_EXP_HINT=$h BE_EXP_LAYOUTS=u16 BE_EXP_K=10000 BE_EXP_NS=300000,1000000 python tools/exp_layouts.py 2>&1 | grep -v amdgpu.ids
done
