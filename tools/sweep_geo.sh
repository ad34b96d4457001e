# geometry variants for short-block plans (u16): minimal slice count (rounded to 8) vs the CU-balanced count
export BE_EXP_LAYOUTS=u16
run() { BE_EXP_NS=$1 BE_EXP_WIDTH16=$3 $2 timeout -k 10 300 python tools/exp_layouts.py 2>&1 | grep -v amdgpu.ids; }
run 1500000 "env BE_EXP_HETERO_ONLY=1" 15625   # 96
run 1500000 "env BE_EXP_HETERO_ONLY=1" 11719   # 128
run 1500000 "env BE_EXP_HOMO_ONLY=1" 31250     # 48
run 1500000 "env BE_EXP_HOMO_ONLY=1" 29412     # 51
run 2500000 "env BE_EXP_HOMO_ONLY=1" 31250     # 80
run 2500000 "env BE_EXP_HOMO_ONLY=1" 29412     # 85
run 4000000 "env BE_EXP_HOMO_ONLY=1" 32768     # 123
run 4000000 "env BE_EXP_HOMO_ONLY=1" 31250     # 128
run 700000 "env BE_EXP_HETERO_ONLY=1" 14584    # 48
run 700000 "env BE_EXP_HETERO_ONLY=1" 13726    # 51
