#!/bin/bash
# both schedules of the rank step against an exchange of realistic LENGTH (BE_EXCHANGE_EMULATE_US: a spin kernel behind the one-rank all-gather)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for us in 0 5 9 13; do for f in "" "--fcn"; do for s in seq ahead_auto; do
  echo -n "emulated +${us} us: "; BE_EXCHANGE_EMULATE_US=$us timeout -k 10 200 python3 tools/rank_step_lab.py $f --schedule $s --check --steps 400 2>&1 | grep "rank 0 of" | tail -1
done; done; done | tee gpurun_out/r06_rank_emulated_exchange.txt
