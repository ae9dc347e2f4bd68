#!/bin/bash
# Round 4, final evidence on the final library, one GPU-box call: the whole GPU suite, smoke, default bench line + rocprofv3 kernel stats of
# the same command + PMC passes + fine timeline (gpu_profile_round.sh), then every BASELINE configuration, batch curve, job mode
# (gpu_final_numbers.sh).
set -u
OUT=gpurun_out/${1:-r04fin}; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --durations=6 2>&1 | grep -v amdgpu.ids | tail -14 | tee $OUT/pytest_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
bash scripts/gpu_profile_round.sh ${1:-r04fin} 2>&1 | tee $OUT/profile_round.log | tail -60
bash scripts/gpu_final_numbers.sh 2>&1 | tee $OUT/final_numbers.txt
