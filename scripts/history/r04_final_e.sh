#!/bin/bash
# the last call of round 4: whole GPU suite, smoke and the default bench line on the final tree (device code as in r04_final_b.sh; host: native
# Mersenne Twister, bench.py with unconditioned workloads)
set -u
OUT=gpurun_out/${1:-r04fin4}; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --durations=4 2>&1 | grep -v amdgpu.ids | tail -10 | tee $OUT/pytest_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
echo "== default bench line"
timeout 900 python bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench_default.json; cut -c1-260 $OUT/bench_default.json
