#!/bin/bash
# Round 4 evidence, part A: the whole GPU suite, smoke, default bench line, guard runs (batch 2).
set -u
OUT=gpurun_out/${1:-r04n}; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --durations=6 2>&1 | grep -v amdgpu.ids | tail -14 | tee $OUT/pytest_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
bash scripts/guard_runs.sh 33 $OUT/guard_runs.txt
echo "== default bench line"
timeout 900 python bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench_default.json; cut -c1-260 $OUT/bench_default.json
