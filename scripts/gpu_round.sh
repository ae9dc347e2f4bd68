#!/bin/bash
# One GPU-box round trip: parity tests, smoke, bench, rocprofv3 kernel stats.  Everything under its own
# `timeout` so a hung kernel cannot hold the box; outputs go to gpurun_out/.
# usage: scripts/gpu_round.sh [tag] [pytest-extra-args]
set -u
TAG=${1:-r01}
shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== rocm-smi" ; rocm-smi --showproductname 2>/dev/null | head -8
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -x -q "$@" 2>&1 | tail -40 | tee $OUT/pytest.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $OUT/smoke.log
echo "== bench"
timeout 600 python bench.py --steps 2 --warmup 1 2>&1 | tail -3 | tee $OUT/bench.log
echo "== rocprofv3 kernel stats (short bench)"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o wnv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-steps 0 > $GRAFT_REPO_ROOT/$OUT/prof_bench.log 2>&1 )
find $OUT/prof -name "*kernel_stats*" | head -3
for f in $(find $OUT/prof -name "*kernel_stats.csv" | head -1); do head -12 $f | cut -c1-220; done
