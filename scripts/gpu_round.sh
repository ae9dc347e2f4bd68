#!/bin/bash
# One GPU-box round trip: parity tests, smoke, bench, rocprofv3 kernel stats.  Everything under its own
# `timeout` so a hung kernel cannot hold the box; outputs go to gpurun_out/<tag>/.
# usage: scripts/gpu_round.sh <tag> <what: all|tests|ring|bench|prof> [bench args]
set -u
TAG=${1:-r01}
WHAT=${2:-all}
shift; shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
if [[ $WHAT == all || $WHAT == tests ]]; then
  echo "== pytest -m gpu"
  timeout 1200 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -40 | tee $OUT/pytest.log
  echo "== smoke"
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $OUT/smoke.log
fi
if [[ $WHAT == ring ]]; then
  echo "== pytest ring"
  timeout 900 python -m pytest tests/test_gpu_ring.py -m gpu -x -q --durations=8 -s 2>&1 | tail -40 | tee $OUT/pytest_ring.log
fi
if [[ $WHAT == all || $WHAT == bench || $WHAT == ring ]]; then
  echo "== bench $*"
  timeout 600 python bench.py "$@" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $OUT/bench.log
fi
if [[ $WHAT == all || $WHAT == prof ]]; then
  echo "== rocprofv3 kernel stats"
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -o wnv -- python $ROOT/bench.py --steps 3 --warmup 1 --no-extras "$@" > $ROOT/$OUT/prof_bench.log 2>&1 )
  grep '^{' $OUT/prof_bench.log | tail -1
  for f in $(find $OUT/prof -name "*kernel_stats.csv" | head -1); do echo $f; head -12 $f | cut -c1-240; done
fi
