#!/bin/bash
# round 6, call 1: the new parity tests first (seed determinism across instantiations, graft on the real reference class), then the whole
# -m gpu suite, smoke, the default bench line and the one-hot configurations the universal log-domain pick touches.
set -u
OUT=gpurun_out/r06a; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_seed_determinism.py tests/test_gpu_graft_reference.py -m gpu -q -s --durations=8 2>&1 | tail -40 > $OUT/new_tests.log
tail -15 $OUT/new_tests.log
timeout 1500 python -m pytest tests -m gpu -x -q --durations=10 2>&1 | tail -30 > $OUT/pytest.log
tail -8 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.log
timeout 600 python bench.py 2>&1 | grep '^{' | tail -1 > $OUT/bench_default.json
python -c "import json; d=json.load(open('$OUT/bench_default.json')); print(d['value'], d['ms_per_step'], d.get('throughput_mode'))"
for w in cfg1_mulaw256 cfg1b_mulaw256_intree cfg0_mulaw256_small; do
  for B in 1 8 48; do
    timeout 300 python bench.py --workload $w --steps 2 --T 8192 --batch $B --cpu-steps 0 --no-extras 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$w B=$B', d['value'])" | tee -a $OUT/onehot.txt
  done
done
