#!/bin/bash
# round 6, call 2: the pair stages (run_stage_pair) -- parity of MODE 1 / MODE 2 against the reference and across instantiations, then rates
set -u
OUT=gpurun_out/r06b; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_seed_determinism.py tests/test_gpu_graft_reference.py tests/test_gpu_vs_reference.py tests/test_gpu_packed.py tests/test_gpu_inkernel_noise.py tests/test_gpu_ring.py tests/test_gpu_parity_depth.py -m gpu -x -q --durations=5 2>&1 | tail -25 > $OUT/pytest_pair.log
tail -12 $OUT/pytest_pair.log
for w in cfg2_mol cfg1_mulaw256; do
for B in 32 40 48 56 64; do
  timeout 300 python bench.py --workload $w --steps 2 --T 8192 --batch $B --cpu-steps 0 --no-extras 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$w B=$B', d['value'], 'us/step', round(d['ms_per_step']*1000/8192, 2))" | tee -a $OUT/rates.txt
done
done
for J in 100 200; do
  timeout 600 python bench.py --job $J --steps 1 --warmup 1 --packed 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); j = d['job']; print('job $J packed', d['value'], j['padding_loss'])" | tee -a $OUT/rates.txt
done
