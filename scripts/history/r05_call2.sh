#!/bin/bash
# round 5, GPU call 2: the chain phase's tail by DEPTH (ubench_phase variants 10-13); packed slots with the bias row fetched a pass ahead
set -u
OUT=gpurun_out/r05b
mkdir -p $OUT
export TMPDIR=/tmp
echo "== ubench_phase"; timeout 120 scripts/ubench_phase.bin 2>&1 | tee $OUT/ubench_phase.txt
echo "== packed tests"
timeout 600 python -m pytest tests/test_gpu_packed.py tests/test_gpu_vs_reference.py -m gpu -x -q -k "packed" 2>&1 | tail -5 | tee $OUT/pytest_packed.log
echo "== cfg4 / egs-mol jobs"
for args in "--workload cfg4_mol_multispeaker --job 128 --packed" "--workload cfg4_mol_multispeaker --job 128 --packed --job-group 48" "--workload cfg4_mol_multispeaker --job 128 --job-group 32" "--workload cfg2_mol --job 100 --packed" "--workload cfg2_mol --job 100"; do
  timeout 600 python bench.py $args --steps 1 --warmup 1 2>>$OUT/bench.err | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('$args', j['value'], 'incl padding', j['job']['kSamples_per_s_incl_padding'], 'padding', j['job']['padding_loss'])" | tee -a $OUT/jobs.txt
done
