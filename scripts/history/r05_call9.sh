#!/bin/bash
# round 5, GPU call 9: packed slots with seg_start asked for a pass ahead in the tap workgroups -- parity of the packed paths, then the jobs
set -u
OUT=gpurun_out/r05i
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_packed.py tests/test_gpu_vs_reference.py tests/test_gpu_postchain.py tests/test_gpu_zz_boundary.py -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest.log
for args in "--workload cfg2_mol --job 40 --packed" "--workload cfg2_mol --job 100 --packed" "--workload cfg2_mol --job 200 --packed" "--workload cfg4_mol_multispeaker --job 128 --packed" "--workload cfg3b_gaussian30 --job 64 --packed" "--workload cfg1_mulaw256 --job 100 --packed" "--workload cfg1_mulaw256 --job 100"; do
  timeout 600 python bench.py $args --steps 1 --warmup 1 2>>$OUT/bench.err | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('$args', j['value'], 'incl padding', j['job']['kSamples_per_s_incl_padding'], 'padding', j['job']['padding_loss'])" | tee -a $OUT/jobs.txt
done
