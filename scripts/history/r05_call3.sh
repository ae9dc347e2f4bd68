#!/bin/bash
# round 5, GPU call 3: the chain phase with row-pair accumulators + the gate on pre-scaled arguments (WNV_PHASE2) -- parity first, then the
# same-box A/B against the same sources built with -DWNV_PHASE2=0; packed slots with the maps prefetched by scalar loads.
set -u
OUT=gpurun_out/r05c
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest.log
echo "== A/B headline"
bash scripts/ab_bench.sh "--steps 5 --warmup 1" wavenet_vocoder_amd/libwnv_phase0.so wavenet_vocoder_amd/libwnv_hip.so 2>&1 | tee $OUT/ab_headline.txt
echo "== A/B other shapes"
for a in "--batch 48 --T 8192" "--batch 32 --T 8192" "--workload cfg1_mulaw256 --batch 1 --T 8192" "--workload cfg1_mulaw256 --batch 8 --T 8192" "--workload cfg4_mol_multispeaker --batch 8 --T 8192" "--workload cfg4_mol_multispeaker --batch 16 --T 8192" "--workload cfg3b_gaussian30 --batch 8 --T 8192"; do
  echo "-- $a"; bash scripts/ab_any.sh "$a --steps 3 --warmup 1" wavenet_vocoder_amd/libwnv_phase0.so wavenet_vocoder_amd/libwnv_hip.so 2>&1 | tee -a $OUT/ab_shapes.txt
done
echo "== jobs"
for args in "--workload cfg4_mol_multispeaker --job 128 --packed" "--workload cfg2_mol --job 100 --packed"; do
  timeout 600 python bench.py $args --steps 1 --warmup 1 2>>$OUT/bench.err | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('$args', j['value'], 'incl padding', j['job']['kSamples_per_s_incl_padding'], 'padding', j['job']['padding_loss'])" | tee -a $OUT/jobs.txt
done
