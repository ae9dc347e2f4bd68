#!/bin/bash
# round 5, GPU call 7: K = 512 -- every stage hands its own skip term to the head parts (WNV_SKIP_DIRECT); addends as accumulator init (ZACC) as the default.
set -u
OUT=gpurun_out/r05g
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 | tee $OUT/pytest.log
echo "== A/B cfg4"
for a in "--workload cfg4_mol_multispeaker --batch 8 --T 8192" "--workload cfg4_mol_multispeaker --batch 16 --T 8192" "--workload cfg4_mol_multispeaker --batch 32 --T 8192" "--workload cfg4_mol_multispeaker --batch 1 --T 8192"; do
  echo "-- $a"; bash scripts/ab_any.sh "$a --steps 3 --warmup 1" wavenet_vocoder_amd/libwnv_nosd.so wavenet_vocoder_amd/libwnv_hip.so wavenet_vocoder_amd/libwnv_nosd.so wavenet_vocoder_amd/libwnv_hip.so 2>&1 | tee -a $OUT/ab_cfg4.txt
done
echo "== headline"; bash scripts/ab_any.sh "--steps 5 --warmup 1" wavenet_vocoder_amd/libwnv_phase0.so wavenet_vocoder_amd/libwnv_hip.so 2>&1 | tee $OUT/ab_headline.txt
echo "== jobs"
for args in "--workload cfg4_mol_multispeaker --job 128 --packed" "--workload cfg4_mol_multispeaker --job 128 --job-group 32" "--workload cfg2_mol --job 200 --packed"; do
  timeout 600 python bench.py $args --steps 1 --warmup 1 2>>$OUT/bench.err | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('$args', j['value'], 'incl padding', j['job']['kSamples_per_s_incl_padding'], 'padding', j['job']['padding_loss'])" | tee -a $OUT/jobs.txt
done
