#!/bin/bash
# round 5, GPU call 11: the whole suite + smoke on the final library; then K = 512 with non-temporal loads for the streamed skip passes
set -u
OUT=gpurun_out/r05k
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
echo "== A/B K = 512 stream loads"
for a in "--workload cfg4_mol_multispeaker --batch 8 --T 8192" "--workload cfg4_mol_multispeaker --batch 16 --T 8192" "--workload cfg4_mol_multispeaker --batch 32 --T 8192" "--workload cfg4_mol_multispeaker --batch 1 --T 8192"; do
  echo "-- $a"; bash scripts/ab_any.sh "$a --steps 3 --warmup 1" wavenet_vocoder_amd/libwnv_hip.so wavenet_vocoder_amd/libwnv_nt.so wavenet_vocoder_amd/libwnv_hip.so wavenet_vocoder_amd/libwnv_nt.so 2>&1 | tee -a $OUT/ab_nt.txt
done
echo "== cfg4 packed at 48 slots"
timeout 600 python bench.py --workload cfg4_mol_multispeaker --job 128 --packed --job-group 48 --steps 1 --warmup 1 2>>$OUT/bench.err | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], 'incl padding', j['job']['kSamples_per_s_incl_padding'], 'padding', j['job']['padding_loss'])" | tee $OUT/job48.txt
