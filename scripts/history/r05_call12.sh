#!/bin/bash
# round 5, GPU call 12: K = 512 with non-temporal loads for the streamed skip passes (same-box A/B)
set -u
OUT=gpurun_out/r05l
mkdir -p $OUT
for a in "--workload cfg4_mol_multispeaker --batch 8 --T 8192" "--workload cfg4_mol_multispeaker --batch 16 --T 8192" "--workload cfg4_mol_multispeaker --batch 32 --T 8192"; do
  echo "-- $a"; bash scripts/ab_any.sh "$a --steps 3 --warmup 1" wavenet_vocoder_amd/libwnv_hip.so wavenet_vocoder_amd/libwnv_nt.so wavenet_vocoder_amd/libwnv_hip.so wavenet_vocoder_amd/libwnv_nt.so 2>&1 | tee -a $OUT/ab_nt.txt
done
