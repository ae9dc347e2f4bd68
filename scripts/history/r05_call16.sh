#!/bin/bash
# round 5, GPU call 16: the ring's categorical head picks in the log domain (WNV_CAT_LOG) -- parity of everything one-hot, then the A/B
set -u
OUT=gpurun_out/r05p
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q -k "onehot or mulaw or cfg0 or cfg1 or categorical or golden or packed or fuzz or stress or reference or philox or wide_skip" 2>&1 | tail -5 | tee $OUT/pytest.log
for a in "--workload cfg1_mulaw256 --batch 1 --T 8192" "--workload cfg1_mulaw256 --batch 8 --T 8192" "--workload cfg1_mulaw256 --batch 48 --T 8192" "--workload cfg1b_mulaw256_intree --batch 8 --T 8192" "--workload cfg0_mulaw256_small --batch 8 --T 8192"; do
  echo "-- $a"; bash scripts/ab_any.sh "$a --steps 3 --warmup 1" wavenet_vocoder_amd/libwnv_nolog.so wavenet_vocoder_amd/libwnv_hip.so wavenet_vocoder_amd/libwnv_nolog.so wavenet_vocoder_amd/libwnv_hip.so 2>&1 | tee -a $OUT/ab_catlog.txt
done
