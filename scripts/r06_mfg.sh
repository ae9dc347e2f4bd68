#!/bin/bash
# round 6 (last): the matrix-pipe tap units (v4) ONLY for models whose layout has one tap workgroup per layer at every batch size
# (30-layer models, K = 512): parity, then a same-box A/B on those models and a headline check (the legacy form must not move).
A=wavenet_vocoder_amd/libwnv_hip.so; Z=wavenet_vocoder_amd/${1:-libwnv_mfg.so}
WNV_LIB=$PWD/$Z timeout 1500 python -m pytest tests/test_gpu_vs_reference.py tests/test_gpu_seed_determinism.py tests/test_gpu_packed.py tests/test_gpu_golden.py tests/test_gpu_parity_depth.py -x -q 2>&1 | tail -4
bash scripts/ab_any.sh "--steps 3 --warmup 1" $A $Z $A $Z $A $Z
bash scripts/ab_any.sh "--batch 48 --T 8192 --steps 2 --warmup 1" $A $Z
for W in cfg3b_gaussian30 cfg4_mol_multispeaker cfg1b_mulaw256_intree; do for B in 1 8 16 32 48 64; do
  bash scripts/ab_any.sh "--workload $W --batch $B --T 8192 --steps 2 --warmup 1" $A $Z
done; done
for lib in $A $Z; do
  echo "configs[3] job (64 utterances, packed), $lib"; WNV_LIB=$PWD/$lib python bench.py --workload cfg3b_gaussian30 --job 64 --packed --cpu-steps 0 --no-extras 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(j["value"])'
  echo "configs[4] job (128 utterances, packed), $lib"; WNV_LIB=$PWD/$lib python bench.py --workload cfg4_mol_multispeaker --job 128 --packed --cpu-steps 0 --no-extras 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(j["value"])'
done
