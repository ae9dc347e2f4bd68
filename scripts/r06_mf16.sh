#!/bin/bash
# round 6: the tap role's passes on the matrix pipe, sixteen utterances (two passes) per multiplication -- parity first, then a same-box A/B
A=wavenet_vocoder_amd/libwnv_hip.so; Z=wavenet_vocoder_amd/${1:-libwnv_mf16.so}
WNV_LIB=$PWD/$Z timeout 1500 python -m pytest tests/test_gpu_vs_reference.py tests/test_gpu_seed_determinism.py tests/test_gpu_packed.py tests/test_gpu_golden.py -x -q 2>&1 | tail -6
bash scripts/ab_any.sh "--steps 3 --warmup 1" $A $Z $A $Z
for B in 16 32 40 48 56 64; do bash scripts/ab_any.sh "--batch $B --T 8192 --steps 2 --warmup 1" $A $Z $A $Z; done
for W in cfg1_mulaw256 cfg4_mol_multispeaker cfg3b_gaussian30; do bash scripts/ab_any.sh "--workload $W --batch 48 --T 8192 --steps 2 --warmup 1" $A $Z; done
for lib in $A $Z; do
  echo "packed job 100 utterances, $lib"; WNV_LIB=$PWD/$lib python bench.py --job 100 --packed --cpu-steps 0 --no-extras 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(j["value"])'
done
