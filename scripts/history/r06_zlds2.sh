#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_vs_reference.py tests/test_gpu_seed_determinism.py tests/test_gpu_packed.py tests/test_gpu_inkernel_noise.py tests/test_gpu_postchain.py -x -q 2>&1 | tail -3
A=wavenet_vocoder_amd/libwnv_hip.so; Z=wavenet_vocoder_amd/libwnv_vZ0.so
for B in 8 48; do bash scripts/ab_any.sh "--batch $B --T 8192 --steps 2 --warmup 1" $A $Z; done
bash scripts/ab_any.sh "--workload cfg4_mol_multispeaker --batch 32 --T 8192 --steps 2 --warmup 1" $A $Z
for lib in $A $Z $A $Z; do
  for spec in "cfg2_mol 100" "cfg2_mol 200" "cfg4_mol_multispeaker 128" "cfg3b_gaussian30 64" "cfg1_mulaw256 100"; do set -- $spec
    echo -n "$lib packed job $1 $2: "; WNV_LIB=$PWD/$lib python bench.py --workload $1 --job $2 --packed --cpu-steps 0 --no-extras 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(j["value"])'
  done
done
