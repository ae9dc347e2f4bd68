#!/bin/bash
# round 6: tap workgroups' bias rows through LDS (WNV_TAP_ZLDS) -- parity of the throughput / packed instantiations, then a same-box A/B
# against the same sources built with -DWNV_TAP_ZLDS=0 (libwnv_vZ0.so)
timeout 1500 python -m pytest tests/test_gpu_vs_reference.py tests/test_gpu_seed_determinism.py tests/test_gpu_packed.py -x -q -k "throughput or packed or determinism" 2>&1 | tail -3
A=wavenet_vocoder_amd/libwnv_hip.so; Z=wavenet_vocoder_amd/libwnv_vZ0.so
for B in 8 32 40 48 56 64; do bash scripts/ab_any.sh "--batch $B --T 8192 --steps 2 --warmup 1" $A $Z $A $Z; done
for W in cfg1_mulaw256 cfg4_mol_multispeaker; do for B in 32 48; do bash scripts/ab_any.sh "--workload $W --batch $B --T 8192 --steps 2 --warmup 1" $A $Z; done; done
for lib in $A $Z; do
  echo "packed job 100 utterances, $lib"; WNV_LIB=$PWD/$lib python bench.py --job 100 --packed --cpu-steps 0 --no-extras 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(j["value"])'
  echo "cfg4 packed job 128, $lib"; WNV_LIB=$PWD/$lib python bench.py --workload cfg4_mol_multispeaker --job 128 --packed --cpu-steps 0 --no-extras 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(j["value"])'
done
