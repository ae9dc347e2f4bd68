#!/bin/bash
# round 6: split tap pass (WNV_TAP_SPLIT) -- parity of the throughput / packed instantiations, then a same-box A/B against -DWNV_TAP_SPLIT=0
OUT=gpurun_out/${1:-r06q}; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_vs_reference.py tests/test_gpu_seed_determinism.py tests/test_gpu_packed.py -x -q -k "throughput or packed or determinism" 2>&1 | tail -4
for B in 32 40 48 56 64; do
  bash scripts/ab_any.sh "--batch $B --T 8192 --steps 2 --warmup 1" wavenet_vocoder_amd/libwnv_hip.so wavenet_vocoder_amd/libwnv_vS0.so wavenet_vocoder_amd/libwnv_hip.so wavenet_vocoder_amd/libwnv_vS0.so
done
for W in cfg1_mulaw256 cfg4_mol_multispeaker; do for B in 48; do
  bash scripts/ab_any.sh "--workload $W --batch $B --T 8192 --steps 2 --warmup 1" wavenet_vocoder_amd/libwnv_hip.so wavenet_vocoder_amd/libwnv_vS0.so
done; done
for lib in libwnv_hip.so libwnv_vS0.so; do
  echo "packed job 100 utterances, $lib"; WNV_LIB=$PWD/wavenet_vocoder_amd/$lib python bench.py --job 100 --packed --cpu-steps 0 --no-extras 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(j["value"])'
done
