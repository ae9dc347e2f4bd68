#!/bin/bash
# round 6: tap role, EVERY round's publish held back behind the next pass's barrier (-DWNV_TAP_DEFER=2): parity of the throughput / packed
# instantiations, then a same-box A/B against the tree's library (last round only).
A=wavenet_vocoder_amd/libwnv_hip.so; Z=wavenet_vocoder_amd/${1:-libwnv_d2.so}
WNV_LIB=$PWD/$Z timeout 1500 python -m pytest tests/test_gpu_vs_reference.py tests/test_gpu_seed_determinism.py tests/test_gpu_packed.py -x -q -k "throughput or packed or determinism" 2>&1 | tail -4
for B in 32 40 48 56 64; do bash scripts/ab_any.sh "--batch $B --T 8192 --steps 2 --warmup 1" $A $Z $A $Z; done
for W in cfg1_mulaw256 cfg4_mol_multispeaker cfg3b_gaussian30; do
  bash scripts/ab_any.sh "--workload $W --batch 48 --T 8192 --steps 2 --warmup 1" $A $Z $A $Z
done
for lib in $A $Z; do
  echo "packed job 100 utterances, $lib"; WNV_LIB=$PWD/$lib python bench.py --job 100 --packed --cpu-steps 0 --no-extras 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(j["value"])'
done
for L in traceD2 traceD2W4; do [ -f wavenet_vocoder_amd/libwnv_$L.so ] && B=64 ROWS=5 REL=barrier WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_$L.so python scripts/trace_tap.py gpurun_out/tap_raw_$L.txt; done
