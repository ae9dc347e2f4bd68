#!/bin/bash
# round 6: the tap role's mat-vec on the matrix pipe (run_tap, MF).  The whole GPU suite on the new library, then a same-box A/B against the
# library of the commit before (wavenet_vocoder_amd/libwnv_old.so, built with scripts/build_rev.sh or from a checkout).
A=wavenet_vocoder_amd/libwnv_old.so; Z=wavenet_vocoder_amd/libwnv_hip.so
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
bash scripts/ab_any.sh "--steps 3 --warmup 1" $A $Z $A $Z
for B in 16 32 40 48 56 64; do bash scripts/ab_any.sh "--batch $B --T 8192 --steps 2 --warmup 1" $A $Z $A $Z; done
for W in cfg1_mulaw256 cfg4_mol_multispeaker cfg3b_gaussian30; do
  bash scripts/ab_any.sh "--workload $W --batch 48 --T 8192 --steps 2 --warmup 1" $A $Z $A $Z
done
for lib in $A $Z; do
  echo "packed job 100 utterances, $lib"; WNV_LIB=$PWD/$lib python bench.py --job 100 --packed --cpu-steps 0 --no-extras 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(j["value"])'
done
