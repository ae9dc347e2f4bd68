#!/bin/bash
# round 6: tap role, the next pass's speculative look examined in front of the pass's last publish (-DWNV_TAP_TAKE=1): a hit publishes at once
# and the next [B] waits for nothing; a miss holds the publish back as before.  Parity of the throughput instantiation, then a same-box A/B.
A=wavenet_vocoder_amd/libwnv_hip.so; Z=wavenet_vocoder_amd/${1:-libwnv_take.so}
WNV_LIB=$PWD/$Z timeout 1500 python -m pytest tests/test_gpu_vs_reference.py tests/test_gpu_seed_determinism.py -x -q -k "throughput or determinism" 2>&1 | tail -4
for B in 32 40 48 56 64; do bash scripts/ab_any.sh "--batch $B --T 8192 --steps 2 --warmup 1" $A $Z $A $Z; done
for W in cfg1_mulaw256 cfg4_mol_multispeaker cfg3b_gaussian30; do
  bash scripts/ab_any.sh "--workload $W --batch 48 --T 8192 --steps 2 --warmup 1" $A $Z $A $Z
done
