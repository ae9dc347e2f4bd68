#!/bin/bash
# round 6: is the speculative record look of the tap role still worth it in the THROUGHPUT instantiation (-DWNV_TAP_SPEC1=0 switches it off)?
A=wavenet_vocoder_amd/libwnv_hip.so; Z=wavenet_vocoder_amd/libwnv_vS1.so
for B in 40 48 56 64; do bash scripts/ab_any.sh "--batch $B --T 8192 --steps 2 --warmup 1" $A $Z $A $Z; done
bash scripts/ab_any.sh "--workload cfg4_mol_multispeaker --batch 32 --T 8192 --steps 2 --warmup 1" $A $Z $A $Z
bash scripts/ab_any.sh "--workload cfg3b_gaussian30 --batch 48 --T 8192 --steps 2 --warmup 1" $A $Z
