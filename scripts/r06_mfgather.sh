#!/bin/bash
# round 6 (last): where the next unit's gathers are issued in run_tap_mf -- behind K group 1 (the tree), in front of the multiplication (-1), behind group 8
A=wavenet_vocoder_amd/libwnv_hip.so
for B in 16 48 64; do bash scripts/ab_any.sh "--workload cfg3b_gaussian30 --batch $B --T 8192 --steps 2 --warmup 1" $A wavenet_vocoder_amd/libwnv_g-1.so wavenet_vocoder_amd/libwnv_g8.so $A wavenet_vocoder_amd/libwnv_g-1.so wavenet_vocoder_amd/libwnv_g8.so; done
bash scripts/ab_any.sh "--workload cfg1b_mulaw256_intree --batch 48 --T 8192 --steps 2 --warmup 1" $A wavenet_vocoder_amd/libwnv_g-1.so wavenet_vocoder_amd/libwnv_g8.so
