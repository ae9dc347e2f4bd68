#!/bin/bash
A=wavenet_vocoder_amd/libwnv_hip.so; Z=wavenet_vocoder_amd/libwnv_vH0.so
timeout 900 python -m pytest tests/test_gpu_vs_reference.py tests/test_gpu_seed_determinism.py -x -q -k "throughput or determinism" 2>&1 | tail -2
for B in 40 48 56 64; do bash scripts/ab_any.sh "--batch $B --T 8192 --steps 2 --warmup 1" $A $Z $A $Z $A $Z; done
bash scripts/ab_any.sh "--workload cfg1_mulaw256 --batch 48 --T 8192 --steps 2 --warmup 1" $A $Z $A $Z
