#!/bin/bash
# diagnostic: at 48 / 64 utterances per GPU, which of a stage's inputs is not there at the first look of its prologue (pre record / h_{l-2} / q)?
set -u
OUT=gpurun_out/${1:-r04ag}; mkdir -p $OUT
WNV_RING_MISS_COUNT=1 WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_miss.so timeout 300 python scripts/exp_rate.py cfg2_mol 4096 48,64 miss 2>&1 | grep -v amdgpu.ids | tee $OUT/miss.txt
