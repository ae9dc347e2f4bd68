#!/bin/bash
# the speculative look at the next h record in the tap workgroups of the K = 256 instantiations too (it fits now): same-box A/B
set -u
OUT=gpurun_out/${1:-r04af}; mkdir -p $OUT
{
for i in 1 2; do
timeout 300 python scripts/exp_rate.py cfg1_mulaw256 8192 8,48 product
WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_v_spec.so timeout 300 python scripts/exp_rate.py cfg1_mulaw256 8192 8,48 spec
done
timeout 300 python scripts/exp_rate.py cfg1_mulaw256 8192 1,16,32,64 product
WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_v_spec.so timeout 300 python scripts/exp_rate.py cfg1_mulaw256 8192 1,16,32,64 spec
} 2>&1 | grep -v amdgpu.ids | tee $OUT/rates.txt
