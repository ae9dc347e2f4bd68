#!/bin/bash
# a lottery, knowingly: the same sources under other scheduling flags (the stage loop's schedule moved +-3 % with unrelated changes this round)
set -u
OUT=gpurun_out/${1:-r04ae}; mkdir -p $OUT
{
for i in 1 2; do
timeout 300 python scripts/exp_rate.py cfg2_mol 8192 8 product
for v in o2 nopost relaxed; do
WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_v_$v.so timeout 300 python scripts/exp_rate.py cfg2_mol 8192 8 $v
done
done
timeout 300 python scripts/exp_rate.py cfg2_mol 8192 48 product
for v in o2 nopost relaxed; do
WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_v_$v.so timeout 300 python scripts/exp_rate.py cfg2_mol 8192 48 $v
done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/rates.txt
