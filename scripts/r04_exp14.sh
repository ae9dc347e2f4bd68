#!/bin/bash
# same-box A/B of the headline configuration: library of the commit before the head changes of item 5 against the product
set -u
OUT=gpurun_out/${1:-r04y}; mkdir -p $OUT
{
for i in 1 2; do
WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_base.so timeout 300 python scripts/exp_rate.py cfg2_mol 8192 8 base
timeout 300 python scripts/exp_rate.py cfg2_mol 8192 8 product
done
WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_base.so timeout 300 python scripts/exp_rate.py cfg2_mol 8192 1,16 base
timeout 300 python scripts/exp_rate.py cfg2_mol 8192 1,16 product
} 2>&1 | grep -v amdgpu.ids | tee $OUT/rates.txt
