#!/bin/bash
# same-box A/B of the one-hot and K = 512 configurations: library of commit 5e465c6 (before item 5) against the product
set -u
OUT=gpurun_out/${1:-r04aa}; mkdir -p $OUT
{
for i in 1 2; do
WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_base.so timeout 300 python scripts/exp_rate.py cfg4_mol_multispeaker 8192 8,16 base
timeout 300 python scripts/exp_rate.py cfg4_mol_multispeaker 8192 8,16 product
done
WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_base.so timeout 300 python scripts/exp_rate.py cfg4_mol_multispeaker 8192 1,32 base
timeout 300 python scripts/exp_rate.py cfg4_mol_multispeaker 8192 1,32 product
WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_base.so timeout 300 python scripts/exp_rate.py cfg1_mulaw256 8192 1,8,16 base
timeout 300 python scripts/exp_rate.py cfg1_mulaw256 8192 1,8,16 product
WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_base.so timeout 300 python scripts/exp_rate.py cfg1b_mulaw256_intree 8192 1,8 base
timeout 300 python scripts/exp_rate.py cfg1b_mulaw256_intree 8192 1,8 product
} 2>&1 | grep -v amdgpu.ids | tee $OUT/rates.txt
