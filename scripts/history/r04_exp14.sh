#!/bin/bash
# same-box A/B of the headline configuration: library of the commit before the head changes of item 5 against the product
set -u
OUT=gpurun_out/${1:-r04y}; mkdir -p $OUT
{
for i in 1 2; do
WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_base.so timeout 300 python scripts/exp_rate.py cfg2_mol 8192 8 base
timeout 300 python scripts/exp_rate.py cfg2_mol 8192 8 product
done
WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_base.so timeout 300 python scripts/exp_rate.py cfg2_mol 8192 1,16,48 base
timeout 300 python scripts/exp_rate.py cfg2_mol 8192 1,16,48 product
timeout 300 python scripts/exp_rate.py cfg3_gaussian 8192 8 product
WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_base.so timeout 300 python scripts/exp_rate.py cfg3_gaussian 8192 8 base
timeout 300 python scripts/exp_rate.py cfg1_mulaw256 8192 1,8 product
timeout 300 python scripts/exp_rate.py cfg4_mol_multispeaker 8192 8 product
} 2>&1 | grep -v amdgpu.ids | tee $OUT/rates.txt
timeout 900 python -m pytest tests/test_gpu_ring.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -4 | tee $OUT/pytest.txt
