#!/bin/bash
# the q store of a stage no longer waits for the acknowledgement of its u store (loop-invariant loads consumed in front of the loop):
# same-box A/B against the library of HEAD, then parity.
set -u
OUT=gpurun_out/${1:-r04ad}; mkdir -p $OUT
{
for i in 1 2; do
WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_base.so timeout 300 python scripts/exp_rate.py cfg2_mol 8192 8 base
timeout 300 python scripts/exp_rate.py cfg2_mol 8192 8 product
done
WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_base.so timeout 300 python scripts/exp_rate.py cfg2_mol 8192 1,32,48,64 base
timeout 300 python scripts/exp_rate.py cfg2_mol 8192 1,32,48,64 product
WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_base.so timeout 300 python scripts/exp_rate.py cfg1_mulaw256 8192 8,48 base
timeout 300 python scripts/exp_rate.py cfg1_mulaw256 8192 8,48 product
WNV_LIB=$PWD/wavenet_vocoder_amd/libwnv_base.so timeout 300 python scripts/exp_rate.py cfg4_mol_multispeaker 8192 8,16 base
timeout 300 python scripts/exp_rate.py cfg4_mol_multispeaker 8192 8,16 product
} 2>&1 | grep -v amdgpu.ids | tee $OUT/rates.txt
timeout 900 python -m pytest tests/test_gpu_ring.py tests/test_gpu_golden.py tests/test_gpu_packed.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -4 | tee $OUT/pytest.txt
